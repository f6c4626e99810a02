#!/bin/bash
# Parity evidence outside the test suite (one gpurun call): every simplex of the full-size workloads against a kd-tree
# over all points, and the randomised stress runs (2D / 3D and 4 - 7 D).   tools/collect_parity.sh r5
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
TAG=${1:-r6}; OUT=$R/gpurun_out/parity_$TAG/profiles; mkdir -p $OUT
for wl in cfg2 cfg3 cfg5 cfg4; do
  timeout 900 python tools/every_simplex.py $wl 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_every_simplex_$wl.txt; tail -2 $OUT/${TAG}_every_simplex_$wl.txt | cut -c1-300
done
timeout 900 python tools/stress_parity.py 48 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stress_parity.txt; tail -2 $OUT/${TAG}_stress_parity.txt | cut -c1-300
timeout 900 python tools/stress_parity_hd.py 30 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stress_parity_hd.txt; tail -2 $OUT/${TAG}_stress_parity_hd.txt | cut -c1-300
