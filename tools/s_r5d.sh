#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export OUTTAG=r5d
mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5d/pytest_gpu.txt
run() { # name lib workload flags...
  name=$1; lib=$2; shift 2
  if [ "$lib" != product ]; then export FLOODER_HIP_LIB=$R/gpurun_in/$lib.so; else unset FLOODER_HIP_LIB; fi
  bash tools/ab_bench.sh $OUTTAG/$name "$*" 2>&1 | sed "s/^/[$name] /"
}
for wl in cfg5 cfg3 cfg2; do
run p_$wl product $wl
run w3_$wl w3 $wl --option cell_grid=768
done
