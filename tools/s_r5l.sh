#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests -m gpu -x -q --timeout 150 > gpurun_out/r5l/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5l/pytest_gpu.txt | cut -c1-600
bash tools/ab_bench.sh r5l/ab "cfg2" "cfg3" "cfg5" 2>&1 | cut -c1-240
