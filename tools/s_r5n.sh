#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r5n; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --timeout 200 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt | cut -c1-400
bash tools/ab_bench.sh r5n/ab "cfg2" "cfg3" "cfg5" 2>&1 | cut -c1-200
