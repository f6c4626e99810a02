#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r5o; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt | cut -c1-400
