#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests -m gpu -x -q --timeout 120 > gpurun_out/r5g/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5g/pytest_gpu.txt | cut -c1-400
