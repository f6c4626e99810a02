#!/bin/bash
# round 5 session a: tests + queue-block A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r5a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
bash tools/ab_bench.sh r5a \
  "cfg5 --option cell_queue_block=-1" "cfg5 --option cell_queue_block=3" "cfg5 --option cell_queue_block=5" "cfg5 --option cell_queue_block=7" "cfg5 --option cell_queue_block=9" \
  "cfg3 --option cell_queue_block=-1" "cfg3 --option cell_queue_block=5" "cfg3 --option cell_queue_block=7" \
  "cfg2 --option cell_queue_block=-1" "cfg2 --option cell_queue_block=5" "cfg2 --option cell_queue_block=3" 2>&1 | tee $OUT/ab.txt
