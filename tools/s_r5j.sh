#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5j
bash tools/ab_bench.sh r5j/ab "cfg2" "cfg2 --option finish_budget_min=32" "cfg2 --option finish_budget_min=64" "cfg2 --option finish_budget_min=128" "cfg2 --option finish_budget_min=100000" \
  "cfg3 --emulate-shard 3/8" "cfg3 --emulate-shard 3/8 --option finish_budget_min=64" "cfg3 --emulate-shard 3/8 --option finish_budget_min=256" \
  "cfg5 --emulate-shard 3/8" "cfg5 --emulate-shard 3/8 --option finish_budget_min=64" "cfg5 --emulate-shard 3/8 --option finish_budget_min=256" \
  "cfg2 --emulate-shard 3/8" "cfg2 --emulate-shard 3/8 --option finish_budget_min=64" 2>&1 | cut -c1-190
