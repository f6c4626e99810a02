#!/bin/bash
# Diagnostic: per-rank step time of a W-rank simplex-sharded run, measured on ONE GPU (no collective).
# usage: tools/emulate_scaling.sh [workload]
wl=${1:-cfg2}
for spec in ${SPECS:-0/1 0/2 0/4 0/8 3/8 7/8}; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-cold --emulate-shard $spec $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$spec', d['ms_per_step'], 'rows', d['config'].get('sub_cloud_rows_rank0'), 'index ready', d['ms_per_step_index_ready'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
done
