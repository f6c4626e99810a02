#!/bin/bash
# usage: tools/sweep_param.sh <bench flag> v1 v2 ...   e.g. tools/sweep_param.sh --alpha 1.0 1.2 1.35
#        tools/sweep_param.sh --option cell_grid=512 cell_grid=768   (library switches)
flag=$1; shift
for v in "$@"; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $flag $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['sweep_stats_rank0'] or {}
print('$flag', '$v', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()}, {k:s.get(k) for k in ('restage_rounds','tiles_flagged','exhaustive_rounds','cell_pairs')})"
done
