#!/bin/bash
# usage: tools/ab.sh <workload> "<flags A>" "<flags B>" ...  - one bench line per flag set
wl=$1; shift
for f in "$@"; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['sweep_stats_rank0'] or {}
print('$wl [$f]', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()}, {k:s.get(k) for k in ('restage_rounds','tiles_flagged','exhaustive_rounds','cell_pairs','points_staged','fallback_leaves_evaluated','fallback_leaves_tested','fallback_nodes_expanded','fallback_max_tests_one_tile','giveup_gather_density','giveup_gather_stage','giveup_lds_full','giveup_doublings','tiles_total')})"
done
