#!/bin/bash
# Diagnostic: per-rank step time of a W-rank simplex-sharded run, measured on ONE GPU (no collective) - EVERY rank of
# W = 8 (the run's step is the MAX over its ranks), one rank each of W = 1, 2, 4.
# usage: tools/emulate_scaling.sh [workload]     SPECS="0/1 0/2 ..." overrides the list
wl=${1:-cfg2}
for spec in ${SPECS:-0/1 0/2 0/4 0/8 1/8 2/8 3/8 4/8 5/8 6/8 7/8}; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-cold --emulate-shard $spec $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['per_rank'][0]
print('$spec', 'step', d['ms_per_step'], 'index ready', d['ms_per_step_index_ready'], 'spans', r['spans'], 'ready', r['spans_index_ready'])"
done | tee /tmp/emul_$wl.txt
python - /tmp/emul_$wl.txt <<'PY'
import sys, re
rows = [l.split() for l in open(sys.argv[1]) if "/8 " in l]
if rows:
    st = [float(r[2]) for r in rows]; rd = [float(r[5]) for r in rows]
    print(f"W = 8, all {len(rows)} ranks: step max {max(st):.3f} min {min(st):.3f} (spread {100 * (max(st) / min(st) - 1):.0f} %), index ready max {max(rd):.3f} min {min(rd):.3f}")
PY
