#!/bin/bash
# A/B of cell sweep options on full runs and emulated shares.  usage: tools/shard_ab.sh "<workloads>" "<specs>" "<variant>|<variant>|..."
run() { wl=$1; spec=$2; shift 2; timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-cold --emulate-shard $spec "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $spec $*', d['ms_per_step_index_ready'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if k in ('sweep','fallback')})"; }
IFS='|' read -ra VARS <<< "${3:-}"
for wl in ${1:-cfg2}; do
  for spec in ${2:-0/1 0/8}; do
    run $wl $spec
    for v in "${VARS[@]}"; do run $wl $spec $v; done
  done
done
