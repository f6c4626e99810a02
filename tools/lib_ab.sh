#!/bin/bash
# Same-box A/B of library variants (gpurun_in/<name>.so | product).  usage: tools/lib_ab.sh "<workloads>" "<lib>[:flags,with,commas] ..." [repeats]
for rep in $(seq ${3:-2}); do
for wl in $1; do
  for spec in $2; do
    IFS=: read -r lib flags <<< "$spec"
    if [ "$lib" != product ]; then export FLOODER_HIP_LIB=$PWD/gpurun_in/$lib.so; else unset FLOODER_HIP_LIB; fi
    timeout 300 python bench.py --workload $wl --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-cold ${flags//,/ } 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $spec', d['ms_per_step'], '+-', d['ms_per_step_std'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if k in ('sweep','fallback','sweep_bvh')})"
  done
done
done
