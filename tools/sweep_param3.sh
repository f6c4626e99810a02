#!/bin/bash
wl=$1; flag=$2; shift; shift
for v in "$@"; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline $flag $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $flag', '$v', d['ms_per_step'], d['kernels_ms_per_step'])"
done
