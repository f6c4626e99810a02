#!/bin/bash
# per-kernel average durations of a workload for library variants.  usage: tools/lib_trace.sh <workload> "<lib> ..."
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
wl=$1
for lib in $2; do
  if [ "$lib" != product ]; then export FLOODER_HIP_LIB=$R/gpurun_in/$lib.so; else unset FLOODER_HIP_LIB; fi
  OUT=$R/gpurun_out/lt_${wl}_$lib; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-cold ${EXTRA_BENCH:-} > $OUT/bench.json 2> $OUT/err.txt
  echo "== $wl $lib"
  python3 - $OUT <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n = r['Name'].replace('void (anonymous namespace)::', '')[:60]
    if 'fps' in n or 'ball_scan' in n: continue
    print(f"  {n:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  rm -rf $OUT/trace
done
