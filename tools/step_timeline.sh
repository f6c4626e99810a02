#!/bin/bash
# Diagnostic: start / duration of every kernel of ONE bench step (rocprofv3 --kernel-trace).  usage: tools/step_timeline.sh [workload...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for wl in ${@:-cfg2}; do
OUT=$R/gpurun_out/tr_$wl; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-cold ${EXTRA_BENCH:-} > $OUT/bench.json 2> $OUT/err.txt
python3 - $OUT $wl <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
last = [i for i, n in enumerate(names) if 'bbox_partial' in n][-1]
t0 = int(rows[last]['Start_Timestamp'])
print("==", sys.argv[2])
for r in rows[last:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {r['Kernel_Name'][:90]}")
PY
done
