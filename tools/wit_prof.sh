#!/bin/bash
# per-kernel durations of the sweep with the witness sweep on: tools/wit_prof.sh <tag> <workload> [option=value ...]
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; WL=$2; shift 2
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python3 $R/tools/wit_check.py $WL 10 "$@" > $OUT/prof_$WL.txt 2>&1
cd $R
python3 - $OUT/trace_$WL <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"]
    nm = nm.split("(anonymous namespace)::")[1] if "(anonymous namespace)::" in nm else nm
    agg[nm.split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
    v2 = sorted(v)
    print(f"{k:72s} n={len(v):4d} mean {sum(v)/len(v):9.1f} us  median {v2[len(v2)//2]:9.1f}  max {v2[-1]:9.1f}")
PY
rm -rf $OUT/trace_$WL
