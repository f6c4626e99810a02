#!/bin/bash
# rocprofv3 evidence for the brute-force FPS sweep (the bandwidth-bound kernel of the path): kernel trace + HBM counters
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-fpsprof}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "16000000 3 128" "1000000 3 128"; do
  set -- $spec; tag=${1}_${2}d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $R/tools/fps_brute.py $spec > $OUT/run_$tag.txt 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/fps_brute.py $spec > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$tag -- python3 $R/tools/fps_brute.py $spec > /dev/null 2>&1
  grep "brute FPS" $OUT/run_$tag.txt
  python3 - $OUT $tag <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
res = {}
f = glob.glob(f"{out}/trace_{tag}/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "fps" in r["Name"]:
            res.setdefault("kernels", []).append({"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6})
for kind in ("fetch", "write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{kind}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fps" in r["Kernel_Name"]:
                a = acc[r["Kernel_Name"][:60]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    res[kind] = {k: {"mean_counter_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for k, v in acc.items()}
json.dump(res, open(f"{out}/fps_brute_{tag}.json", "w"), indent=1)
print(json.dumps(res)[:900])
PY
  rm -rf $OUT/trace_$tag $OUT/fetch_$tag $OUT/write_$tag
done
