#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5h
bash tools/ab_bench.sh r5h/ab "cfg3 --option finish_flip=0" "cfg3" "cfg3 --option finish_flip=2" "cfg3 --option finish_flip=8" "cfg5 --option finish_flip=0" "cfg5" "cfg5 --option finish_flip=8" "cfg2 --option finish_flip=0" "cfg2" 2>&1 | cut -c1-330
SECONDS=0
timeout 900 python bench.py > gpurun_out/r5h/bench_default.json 2> gpurun_out/r5h/bench_default.err; echo "default bench wall: $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5h/bench_default.json") if l.startswith("{")][-1])
print("cfg2", d["ms_per_step"], d["roofline"]["frac"], d["parity"], d["e2e"])
for k,v in d.get("extra_workloads",{}).items(): print(k, {a:v.get(a) for a in ("ms_per_step","roofline","parity","wall_s","error")})
PY
