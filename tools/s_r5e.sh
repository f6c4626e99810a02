#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5e
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5e/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5e/pytest_gpu.txt
/usr/bin/time -v python bench.py > gpurun_out/r5e/bench_default.json 2> gpurun_out/r5e/bench_default.err; grep "Elapsed" gpurun_out/r5e/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5e/bench_default.json") if l.startswith("{")][-1])
print("cfg2", d["ms_per_step"], d["roofline"]["frac"], d["parity"], d["e2e"])
for k,v in d.get("extra_workloads",{}).items(): print(k, {a:v.get(a) for a in ("ms_per_step","roofline","parity","wall_s","error")})
PY
