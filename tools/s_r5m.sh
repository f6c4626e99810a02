#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r5m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_synthetic.py tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "device or reused or e2e" > $OUT/pytest_k.txt 2>&1; tail -3 $OUT/pytest_k.txt | cut -c1-300
FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1
for wl in cfg3 cfg2; do
FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so timeout 300 python tools/phase_timers.py $wl 2>&1 | grep -v amdgpu.ids > $OUT/phase_timers_$wl.txt; cat $OUT/phase_timers_$wl.txt
FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so timeout 300 python tools/chunk_times.py 1 $wl > $OUT/chunk_times_$wl.txt 2>&1
grep -v "^  ends\|kcycles\|^  dur\|amdgpu.ids" $OUT/chunk_times_$wl.txt | tail -14
done
timeout 300 python bench.py --no-cpu-baseline --no-cold --steps 10 --warmup 2 --extra-workloads none > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5m/bench_e2e.json")); print("e2e", d["e2e"]); print("fps", {k:d["fps"][k] for k in ("ms","ms_index_ready","ms_cold_first_call")})
PY
