#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r5b; mkdir -p $OUT
FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1
FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so timeout 300 python tools/phase_timers.py cfg5 > $OUT/phase_timers_cfg5.txt 2>&1
cat $OUT/phase_timers_cfg5.txt | grep -v amdgpu.ids
FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so timeout 300 python tools/chunk_times.py 1 cfg5 > $OUT/chunk_times_cfg5.txt 2>&1
grep -v "^  ends\|kcycles\|^  dur\|amdgpu.ids" $OUT/chunk_times_cfg5.txt | head -60
cd /tmp && export TMPDIR=/tmp
for qb in -1 5; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$qb -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-cold --no-e2e --option cell_queue_block=$qb > $OUT/bench_fetch_$qb.json 2> $OUT/err_fetch_$qb.txt
  python3 - $OUT/pmc_fetch_$qb $qb <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cell_sweep" not in k and "finish_faces" not in k: continue
        k = k.split("(anonymous namespace)::")[1].split("(")[0][:50]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("qblk", sys.argv[2], k, c, f"{v/n:.4g} per launch ({n} launches)  x2 x1KB = {v/n*2*1024/1e9:.2f} GB")
PY
  rm -rf $OUT/pmc_fetch_$qb
done
