#!/bin/bash
# Diagnostic PMC passes over the cell sweep (run on the GPU box from the repo root): what does the memory path wait on?
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT
i=0
# (a pass with the TA_* counters hung the profiler on this pool: left out; every pass has its own timeout)
for set in \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b$i.json 2> $OUT/e$i.txt
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_probe/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cell_sweep' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(acc):
    print(f"{k:45s} {acc[k][0] / acc[k][1]:.4g}  (per launch, {acc[k][1]} launches)")
PY
