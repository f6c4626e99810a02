#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box from the repo root):
#   1. --kernel-trace --stats  : per-kernel average durations (must agree with bench.py's HIP-event figure)
#   2. --pmc FETCH_SIZE        : HBM read traffic  (own pass; TCC counters do not fit together)
#   3. --pmc WRITE_SIZE        : HBM write traffic (own pass)
#   4. --pmc SQ_*              : instruction / wave-cycle counters (issue utilisation)
# Output: gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq}; tools/summarize_profiles.py turns it into profiles/.
# (python3 directly after "--": the profiler must not see an exec hop.)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-cfg2}
STEPS=${2:-10}
EXTRA_BENCH=${EXTRA_BENCH:-}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $TAG --steps $STEPS --warmup 2 --no-cpu-baseline --no-cold --no-e2e $EXTRA_BENCH > $OUT/bench_trace.json 2> $OUT/err_trace.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --workload $TAG --steps 3 --warmup 1 --no-cpu-baseline --no-cold --no-e2e $EXTRA_BENCH > $OUT/bench_fetch.json 2> $OUT/err_fetch.txt
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --workload $TAG --steps 3 --warmup 1 --no-cpu-baseline --no-cold --no-e2e $EXTRA_BENCH > $OUT/bench_write.json 2> $OUT/err_write.txt
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 bench.py --workload $TAG --steps 3 --warmup 1 --no-cpu-baseline --no-cold --no-e2e $EXTRA_BENCH > $OUT/bench_sq.json 2> $OUT/err_sq.txt
# keep only what summarize_profiles.py reads (gpurun_out is capped at 64 MiB)
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls $OUT/*/*/ 2>/dev/null | head -20
