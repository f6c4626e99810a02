#!/bin/bash
# One gpurun call: tests, bench lines, profiles, diagnostics.  usage: tools/gpu_session.sh <tag> [steps...]
# steps: test bench bench2 prof:<wl> timers   (default: test bench)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=$1; shift
STEPS=${@:-test bench}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for s in $STEPS; do
  case $s in
    test)   timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;;
    testall) timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt ;;
    bench)  timeout 600 python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; cut -c1-600 $OUT/bench_cfg2.json ;;
    bench3) timeout 600 python bench.py --workload cfg3 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; cut -c1-400 $OUT/bench_cfg3.json ;;
    bench5) timeout 900 python bench.py --workload cfg5 --no-cpu-baseline --steps 20 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; cut -c1-400 $OUT/bench_cfg5.json ;;
    bench2) timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $OUT/bench_cfg2_2ranks.json 2> $OUT/bench_cfg2_2ranks.err; cut -c1-400 $OUT/bench_cfg2_2ranks.json; tail -3 $OUT/bench_cfg2_2ranks.err ;;
    prof:*) bash tools/collect_profiles.sh ${s#prof:} > $OUT/collect_${s#prof:}.log 2>&1; tail -5 $OUT/collect_${s#prof:}.log ;;
    emul)   bash tools/emulate_scaling.sh cfg2 > $OUT/emulate_cfg2.txt 2>&1; cat $OUT/emulate_cfg2.txt ;;
    timers) cp flooder_amd/libflooder_hip.so /tmp/libflooder_hip.so.keep
            FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --force > $OUT/build_timers.log 2>&1
            timeout 300 python tools/phase_timers.py > $OUT/phase_timers.txt 2>&1
            timeout 300 python tools/chunk_times.py 1 > $OUT/chunk_times_1.txt 2>&1
            timeout 300 python tools/chunk_times.py 8 > $OUT/chunk_times_8.txt 2>&1
            timeout 300 python tools/bvh_phase.py cfg3 > $OUT/bvh_phase_cfg3.txt 2>&1
            cp /tmp/libflooder_hip.so.keep flooder_amd/libflooder_hip.so
            cat $OUT/phase_timers.txt; head -20 $OUT/chunk_times_1.txt; cat $OUT/bvh_phase_cfg3.txt ;;
    *) echo "unknown step $s" ;;
  esac
done
