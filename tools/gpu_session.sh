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
    test)   timeout 900 python -m pytest tests -m gpu -x -q --timeout 200 --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;;
    testall) timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt ;;
    bench)  timeout 600 python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; cut -c1-600 $OUT/bench_cfg2.json ; mkdir -p $OUT/profiles; cp $OUT/bench_cfg2.json $OUT/profiles/r6_cfg2_bench.json ;;
    bench3) timeout 900 python bench.py --workload cfg3 --cpu-sample 600 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; cut -c1-400 $OUT/bench_cfg3.json ; mkdir -p $OUT/profiles; cp $OUT/bench_cfg3.json $OUT/profiles/r6_cfg3_bench.json ;;
    bench4) timeout 1500 python bench.py --workload cfg4 --steps 5 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; cut -c1-1500 $OUT/bench_cfg4.json; tail -3 $OUT/bench_cfg4.err ; mkdir -p $OUT/profiles; cp $OUT/bench_cfg4.json $OUT/profiles/r6_cfg4_bench.json ;;
    bench5) timeout 1500 python bench.py --workload cfg5 --cpu-sample 300 --steps 20 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; cut -c1-400 $OUT/bench_cfg5.json ; mkdir -p $OUT/profiles; cp $OUT/bench_cfg5.json $OUT/profiles/r6_cfg5_bench.json ;;
    bench2) timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $OUT/bench_cfg2_2ranks.json 2> $OUT/bench_cfg2_2ranks.err; cut -c1-400 $OUT/bench_cfg2_2ranks.json; tail -3 $OUT/bench_cfg2_2ranks.err ;;
    ranks2:*) # ranks2:<workload>:<shard mode>  two ranks on the ONE GPU (gloo): functional check of bench.py --gpus N
            IFS=: read -r _ wl sh <<< "$s"
            timeout 900 python bench.py --gpus 2 --workload $wl --shard $sh --no-cpu-baseline --steps 5 --warmup 1 > $OUT/ranks2_${wl}_${sh}.json 2> $OUT/ranks2_${wl}_${sh}.err
            python - $OUT/ranks2_${wl}_${sh}.json "$wl $sh" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])   # (gloo prints its own lines on stdout)
    print("2 ranks / 1 GPU", sys.argv[2], d["config"]["parallelism"], "ms/step", d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
            ;;
    prof:*) bash tools/collect_profiles.sh ${s#prof:} > $OUT/collect_${s#prof:}.log 2>&1; tail -3 $OUT/collect_${s#prof:}.log
            python tools/summarize_profiles.py ${s#prof:} r6_${s#prof:} > $OUT/summ_${s#prof:}.log 2>&1; tail -4 $OUT/summ_${s#prof:}.log
            mkdir -p $OUT/profiles && cp profiles/r6_${s#prof:}_kernel_stats.csv profiles/r6_${s#prof:}_pmc.json profiles/r6_${s#prof:}_bench_under_rocprof.json profiles/traffic.json $OUT/profiles/
            rm -rf $R/gpurun_out/prof_${s#prof:} ;;   # (only the summaries travel back: gpurun_out is capped at 64 MiB)
    emul)   bash tools/emulate_scaling.sh cfg2 > $OUT/emulate_cfg2.txt 2>&1; cat $OUT/emulate_cfg2.txt ;;
    emul:*) bash tools/emulate_scaling.sh ${s#emul:} > $OUT/emulate_${s#emul:}.txt 2>&1; cat $OUT/emulate_${s#emul:}.txt ;;
    timers) FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1; export FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so
            timeout 300 python tools/phase_timers.py > $OUT/phase_timers.txt 2>&1
            timeout 300 python tools/chunk_times.py 1 > $OUT/chunk_times_1.txt 2>&1
            timeout 300 python tools/chunk_times.py 8 > $OUT/chunk_times_8.txt 2>&1
            timeout 300 python tools/bvh_phase.py cfg3 > $OUT/bvh_phase_cfg3.txt 2>&1
            unset FLOODER_HIP_LIB
            cat $OUT/phase_timers.txt; head -20 $OUT/chunk_times_1.txt; cat $OUT/bvh_phase_cfg3.txt ;;
    ab:*)   # ab:<workload>:<name>:<flags with , for space>   one bench line per variant
            IFS=: read -r _ wl name flags <<< "$s"
            timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 --no-cold ${flags//,/ } > $OUT/ab_${wl}_${name}.json 2> $OUT/ab_${wl}_${name}.err
            python - "$OUT/ab_${wl}_${name}.json" "$wl $name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    st = d["config"]["sweep_stats_rank0"] or {}
    print(sys.argv[2], "ms/step", d["ms_per_step"], "+-", d["ms_per_step_std"], {k: v["ms_per_step"] for k, v in d["kernels"].items()},
          {k: st.get(k) for k in ("leaves_evaluated_per_tile", "leaves_tested_per_tile", "nodes_expanded_per_tile", "tiles_flagged", "fallback_leaves_evaluated", "fallback_nodes_expanded", "finish_tiles_dropped_on_arrival", "finish_samples_live_on_arrival", "finish_focus_rounds", "exhaustive_rounds", "deferred_chunks", "chunks_total", "finish_shared_rounds")})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
            ;;
    var:*)  # var:<lib in gpurun_in/ without .so | product>:<workload>:<name>[:<flags with , for space>]  bench line of a library variant
            IFS=: read -r _ lib wl name flags <<< "$s"
            if [ "$lib" != product ]; then export FLOODER_HIP_LIB=$R/gpurun_in/$lib.so; fi
            timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 --no-cold ${flags//,/ } > $OUT/var_${wl}_${name}.json 2> $OUT/var_${wl}_${name}.err
            unset FLOODER_HIP_LIB
            python - "$OUT/var_${wl}_${name}.json" "$wl $name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms/step", d["ms_per_step"], "+-", d["ms_per_step_std"], "min", d["ms_per_step_min"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
            ;;
    lds:*)  # LDS counters of the sweep kernels: bank conflicts, LDS-array cycles
            wl=${s#lds:}
            ( cd /tmp && export TMPDIR=/tmp
              timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_lds_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-cold > $OUT/lds_$wl.json 2> $OUT/lds_$wl.err )
            python - $OUT $wl > $OUT/lds_summary_$wl.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{sys.argv[1]}/pmc_lds_{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "anonymous namespace" not in k: continue
        k = k.split("(anonymous namespace)::")[1].split("(")[0]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
last = None
for (k, c) in sorted(acc):
    v, n = acc[(k, c)]
    if k != last: print(k); last = k
    print(f"    {c:32s} {v / n:14.4g} per launch ({n} launches)")
PY
            grep -A8 "cell_sweep\|finish_faces\|sweep_bvh\|sweep_sorted" $OUT/lds_summary_$wl.txt
            rm -rf $OUT/pmc_lds_$wl ;;
    pytest:*) k="${s#pytest:}"; timeout 1200 python -m pytest tests -m gpu -x -q -k "${k//,/ }" > $OUT/pytest_k.txt 2>&1; tail -8 $OUT/pytest_k.txt ;;
    trace:*) # per-dispatch durations of the finish passes (probe, top, rest) from a kernel trace
            wl=${s#trace:}
            ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_$wl.json 2> $OUT/trace_$wl.err )
            python - $OUT/trace_$wl <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fin = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "finish_faces" in r["Kernel_Name"]]
print("finish passes (us), triples probe/top/rest:", [round(x, 1) for x in fin[-9:]])
agg = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    nm = nm.split("(anonymous namespace)::")[1] if "(anonymous namespace)::" in nm else nm
    agg[nm.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k:62s} n={len(v):4d} mean {sum(v)/len(v):9.1f} us")
PY
            rm -rf $OUT/trace_$wl ;;
    timeline:*) # every dispatch of ONE step (back-to-back steps, the last but one): offset, duration, gap to the previous end
            wl=${s#timeline:}
            ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$wl -- python3 $R/bench.py --workload $wl --steps 12 --warmup 3 --no-cpu-baseline --no-cold --no-e2e > $OUT/tl_$wl.json 2> $OUT/tl_$wl.err )
            python - $OUT/tl_$wl > $OUT/timeline_$wl.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    nm = nm.split("(anonymous namespace)::")[1] if "(anonymous namespace)::" in nm else nm
    return nm.split("(")[0][:70]
starts = [i for i, r in enumerate(rows) if "bbox_partial" in r["Kernel_Name"]]
# back-to-back timed steps: the longest run of equal spacing near the end; take the 4th step from the end
i0, i1 = starts[-5], starts[-4]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = None; busy = 0
print(f"one step = {i1 - i0} dispatches, {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us start to start")
for r in rows[i0:i1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += en - st
    print(f"{(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:8.1f}  gap {gap:6.1f}  {short(r['Kernel_Name'])}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
    prev_end = max(prev_end or en, en)
print(f"sum of durations {busy / 1e3:.1f} us")
PY
            rm -rf $OUT/tl_$wl; cat $OUT/timeline_$wl.txt ;;
    ctimes:*) IFS=: read -r _ W wl <<< "$s"      # ctimes:<W>[:<workload>]
            FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1; export FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so
            timeout 400 python tools/chunk_times.py $W ${wl:-cfg2} > $OUT/chunk_times_${W}_${wl:-cfg2}.txt 2>&1
            unset FLOODER_HIP_LIB
            grep -v "^  ends\|kcycles\|^  dur" $OUT/chunk_times_${W}_${wl:-cfg2}.txt ;;
    wends:*) IFS=: read -r _ wl W <<< "$s"
            FLOODER_HIPCC_FLAGS=-DFLOODER_WAVE_END python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_wend.log 2>&1; export FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so
            timeout 300 python tools/wave_ends.py $wl ${W:-1} > $OUT/wave_ends_${wl}_${W:-1}.txt 2>&1
            unset FLOODER_HIP_LIB
            tail -2 $OUT/wave_ends_${wl}_${W:-1}.txt ;;
    fphase:*) wl=${s#fphase:}
            FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1; export FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so
            timeout 300 python tools/bvh_phase.py $wl 2>&1 | grep -v amdgpu.ids > $OUT/fin_phase_$wl.txt
            unset FLOODER_HIP_LIB
            cat $OUT/fin_phase_$wl.txt ;;
    ptimers:*) IFS=: read -r _ wl md <<< "$s"
            FLOODER_HIPCC_FLAGS=-DFLOODER_PHASE_TIMERS python -m flooder_amd.build --out /tmp/libflooder_hip_diag.so > $OUT/build_timers.log 2>&1; export FLOODER_HIP_LIB=/tmp/libflooder_hip_diag.so
            timeout 300 python tools/phase_timers.py $wl ${md:-} > $OUT/phase_timers_$wl.txt 2>&1
            unset FLOODER_HIP_LIB
            cat $OUT/phase_timers_$wl.txt ;;
    sortbench) hipcc -O3 --offload-arch=gfx950 tools/sort_bench.hip -o /tmp/sort_bench > $OUT/sort_build.log 2>&1 && timeout 120 /tmp/sort_bench > $OUT/sort_bench.txt 2>&1; cat $OUT/sort_bench.txt ;;
    cweights:*) timeout 300 python tools/check_weights.py ${s#cweights:} 2>&1 | grep -v amdgpu.ids ;;
    e2e:*) timeout 600 python tools/profile_e2e.py ${s#e2e:} tree 2>&1 | grep -v amdgpu.ids > $OUT/e2e_${s#e2e:}.txt; head -60 $OUT/e2e_${s#e2e:}.txt ;;
    tfps) timeout 600 python tools/time_fps.py > $OUT/time_fps.txt 2>&1; cat $OUT/time_fps.txt ;;
    icache:*) # instruction-cache counters of the sweep kernels (is the 66-83 KB kernel thrashing the 64 KB cache two CUs share?)
            wl=${s#icache:}
            ( cd /tmp && export TMPDIR=/tmp
              timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_icache_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-cold > $OUT/icache_$wl.json 2> $OUT/icache_$wl.err
              timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_inst_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-cold > $OUT/inst_$wl.json 2>> $OUT/icache_$wl.err )
            python - $OUT $wl > $OUT/icache_summary_$wl.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{sys.argv[1]}/pmc_i*_{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "anonymous namespace" not in k: continue
        k = k.split("(anonymous namespace)::")[1].split("(")[0]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
last = None
for (k, c) in sorted(acc):
    v, n = acc[(k, c)]
    if v / n < 1e4: continue
    if k != last: print(k); last = k
    print(f"    {c:32s} {v / n:14.4g} per launch ({n} launches)")
PY
            grep -A17 "cell_sweep\|finish_faces\|sweep_bvh" $OUT/icache_summary_$wl.txt
            rm -rf $OUT/pmc_icache_$wl $OUT/pmc_inst_$wl ;;
    *) echo "unknown step $s" ;;
  esac
done
