"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
   profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (average duration per kernel)
   profiles/<name>_pmc.json           mean counter values per launch and kernel (separate --pmc passes)
   profiles/<name>_bench_under_rocprof.json   the bench line of the traced run
   profiles/traffic.json              what bench.py quotes: HBM bytes per launch (FETCH_SIZE x 2 for 16 B/lane loads,
                                      as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE) and the VALU issue
                                      utilisation = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz),
                                      tagged with the fingerprint of the kernel sources they were measured with."""
import csv, glob, collections, json, os, shutil, sys

sys.path.insert(0, ".")
from bench import kernel_source_sha

tag = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
name = sys.argv[2] if len(sys.argv) > 2 else f"r6_{tag}"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
FROM_PMC = "--from-pmc" in sys.argv   # recompute traffic.json from the committed summaries (no raw profiler output)
avg_ns = {}
out = {}
if FROM_PMC:
    out = json.load(open(f"profiles/{name}_pmc.json"))
else:
    stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
    shutil.copy(stats, f"profiles/{name}_kernel_stats.csv")
    for row in csv.DictReader(open(stats)):
        avg_ns[row["Name"]] = float(row["AverageNs"])
for sub in (() if FROM_PMC else ("pmc_fetch", "pmc_write", "pmc_sq")):
    files = glob.glob(f"{src}/{sub}/*/*_counter_collection.csv")
    if not files:
        continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        agg[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in agg.items():
        if "anonymous namespace" in k:
            key = k.split("(anonymous namespace)::")[1].split("(")[0]
            out.setdefault(key, {})[c + "_mean_per_launch"] = round(sum(v) / len(v), 1)
            out[key]["launches_profiled"] = len(v)
for k, ns in avg_ns.items():
    if "anonymous namespace" in k:
        key = k.split("(anonymous namespace)::")[1].split("(")[0]
        if key in out:
            out[key]["avg_duration_us_kernel_trace"] = round(ns / 1e3, 2)
if not FROM_PMC:
    json.dump(out, open(f"profiles/{name}_pmc.json", "w"), indent=1)
traffic = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
sha = kernel_source_sha()
# span of bench.py -> the kernels launched under it (the cell sweep is two launches: runs of four chunks, then chunk by chunk)
# (method, span, ((kernel, launches per step), ...))
# (round 4: the witness sweep - its list kernel and the sweep itself - runs ahead of the two cell-sweep launches)
SPANS = (("cell", "sweep", (("wit_sweep_kernel<3>", 1), ("wit_list_kernel", 1), ("cell_sweep_kernel<3, true, 4>", 1),
                            ("cell_sweep_kernel<3, false, 4>", 1))),
         # the pass over the flagged tiles, its hard tiles (one workgroup each), the ordering of the flagged tiles:
         # one launch each per step (the "top pass" that doubled the first two is off by default since round 2)
         # (round 6: a third template argument - waves per workgroup: 8 on clouds of 4 M points and more)
         ("cell", "fallback", (("finish_faces_kernel<3, false>", 1), ("finish_faces_kernel<3, true>", 1),
                               ("finish_faces_kernel<3, false, 4>", 1), ("finish_faces_kernel<3, false, 8>", 1),
                               ("finish_faces_kernel<3, true, 4>", 1), ("order_flags_kernel", 1))),
         ("bvh", "sweep_bvh", (("sweep_bvh_kernel<3, 2, 1>", 1),)),
         # tree sweep over sorted samples: keys, the radix sort's launches (rocprim kernels are not listed by name:
         # their share is in kernel_stats.csv), the sweep
         ("bvh", "sweep_bvh", (("sweep_sorted_kernel<6, 1, false>", 1), ("sample_keys_kernel<6>", 1))),
         ("ball", "sweep_ball", (("sweep_kernel<3, true>", 1),)))
for method, span, kerns, in SPANS:
    have = [(k, m) for k, m in kerns if k in out and "FETCH_SIZE_mean_per_launch" in out[k]]
    if not have:
        continue
    f = sum(out[k]["FETCH_SIZE_mean_per_launch"] * m for k, m in have)
    w = sum(out[k].get("WRITE_SIZE_mean_per_launch", 0.0) * m for k, m in have)
    ent = {"kernels": [k for k, _ in have], "launches_per_step": sum(m for _, m in have), "fetch_size_kb": round(f, 1),
           "write_size_kb": round(w, 1), "fetch_correction": 2.0, "bytes_per_launch": int((2.0 * f + w) * 1024),
           "bytes_note": "HBM bytes per step of this span: sum over its launches of 2 x FETCH_SIZE + WRITE_SIZE",
           "source": f"profiles/{name}_pmc.json", "kernel_src_sha": sha}
    us = sum(out[k].get("avg_duration_us_kernel_trace", 0.0) * m for k, m in have)
    valu = sum(out[k].get("SQ_INSTS_VALU_mean_per_launch", 0.0) * m for k, m in have)
    if us and valu:
        ent["issue_util"] = round(valu * 2.0 / (1024 * us * 1e-6 * 2.4e9), 4)
        ent["valu_wave_instructions"] = valu
        ent["duration_us_kernel_trace"] = round(us, 1)
        ent["issue_util_note"] = "SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x duration x 2.4 GHz)"
    traffic[f"{tag}:{method}:{span}"] = ent
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
bench = f"{src}/bench_trace.json"
if os.path.exists(bench):
    shutil.copy(bench, f"profiles/{name}_bench_under_rocprof.json")
print(json.dumps({k: v for k, v in out.items() if "sweep" in k or "face" in k}, indent=1)); print(traffic)
