"""Turn gpurun_out/prof_final_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import csv, glob, collections, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
name = sys.argv[2] if len(sys.argv) > 2 else f"r1_final_{tag}"
src = f"gpurun_out/prof_final_{tag}"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(f"{src}/trace/runc/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{name}_kernel_stats.csv")
out = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(f"{src}/{sub}/runc/*_counter_collection.csv")
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
json.dump(out, open(f"profiles/{name}_pmc.json", "w"), indent=1)
# traffic table used by bench.py: HBM bytes per launch of the dominant kernel, corrected as
# MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE (KB) under-reports wide 16 B/lane loads by 2x,
# WRITE_SIZE (KB) is exact
traffic = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
for kern, method in (("cell_sweep_kernel<3>", "cell"), ("sweep_bvh_kernel<3, 8>", "bvh"), ("sweep_kernel<3, true>", "ball")):
    if kern in out and "FETCH_SIZE_mean_per_launch" in out[kern]:
        f, w = out[kern]["FETCH_SIZE_mean_per_launch"], out[kern].get("WRITE_SIZE_mean_per_launch", 0.0)
        traffic[f"{tag}:{method}"] = {"kernel": kern, "fetch_size_kb": f, "write_size_kb": w, "fetch_correction": 2.0,
                                      "bytes_per_launch": int((2.0 * f + w) * 1024),
                                      "source": f"profiles/{name}_pmc.json"}
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
bench = f"{src}/bench_trace.json"
if os.path.exists(bench):
    shutil.copy(bench, f"profiles/{name}_bench_under_rocprof.json")
print(json.dumps({k: v for k, v in out.items() if "sweep" in k or "face" in k}, indent=1)); print(traffic)
