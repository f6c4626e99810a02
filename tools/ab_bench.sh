#!/bin/bash
# one-line bench summaries: tools/ab_bench.sh <tag> "<workload> <bench flags>" ...
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
i=0
for spec in "$@"; do
  i=$((i+1)); set -- $spec; wl=$1; shift
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 --no-cold "$@" > $OUT/ab_$i.json 2> $OUT/ab_$i.err
  python - $OUT/ab_$i.json "$spec" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); st = d["config"]["sweep_stats_rank0"] or {}
    w = st.get("witness") or {}
    print(sys.argv[2], "| ms/step", d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()},
          "| wit handled", w.get("simplices_handled"), "dense/abandoned", w.get("too_dense"), "live", w.get("samples_live_after_bound"), "open", w.get("samples_open_after_stage"),
          "wit tiles", w.get("tiles_flagged"), "focus", w.get("focus_rounds"), "| cell tiles", st.get("tiles_flagged"), "| parity", (d.get("parity") or {}).get("max_rel_err"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
