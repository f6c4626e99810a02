cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k14
timeout 1500 python -m pytest tests -m gpu -x -q -k "kd or sorted or fps or bvh or dim or 6d or gauss6d or index or face or fullsize or unfused" 2>&1 | tail -2
tools/ab_bench.sh k14 "cfg4" "small6d" 2>&1 | cut -c1-120
