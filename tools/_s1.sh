cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k11
for a in "2000000 6" "300007 5" "1000 8" "1025 4" "17 6" "40 7" "5000 4"; do timeout 200 python tools/kd_check.py $a 2>&1 | grep -v amdgpu.ids | grep kd; done
timeout 1500 python -m pytest tests -m gpu -x -q -k "kd or sorted or fps or bvh or dim or 6d or gauss6d or index or face or fullsize" 2>&1 | tail -5
tools/ab_bench.sh k11 "cfg4" "small6d" 2>&1 | cut -c1-220
