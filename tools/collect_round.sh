#!/bin/bash
# One gpurun call that produces the evidence a round commits under profiles/ (run on the GPU box from the repo root):
#   tools/collect_round.sh r5   ->   gpurun_out/final_r5/profiles/r5_*  (copy into profiles/ afterwards)
# GPU tests, the default bench line (cfg 2 + extra_workloads), full bench lines of cfg 3 / 4 / 5, the rocprofv3 passes
# of cfg 2 / 3 / 5 / 4 (kernel trace + FETCH_SIZE / WRITE_SIZE / SQ counters, separate passes), emulated rank shares,
# landmark-selection times, the reference's end-to-end protocol.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
TAG=${1:-r6}
bash tools/gpu_session.sh final_$TAG test bench bench3 bench5 bench4 prof:cfg2 prof:cfg3 prof:cfg5 prof:cfg4 emul:cfg2 emul:cfg3 emul:cfg5 emul:cfg4 tfps
OUT=$R/gpurun_out/final_$TAG
cp $OUT/pytest_gpu.txt $OUT/profiles/${TAG}_pytest_gpu.txt
for wl in cfg2 cfg3 cfg4 cfg5; do cp $OUT/emulate_$wl.txt $OUT/profiles/${TAG}_${wl}_emulated_shards.txt; done
cp $OUT/time_fps.txt $OUT/profiles/${TAG}_time_fps.txt
timeout 600 python examples/flood_ph_timing.py cheese --sizes 1000000 --reps 5 > $OUT/profiles/${TAG}_e2e_cheese_1m_reference_protocol.txt 2>&1
tail -5 $OUT/profiles/${TAG}_e2e_cheese_1m_reference_protocol.txt
ls $OUT/profiles
