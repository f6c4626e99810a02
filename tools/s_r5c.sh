#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export OUTTAG=r5c
run() { # name lib workload flags...
  name=$1; lib=$2; shift 2
  if [ "$lib" != product ]; then export FLOODER_HIP_LIB=$R/gpurun_in/$lib.so; else unset FLOODER_HIP_LIB; fi
  bash tools/ab_bench.sh $OUTTAG/$name "$*" 2>&1 | sed "s/^/[$name] /"
}
run p5 product cfg5
run w4_5 w4 cfg5
run w4g_5 w4 cfg5 --option cell_grid=1024
run p3 product cfg3
run w4_3 w4 cfg3
run w4g_3 w4 cfg3 --option cell_grid=1024
run p2 product cfg2
run w4_2 w4 cfg2
run w4g_2 w4 cfg2 --option cell_grid=1024
