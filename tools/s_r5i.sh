#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/r5i
bash tools/ab_bench.sh r5i/ab "cfg5 --option cell_brute_max=224" "cfg5 --option cell_brute_max=300" "cfg5 --alpha 1.2" "cfg5 --alpha 1.5" "cfg3 --option cell_brute_max=224" "cfg3 --option cell_brute_max=300" "cfg3 --alpha 1.2" "cfg3 --alpha 1.5" "cfg2 --option cell_brute_max=224" "cfg2 --option cell_brute_max=300" "cfg2 --alpha 1.2" "cfg2 --alpha 1.5" 2>&1 | cut -c1-200
