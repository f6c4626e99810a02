#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r5t
timeout 300 python tools/profile_e2e.py cfg2 tree 2>&1 | grep -v amdgpu.ids > gpurun_out/r5t/e2e_cfg2_tree.txt; head -45 gpurun_out/r5t/e2e_cfg2_tree.txt | cut -c1-150
timeout 300 python tools/profile_e2e.py cfg2 2>&1 | grep -v amdgpu.ids > gpurun_out/r5t/e2e_cfg2_dict.txt; sed -n 1,30p gpurun_out/r5t/e2e_cfg2_dict.txt | cut -c1-150
