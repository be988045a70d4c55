#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_x; mkdir -p $O; cd $R
python tools/r06/e2e_d2h.py 2>&1 | grep -v amdgpu.ids | tee $O/e2e_d2h.txt
HSA_ENABLE_SDMA=1 python tools/r06/e2e_d2h.py 2>&1 | grep -v amdgpu.ids | tee -a $O/e2e_d2h.txt
HSA_ENABLE_SDMA=0 python tools/r06/e2e_d2h.py 2>&1 | grep -v amdgpu.ids | tee -a $O/e2e_d2h.txt
