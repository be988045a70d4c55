#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_q; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kp_detector.py -m gpu -q -x 2>&1 | tail -2
for cb in 128 64 32; do timeout 300 python tools/r06/dual_handle.py $cb 2>&1 | grep -v amdgpu.ids | tee -a $O/dual_handle.txt; done
