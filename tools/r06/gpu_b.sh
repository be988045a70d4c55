#!/bin/bash
# round 6, call b: the new parity cases (num_kp 1/5/15/30, bit-exact One-Euro, pinned-pool aliasing) on the GPU
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_kp_detector.py tests/test_gpu_pipeline.py tests/test_gpu_clip.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_new.log
grep -E "^(tiny64_kp|full256_kp|kp_tiny64_k|kpa_tiny_k|one-euro)|passed|failed|Error|FAILED" $O/pytest_new.log | head -60
