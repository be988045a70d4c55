#!/bin/bash
# round 4, call f: HIP backward of the dense-motion front end / flow head -- operator tests, the generator-level gradient fixtures
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_motion_ops.py -m gpu -q -s > $O/pytest_motion.log 2>&1; echo "pytest motion rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|Error|assert" $O/pytest_motion.log | head -30
timeout 1200 python -m pytest tests/test_train_backward.py tests/test_train_mode.py -m gpu -q -s > $O/pytest_train.log 2>&1; echo "pytest train rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|worst error" $O/pytest_train.log | head -30
timeout 300 python tools/train_step_bench.py 8 5 2>&1 | grep -v amdgpu.ids | tee $O/train_step.txt
