#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_g; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_backward.py -m gpu -q -s > $O/pytest_train.log 2>&1; echo "pytest train rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|worst|^E  " $O/pytest_train.log | head -40
