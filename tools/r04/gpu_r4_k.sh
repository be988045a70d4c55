#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_k; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/pytest_gpu.log | tail -14
