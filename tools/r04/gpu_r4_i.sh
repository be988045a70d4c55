#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_i; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -8
timeout 400 python tools/module_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/module_latency.txt
timeout 300 python tools/bn_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bn_bench_tail.txt
