#!/bin/bash
# round 6, call k: K = 15 whole-chain test, front-end batch sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_clip.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "K = 15|passed|failed|Error" $O/pytest.log | head
for b in 64 128 256; do timeout 300 python tools/front_bench.py 2048 $b 2>&1 | grep "front end" | tee -a $O/front_bench_sweep.txt; done
