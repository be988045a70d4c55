#!/bin/bash
# round 4, call d: packed subtraction in the up-block kernel, DMA placement in the 4x4x1 final kernel, graph replay of the one-frame pass
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -5
for e in 8 4; do EAMM_COL7Q_DMA_EVERY=$e timeout 120 python tools/final_layer_bench.py 8 16 2>&1 | grep -v amdgpu.ids | sed "s/^/every=$e /"; done | tee $O/final_layer_bench.txt
timeout 400 bash tools/exp_env.sh "EAMM_COL7Q_DMA_EVERY=8" "EAMM_COL7Q_DMA_EVERY=4" "EAMM_COL7Q_DMA_EVERY=8" "EAMM_COL7Q_DMA_EVERY=4" 2>&1 | tee $O/exp_every.txt
timeout 300 python tools/conv_bench.py 8 up0,up1 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_up.txt
timeout 400 python tools/module_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/module_latency.txt
