#!/bin/bash
# round 4, call c: final layer on the 4x4x1 multi-block MFMA -- parity, kernel timing with diagnostics, pipeline effect
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator.py -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -15
for m in 0 1; do for d in 0 1 2 3; do EAMM_FINAL_MFMA4=$m EAMM_COL7_DBG=$d timeout 120 python tools/final_layer_bench.py 8 16 2>&1 | grep -v amdgpu.ids; done; done | tee $O/final_layer_bench.txt
timeout 300 bash tools/exp_env.sh "EAMM_FINAL_MFMA4=0" "EAMM_FINAL_MFMA4=1" "EAMM_FINAL_MFMA4=0" "EAMM_FINAL_MFMA4=1" 2>&1 | tee $O/exp_final.txt
