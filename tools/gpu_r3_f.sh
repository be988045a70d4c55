#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03_f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fused_final or col7" 2>&1 | tail -12 | tee $O/ops.log
timeout 600 python -m pytest tests/test_gpu_generator.py -q -x 2>&1 | tail -5 | tee $O/gen.log
BENCH_ARGS="--steps 30" bash tools/exp_env.sh "EAMM_FINAL_FUSED=1" "EAMM_FINAL_FUSED=0" "EAMM_FINAL_FUSED=1" "EAMM_FINAL_FUSED=0" 2>&1 | tee $O/variants.txt
BENCH_ARGS="--size 512 --steps 10" bash tools/exp_env.sh "EAMM_FINAL_FUSED=1" "EAMM_FINAL_FUSED=0" 2>&1 | tee $O/variants512.txt
BENCH_ARGS="--batch 1 --steps 50" bash tools/exp_env.sh "EAMM_FINAL_FUSED=1" "EAMM_FINAL_FUSED=0" 2>&1 | tee $O/variants_b1.txt
