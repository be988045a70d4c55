#!/bin/bash
# round 4, call a: GPU tests, the contract line, the stagger experiment, the 256x256 profile (kernel trace + timeline + PMC)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export ROUND=r04
O=$R/gpurun_out/r04_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; cut -c1-300 $O/bench_256_b16.json
timeout 300 bash tools/exp_env.sh "EAMM_BNECK_STAGGER=0" "EAMM_BNECK_STAGGER=1" "EAMM_BNECK_STAGGER=0" "EAMM_BNECK_STAGGER=1" > $O/exp_stagger.txt 2>&1; cat $O/exp_stagger.txt
timeout 900 bash tools/gpu_profile.sh 256 16 r04a_256_b16 > $O/profile_256.log 2>&1; tail -12 $O/profile_256.log
cat $R/gpurun_out/prof_r04a_256_b16/bneck_timeline.txt | tail -8
