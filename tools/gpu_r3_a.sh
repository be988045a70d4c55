#!/bin/bash
# round-3 GPU call: the GPU test suite, then bench.py per environment setting (first argument = output tag, the rest = settings)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}; shift
O=$R/gpurun_out/r03_$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
if [ $# -gt 0 ]; then BENCH_ARGS="--steps 30" bash tools/exp_env.sh "$@" 2>&1 | tee $O/variants.txt; fi
