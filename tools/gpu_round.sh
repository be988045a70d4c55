#!/bin/bash
# One gpurun call: GPU tests, default bench line, 512x512 bench line, module latency, profiles at both BASELINE sizes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh <tag>'      (ROUND=r04 by default: output names carry it)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
ROUND=${ROUND:-r06}
O=$R/gpurun_out/${ROUND}_$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; cut -c1-600 $O/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > $O/bench_512_b8.log 2>&1; grep '^{' $O/bench_512_b8.log > $O/bench_512_b8.json; cut -c1-400 $O/bench_512_b8.json
timeout 300 python tools/module_latency.py > $O/module_latency.txt 2>&1; cat $O/module_latency.txt
timeout 300 python tools/bn_bench.py 2>&1 < /dev/null | grep -v amdgpu.ids > $O/bn_bench.txt; cat $O/bn_bench.txt
timeout 300 python tools/backward_bench.py 16 2>&1 < /dev/null | grep -v amdgpu.ids > $O/backward_bench.txt; cut -c1-260 $O/backward_bench.txt
timeout 300 python tools/train_step_bench.py 8 5 2>&1 < /dev/null | grep -v amdgpu.ids > $O/train_step.txt; cat $O/train_step.txt
if [ "${PROFILE:-1}" = "1" ]; then
  timeout 900 bash tools/gpu_profile.sh 256 16 ${ROUND}${TAG}_256_b16 > $O/profile_256.log 2>&1; tail -25 $O/profile_256.log
  timeout 900 bash tools/gpu_profile.sh 512 8 ${ROUND}${TAG}_512_b8 > $O/profile_512.log 2>&1; tail -25 $O/profile_512.log
fi
