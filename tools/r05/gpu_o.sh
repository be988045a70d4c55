#!/bin/bash
# round 5, call o: flakiness check of the threaded / captured launch-plan tests and the pipeline tests, then the bench tests
mkdir -p gpurun_out/r05_o
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_gpu_plan64.py tests/test_gpu_pipeline.py -x -q -m gpu -s > gpurun_out/r05_o/run$i.log 2>&1
  tail -1 gpurun_out/r05_o/run$i.log
  grep -n "^FAILED\|^E  " gpurun_out/r05_o/run$i.log | head -8 | cut -c1-400
done
grep -n "best of 10" gpurun_out/r05_o/run*.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r05_o/bench_tests.log 2>&1; tail -2 gpurun_out/r05_o/bench_tests.log
