#!/bin/bash
# round 5, call a: the new launch-plan tests, the bench tests, the capture tests, then the default bench line
mkdir -p gpurun_out/r05_a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_plan64.py tests/test_gpu_bench.py tests/test_gpu_generator.py -x -q -m gpu -s > gpurun_out/r05_a/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r05_a/pytest.log
tail -30 gpurun_out/r05_a/pytest.log
timeout 300 python bench.py > gpurun_out/r05_a/bench.json 2> gpurun_out/r05_a/bench.err
echo "bench rc=$?"
head -c 1500 gpurun_out/r05_a/bench.json
