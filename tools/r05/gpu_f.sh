#!/bin/bash
# round 5, call f: image channels 4 .. 6 (generator fixtures, variants, clip + source cache), everything else in test_gpu_generator
mkdir -p gpurun_out/r05_f
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_motion_ops.py tests/test_kp_detector.py -x -q -m gpu -s > gpurun_out/r05_f/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r05_f/pytest.log | cut -c1-300
grep -n "rgba\|six\|five" gpurun_out/r05_f/pytest.log | cut -c1-300
