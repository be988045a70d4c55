#!/bin/bash
# round 4, call e: joint warp launch (one feature-warp launch for both chains' frames)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_e; mkdir -p $O
EAMM_WARP_JOINT=1 timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_clip.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -5
timeout 400 bash tools/exp_env.sh "EAMM_WARP_JOINT=0" "EAMM_WARP_JOINT=1" "EAMM_WARP_JOINT=0" "EAMM_WARP_JOINT=1" 2>&1 | tee $O/exp_joint.txt
EAMM_WARP_JOINT=1 timeout 300 python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['roofline_warp'])"
