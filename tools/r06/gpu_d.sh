#!/bin/bash
# round 6, call d: key-point heads on the LDS-DMA tile (7x7): parity + the end-to-end leg's front-end phase
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kp_detector.py tests/test_gpu_pipeline.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^batched heads|passed|failed|Error|FAILED|assert" $O/pytest.log | head -40
timeout 300 python tools/kp_bench.py 64 > $O/kp_bench.txt 2>&1; tail -12 $O/kp_bench.txt
for v in 16384 -1; do
EAMM_TUNING=1 EAMM_KP_HEAD_DMA_MIN_M=$v timeout 600 python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 > $O/bench_e2e_$v.log 2>&1; grep '^{' $O/bench_e2e_$v.log > $O/bench_e2e_$v.json
python - <<P
import json
d=json.load(open('gpurun_out/r06_d/bench_e2e_$v.json'))
print('$v', d['value'], d['e2e_clip']['frames_per_s'], d['e2e_clip']['phases_ms_rank0'])
P
done
