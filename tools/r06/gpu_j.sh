#!/bin/bash
# round 6, call j: full GPU suite on the current tree (num_kp variants in training / clip tests), default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_j; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
( time timeout 600 python bench.py > $O/bench_256_b16.log 2>&1 ) 2>&1 | grep real; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_j/bench_256_b16.json'))
print(d['value'], d['roofline']['frac'], d['all_outputs']['frames_per_s'], d['latency_b1']['ms_per_frame'], d['clip']['frames_per_s'], d['e2e_clip']['frames_per_s'], d['e2e_clip']['phases_ms_rank0']['front_ms'], d['train_step']['step_ms'], d['cpu_baseline']['value'])
P
