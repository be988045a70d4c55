#!/bin/bash
# round 6, call l: end-to-end leg by frames per front-end call
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_l; mkdir -p $O; cd $R
for fb in 64 128 256 128 256; do
timeout 600 python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 --latency-frames 0 --no-all-outputs --e2e-front-batch $fb > $O/b.log 2>&1; grep '^{' $O/b.log > $O/b_$fb.json
python - <<P
import json
d=json.load(open('gpurun_out/r06_l/b_$fb.json'))
print('front_batch $fb:', d['value'], 'e2e', d['e2e_clip']['frames_per_s'], d['e2e_clip']['phases_ms_rank0'])
P
done 2>&1 | tee $O/e2e_front_batch_sweep.txt
