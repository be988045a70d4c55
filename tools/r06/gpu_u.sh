#!/bin/bash
# round 6, call u: 512x512 x 8 by number of whole-pass chains (a 2-frame chain's GEMM is 128 workgroups, like the 8-frame chain's at 256x256)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_u; mkdir -p $O; cd $R
for pc in 2 4 3 2 4; do
EAMM_PASS_CHAINS=$pc timeout 300 python bench.py --size 512 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 --latency-frames 0 --no-all-outputs > $O/b.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r06_u/b.log') if x.startswith('{')]
d=json.loads(l[0]); print('EAMM_PASS_CHAINS=$pc', d['value'], d['roofline']['frac'], d['knobs']['plan']['pass_chains'], d['knobs']['plan']['frames_per_chain'], d['parity_check']['max_abs_err'])
P
done 2>&1 | tee $O/pass_chains_512.txt
