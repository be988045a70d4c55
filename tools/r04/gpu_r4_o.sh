#!/bin/bash
# round 4: more bottleneck chains (shorter input transforms per chain) with more hardware queues
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_o; mkdir -p $O
run() { env "$@" python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d['stage_ms_per_step'].items()}, 'chains', d['roofline']['chains'], d['roofline']['pass_chains'], 'frac', d['roofline']['frac'])"; }
run EAMM_X=0
run EAMM_PASS_CHAINS=1 EAMM_BNECK_CHAINS=2
run EAMM_PASS_CHAINS=1 EAMM_BNECK_CHAINS=4
run EAMM_PASS_CHAINS=1 EAMM_BNECK_CHAINS=4 GPU_MAX_HW_QUEUES=8
run EAMM_PASS_CHAINS=1 EAMM_BNECK_CHAINS=8 GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8
run EAMM_PASS_CHAINS=4 GPU_MAX_HW_QUEUES=8
