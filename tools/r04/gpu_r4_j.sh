#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_j; mkdir -p $O
timeout 400 bash tools/exp_env.sh "EAMM_WINO4_EPI_V=0" "EAMM_WINO4_EPI_V=1" "EAMM_WINO4_EPI_V=0" "EAMM_WINO4_EPI_V=1" 2>&1 | tee $O/exp_epi_v.txt
EAMM_WINO4_EPI_V=1 timeout 200 python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('EPI_V=1', d['value'], 'frac', r['frac'], 'union', r['bneck_union_ms_per_step'], 'per_launch ms', r['per_launch']['avg_launch_ms'], 'transform ms', r['avg_input_transform_ms'])" | tee -a $O/exp_epi_v.txt
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -s -k "gen_" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|conv backward gen" $O/pytest_conv.log | tail -6
