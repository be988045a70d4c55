#!/bin/bash
# Run bench.py (no CPU baseline, no clip leg) once per environment setting and print value + stage split.
#   tools/exp_env.sh "EAMM_WINO4_VARIANT=0" "EAMM_WINO4_VARIANT=3" ...        (extra bench args in $BENCH_ARGS)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for kv in "$@"; do
  env EAMM_TUNING=1 $kv python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$kv', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
