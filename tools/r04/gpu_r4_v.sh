#!/bin/bash
# default bench line with the clip leg at 64 frames per call + bench tests
O=gpurun_out/r04_v; mkdir -p $O
( time python bench.py ) > $O/bench.log 2>&1
grep '^{' $O/bench.log > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_v/bench.json'))
print(d['value'], d['ms_per_step'], d['clip'], d['train_step']['step_ms'])
PY
tail -4 $O/bench.log
python -m pytest tests/test_gpu_bench.py -q -x -m gpu 2>&1 | tail -3
