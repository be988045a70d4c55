#!/bin/bash
# final state of the round: smoke, the whole GPU suite, the default bench line
O=gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json
python -c "
import json; d=json.load(open('$O/bench_256_b16.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['clip']['frames_per_s'], d['cpu_baseline']['value'], d['train_step']['step_ms'])"
