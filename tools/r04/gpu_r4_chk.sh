#!/bin/bash
for i in 1 2 3; do python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
python bench.py --graph --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('with --graph: eager', d['value'], 'graph', d['graph']['value'])"
