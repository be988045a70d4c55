#!/bin/bash
# the 16-frame step eager vs replayed as a HIP graph, unprofiled
O=gpurun_out/r04_g; mkdir -p $O
for i in 1 2; do
python bench.py --graph --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'], 'graph', d['graph']['value'], d['graph']['ms_per_step'])" | tee -a $O/graph.txt
done
python bench.py --graph --batch 64 --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch 64: eager', d['value'], d['ms_per_step'], 'graph', d['graph']['value'], d['graph']['ms_per_step'])" | tee -a $O/graph.txt
