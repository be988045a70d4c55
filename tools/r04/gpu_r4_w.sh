#!/bin/bash
# clip leg at 64 frames per call: two chains vs the automatic four (side streams from the device's shared pool), alternating on one box
O=gpurun_out/r04_w; mkdir -p $O
for i in 1 2; do
for pc in 2 0; do
EAMM_PASS_CHAINS=$pc python bench.py --cpu-frames 0 --train-pairs 0 --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('EAMM_PASS_CHAINS=$pc value', d['value'], 'clip', d['clip']['frames_per_s'], d['clip']['phases_ms_rank0'])" | tee -a $O/clip2.txt
done; done
python -m pytest tests/test_gpu_clip.py tests/test_gpu_generator.py -q -x -m gpu 2>&1 | tail -3
