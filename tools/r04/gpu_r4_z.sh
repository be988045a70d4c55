#!/bin/bash
# the 512x512 bench line (clip leg at 16 frames per call)
O=gpurun_out/r04_y; mkdir -p $O
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > $O/bench_512_b8.log 2>&1; grep '^{' $O/bench_512_b8.log > $O/bench_512_b8.json; cut -c1-300 $O/bench_512_b8.json
python -c "
import json; d=json.load(open('$O/bench_512_b8.json')); print(d['value'], d['roofline']['frac'], d['clip']['frames_per_s'], d['clip']['batch'])"
