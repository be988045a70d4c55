#!/bin/bash
# input transform: channels per thread (EAMM_WINO4_TR_VEC) in the two-chain regime
O=gpurun_out/r04_x2; mkdir -p $O
B="python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0"
run() { echo -n "$* : " >> $O/sweep.txt; env "$@" $B 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('%.1f frames/s  transform %.3f conv %.3f' % (d['value'], s['bneck_transform'], s['bneck_conv']))" >> $O/sweep.txt; }
run A=0
run EAMM_WINO4_TR_VEC=2
run EAMM_WINO4_TR_VEC=1
run A=0
cat $O/sweep.txt
