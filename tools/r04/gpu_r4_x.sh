#!/bin/bash
# hourglass encoder: which levels take the F(4x4) form (per chain of 8 frames: enc3 = 32 tiles, enc2 = 128, enc1 = 512, enc0 = 2048)
O=gpurun_out/r04_x; mkdir -p $O
B="python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0"
run() { echo -n "$* : " >> $O/sweep.txt; env "$@" $B 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('%.1f frames/s  hg_enc %.3f hg_dec %.3f' % (d['value'], s.get('hg_enc', s.get('hourglass_enc', -1)), s.get('hg_dec', s.get('hourglass_dec', -1))))" >> $O/sweep.txt; }
run A=0
run EAMM_ENC_WINO_MIN_TILES=64
run EAMM_ENC_WINO_MIN_TILES=256
run EAMM_ENC_WINO_MIN_TILES=16
run A=0
cat $O/sweep.txt
