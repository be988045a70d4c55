#!/bin/bash
# 32 frames in flight: 2 chains of 16 vs 4 chains of 8 (each chain's bottleneck GEMM = 128 one-per-CU workgroups; two of the four run at
# a time, the other chains' transforms / hourglass / up blocks beside them)
O=gpurun_out/r04_u; mkdir -p $O
B="python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 --steps 20 --warmup 5"
run() { echo -n "$* : " >> $O/sweep.txt; env "${@:2}" $B --batch $1 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.3f ms/step  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))" >> $O/sweep.txt; }
run 16 A=0
run 32 A=0
run 32 EAMM_PASS_CHAINS=4
run 32 EAMM_PASS_CHAINS=4 GPU_MAX_HW_QUEUES=8
run 32 EAMM_PASS_CHAINS=3
run 48 EAMM_PASS_CHAINS=3
run 64 EAMM_PASS_CHAINS=4
run 64 A=0
cat $O/sweep.txt
