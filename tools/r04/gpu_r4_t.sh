#!/bin/bash
# training step: stacked flow head; sweep of the tile thresholds of the Winograd forms in the operator composition
O=gpurun_out/r04_t; mkdir -p $O
python -m pytest tests/test_gpu_motion_ops.py tests/test_train_backward.py -q -x -m gpu 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
run() { echo -n "$* : " >> $O/sweep.txt; env "$@" python tools/train_step_bench.py 8 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd %.2f bwd %.2f step %.2f' % (d['forward_ms'], d['backward_ms'], d['step_ms']))" >> $O/sweep.txt; }
run A=0
run EAMM_CONV_DEV_WINO4_MIN_TILES=512
run EAMM_CONV_DEV_WINO4_MIN_TILES=1024
run EAMM_CONV_DEV_WINO4_MIN_TILES=128
run EAMM_WGRAD_WINO4_MIN_TILES=128
run EAMM_WGRAD_WINO4_MIN_TILES=2048
run A=0
cat $O/sweep.txt
