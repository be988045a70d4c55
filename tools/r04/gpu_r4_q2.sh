#!/bin/bash
# BatchNorm: finalize step fused into the slice-combining kernel (one replica)
O=gpurun_out/r04_q2; mkdir -p $O
python -m pytest tests/test_gpu_backward.py tests/test_batchnorm.py tests/test_train_backward.py tests/test_train_mode.py -q -x -m gpu 2>&1 | tail -4
for i in 1 2; do python tools/train_step_bench.py 8 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd %.2f bwd %.2f step %.2f' % (d['forward_ms'], d['backward_ms'], d['step_ms']))"; done | tee $O/train.txt
