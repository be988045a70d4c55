#!/bin/bash
# weight gradient on a side stream beside the data gradient: training step off / on, gradient parity
mkdir -p gpurun_out/r04_q
for v in 0 1 0 1; do
  EAMM_WGRAD_STREAM=$v python tools/train_step_bench.py 8 5 2>/dev/null | tail -1 | sed "s/^/wgrad_stream=$v /" >> gpurun_out/r04_q/train_step.txt
done
EAMM_WGRAD_STREAM=1 python tools/train_step_bench.py 16 4 2>/dev/null | tail -1 | sed "s/^/wgrad_stream=1 /" >> gpurun_out/r04_q/train_step.txt
EAMM_WGRAD_STREAM=0 python tools/train_step_bench.py 16 4 2>/dev/null | tail -1 | sed "s/^/wgrad_stream=0 /" >> gpurun_out/r04_q/train_step.txt
cat gpurun_out/r04_q/train_step.txt
python -m pytest tests/test_train_backward.py tests/test_gpu_backward.py -q -x -m gpu 2>&1 | tail -5 > gpurun_out/r04_q/tests.txt
cat gpurun_out/r04_q/tests.txt
