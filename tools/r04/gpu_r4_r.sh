#!/bin/bash
# training step: forward's transformed input kept for the weight gradient (EAMM_SAVE_TRANSFORM), no bias padding launches
mkdir -p gpurun_out/r04_r
for v in 0 1 0 1; do
  EAMM_SAVE_TRANSFORM=$v python tools/train_step_bench.py 8 5 2>/dev/null | tail -1 | sed "s/^/save_transform=$v /" >> gpurun_out/r04_r/train_step.txt
done
EAMM_SAVE_TRANSFORM=1 python tools/train_step_bench.py 16 4 2>/dev/null | tail -1 | sed "s/^/save_transform=1 /" >> gpurun_out/r04_r/train_step.txt
cat gpurun_out/r04_r/train_step.txt
python -m pytest tests/test_train_backward.py tests/test_gpu_backward.py tests/test_abi_and_host.py -q -x -m gpu 2>&1 | tail -5 > gpurun_out/r04_r/tests.txt
cat gpurun_out/r04_r/tests.txt
