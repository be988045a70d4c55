#!/bin/bash
# num_channels 1 / 2: reference fixtures through the module, the clip interface, the operator composition's gradients
mkdir -p gpurun_out/r04_p
python -m pytest tests/test_gpu_generator.py -q -x -m gpu -k "reference_fixture or gray or constructor_variants" -s 2>&1 | tail -40 > gpurun_out/r04_p/gen.txt
python -m pytest tests/test_train_backward.py tests/test_train_mode.py -q -x -m gpu -k "variants or not_multiples" -s 2>&1 | tail -30 > gpurun_out/r04_p/bwd.txt
tail -n 5 gpurun_out/r04_p/gen.txt gpurun_out/r04_p/bwd.txt
