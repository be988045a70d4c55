#!/usr/bin/env python
"""Timing of the deconvolution tail (N3): eamm_amd.DeconvTail (one batched call per clip chunk) next to the
reference's stock-PyTorch nn.Sequential on the same GPU, driven the way AT_net2.forward drives it (one batch-1 call
per frame, util.py:603-607) and batched.  Usage: python tools/deconv_bench.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import DeconvTail  # noqa: E402
from eamm_amd.weights import deconv_state_dict_spec, synthetic_state_dict  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = synthetic_state_dict(None, seed=99, spec=deconv_state_dict_spec())
m = DeconvTail(max_batch=T).eval()
m.load_state_dict(sd)
m.cuda()
x = torch.randn(T, 256, 1, 1, device="cuda")


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


stock = lambda inp: torch.nn.Sequential.forward(m, inp)      # the children are the reference's layers (PyTorch-ROCm/MIOpen)
with torch.no_grad():
    ref = stock(x)
    err = float((m(x) - ref).abs().max())
    t_hip = timed(lambda: m(x))
    t_stock_b = timed(lambda: stock(x))
    t_stock_1 = timed(lambda: [stock(x[i:i + 1]) for i in range(T)], n=3)
flop = 2.0 * T * sum(ci * co * (16 if i == 0 else 4 * (4 << (i - 1)) ** 2 * 4) for i, (ci, co) in
                     enumerate(zip(m.channels[:-1], m.channels[1:])))
print(f"frames={T}  max|hip - stock| = {err:.2e}")
print(f"eamm_amd.DeconvTail, one call      : {t_hip:8.3f} ms  ({t_hip / T * 1e3:7.1f} us/frame, {flop / t_hip / 1e9:6.2f} TFLOP/s)")
print(f"stock PyTorch-ROCm, one batched call: {t_stock_b:8.3f} ms  ({t_stock_b / T * 1e3:7.1f} us/frame)")
print(f"stock PyTorch-ROCm, per frame (ref) : {t_stock_1:8.3f} ms  ({t_stock_1 / T * 1e3:7.1f} us/frame)")
