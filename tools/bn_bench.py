#!/usr/bin/env python
"""HBM rate of the training-mode BatchNorm forward kernels (N4 first slice) at the generator's activation sizes.
Algorithmic bytes: statistics pass 4 B/element, apply pass 8 B/element."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import SynchronizedBatchNorm2d

for shape in [(16, 256, 64, 64), (16, 128, 128, 128), (16, 64, 256, 256), (16, 1024, 4, 4)]:
    x = torch.randn(shape, device="cuda")
    m = SynchronizedBatchNorm2d(shape[1]).cuda().train()
    for _ in range(3):
        m(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    it = 20
    ev[0].record()
    for _ in range(it):
        s = m._ops.local_sums(x)
    ev[1].record()
    mean, scale = m._ops.finalize(s, m, 1)
    for _ in range(it):
        y = m._ops.apply(x, mean, scale, m.bias)
    ev[2].record()
    torch.cuda.synchronize()
    t_s, t_a = ev[0].elapsed_time(ev[1]) / it, ev[1].elapsed_time(ev[2]) / it
    b = x.numel() * 4
    print(f"{str(shape):22s} sums {t_s*1e3:7.1f} us = {b/t_s/1e6:6.0f} GB/s ({b/t_s/1e6/8000:.2f} of 8 TB/s)   "
          f"apply {t_a*1e3:7.1f} us = {2*b/t_a/1e6:6.0f} GB/s ({2*b/t_a/1e6/8000:.2f})")
