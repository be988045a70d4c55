#!/usr/bin/env python
"""HBM rate of the training-mode BatchNorm kernels (N4) at the generator's activation sizes.  Algorithmic bytes: forward
statistics pass 4 B/element, apply pass 8 B/element; backward sums 8 B/element (x and dy), apply 12 B/element.
Then the generator's .train() forward (N4 second slice) at 256x256, 16 pairs."""
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
    mean, scale, inv_std = m._ops.finalize(s, m, 1)
    for _ in range(it):
        y = m._ops.apply(x, mean, scale, m.bias)
    ev[2].record()
    torch.cuda.synchronize()
    t_s, t_a = ev[0].elapsed_time(ev[1]) / it, ev[1].elapsed_time(ev[2]) / it
    b = x.numel() * 4
    dy = torch.randn_like(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(it):
        bs = m._ops.backward_sums(x, dy, mean)
    ev[1].record()
    coef, dw, db = m._ops.backward_finalize(bs, bs, inv_std, m.weight, m.eps, 1, True)
    for _ in range(it):
        dx = m._ops.backward_apply(x, dy, mean, coef)
    ev[2].record()
    torch.cuda.synchronize()
    t_bs, t_ba = ev[0].elapsed_time(ev[1]) / it, ev[1].elapsed_time(ev[2]) / it
    print(f"{str(shape):22s} sums {t_s*1e3:7.1f} us = {b/t_s/1e6:6.0f} GB/s ({b/t_s/1e6/8000:.2f} of 8 TB/s)   "
          f"apply {t_a*1e3:7.1f} us = {2*b/t_a/1e6:6.0f} GB/s ({2*b/t_a/1e6/8000:.2f})   "
          f"bwd sums {t_bs*1e3:7.1f} us = {2*b/t_bs/1e6:6.0f} GB/s ({2*b/t_bs/1e6/8000:.2f})   "
          f"bwd apply {t_ba*1e3:7.1f} us = {3*b/t_ba/1e6:6.0f} GB/s ({3*b/t_ba/1e6/8000:.2f})")

# ---- the generator in .train(): 16 (source, key point) pairs at 256x256, every BatchNorm on batch statistics
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
cfg = hot_path_config()
gen = OcclusionAwareGenerator(**cfg); gen.load_state_dict(synthetic_state_dict(cfg)); gen = gen.cuda().train()
n = 16
src = synthetic_source(256, batch=n).cuda()
kp_s = {k: v.cuda() for k, v in synthetic_keypoints(n, 10, seed=0).items()}
kp_d = {k: v.cuda() for k, v in synthetic_keypoints(n, 10, seed=2).items()}
def timed(label):
    for _ in range(2):
        gen(src, kp_source=kp_s, kp_driving=kp_d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        gen(src, kp_source=kp_s, kp_driving=kp_d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"generator.train() forward, 256x256 x {n} pairs, {label}: {dt*1e3:.2f} ms = {n/dt:.0f} pairs/s")
with torch.no_grad():
    timed("graph-free resumable engine under torch.no_grad() (encoder + 27 BatchNorm sites on batch statistics, raw-weight direct convolutions)")
timed("with the autograd graph (differentiable HIP operators of train_graph: Winograd forms, fused BatchNorm + ReLU, motion_ops)")
