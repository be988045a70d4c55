#!/usr/bin/env python
"""Diagnostic: per-interval cycle stamps of block 0 of wino4_gemm_kernel (variant 16 writes them into `out`)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import _lib
# diagnostics that compute wrong results live in the experiments build only (make -C eamm_amd/csrc EXPERIMENTS=1)
_exp = os.path.join(os.path.dirname(_lib.LIB_PATH), "libeamm_hip_exp.so")
if os.path.exists(_exp):
    _lib.LIB_PATH = _exp
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, H, W, Cc = 16, 64, 64, 256
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(B, H, W, Cc, generator=g)).to(dev)
w = (torch.randn(Cc, Cc, 3, 3, generator=g) * (2.0 / (Cc * 9)) ** 0.5).contiguous(); b = torch.zeros(Cc)
out = torch.zeros(B, H, W, Cc, device=dev)
dbg = torch.zeros(B, H, W, Cc, device=dev)
ms = C.c_float()
rc = L.eamm_op_conv(0, x.data_ptr(), Cc, None, 0, B, H, W, 0, w.data_ptr(), b.data_ptr(), Cc, 3, 3, 0, 0, dbg.data_ptr(), 0, int(sys.argv[1]) if len(sys.argv) > 1 else 2116,
                    out.data_ptr(), 0, C.byref(ms), st)
_lib.check(rc, None); torch.cuda.synchronize()
t = dbg.view(torch.int64).flatten()[: 8 * 72 * 4].cpu().numpy().reshape(8, 72, 4)
t0 = t[:, 0, 0].min()
print("wave: interval: start  compute  fold  dma-wait | barrier-to-next-start")
for wv in (0, 4):
    for sc in list(range(2, 8)):
        a = t[wv, sc]
        nxt = t[wv, sc + 1, 0] - a[3] if sc + 1 < 72 else 0
        print(f"w{wv} i{sc:2d}: start {a[0]-t0:8d}  compute {a[1]-a[0]:6d}  fold {a[2]-a[1]:5d}  dmawait {a[3]-a[2]:6d}  barrier {nxt:6d}")
print("per-wave mean compute:", [(int((t[w, :, 1] - t[w, :, 0]).mean())) for w in range(8)])
print("per-wave mean fold   :", [(int((t[w, :, 2] - t[w, :, 1]).mean())) for w in range(8)])
print("per-wave mean barrier:", [(int((t[w, 1:, 0] - t[w, :-1, 3]).mean())) for w in range(8)])
tot = t[:, 71, 3].max() - t0
comp = (t[:, :, 1] - t[:, :, 0]).mean(); fold = (t[:, :, 2] - t[:, :, 1]).mean(); dw = (t[:, :, 3] - t[:, :, 2]).mean()
bar = (t[:, 1:, 0] - t[:, :-1, 3]).mean()
print(f"total {tot} cycles; mean per interval: compute {comp:.0f} fold {fold:.0f} dmawait {dw:.0f} barrier {bar:.0f}")
