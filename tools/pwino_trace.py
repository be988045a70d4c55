#!/usr/bin/env python
"""Diagnostic: per-interval cycle stamps of workgroup 0 of conv_patch_wino_kernel (eamm_op_conv tile 3002)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, H, W, Ci, Co = 16, 64, 64, 256, 128
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(B, H, W, Ci, generator=g)).to(dev)
w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (Ci * 9)) ** 0.5).contiguous(); b = torch.zeros(Co)
out = torch.zeros(B, 2 * H, 2 * W, Co, device=dev)
dbg = torch.zeros(8 * 72 * 3, dtype=torch.int64, device=dev)
ms = C.c_float()
rc = L.eamm_op_conv(0, x.data_ptr(), Ci, None, 0, B, H, W, 1, w.data_ptr(), b.data_ptr(), Co, 3, 3, 1, 0, dbg.data_ptr(), 0, 3002,
                    out.data_ptr(), 0, C.byref(ms), st)
_lib.check(rc, None); torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(8, 72, 3)
print("per-wave mean compute :", [int((t[w, :, 1] - t[w, :, 0]).mean()) for w in range(8)])
print("per-wave mean dma wait:", [int((t[w, :, 2] - t[w, :, 1]).mean()) for w in range(8)])
print("per-wave mean barrier :", [int((t[w, 1:, 0] - t[w, :-1, 2]).mean()) for w in range(8)])
print("interval (wave 0)     :", int((t[0, 1:, 0] - t[0, :-1, 0]).mean()), "cycles; by transform point:",
      [int((t[0, xi + 9:72:9, 0] - t[0, xi + 8:71:9, 0]).mean()) if xi else int((t[0, 9:72:9, 0] - t[0, 8:71:9, 0]).mean()) for xi in range(9)])
