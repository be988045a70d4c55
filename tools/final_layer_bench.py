#!/usr/bin/env python
"""The final 7x7 layer (64 -> 3, + sigmoid) alone: eamm_op_conv tile 4002 = the fused column-patch kernel, timed with HIP events
(20 launches).  Run once per environment setting (the knobs are read once per process):
    EAMM_FINAL_MFMA4=0|1  EAMM_COL7_DBG=0|1|2|3   python tools/final_layer_bench.py [frames ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import _lib  # noqa: E402
# diagnostics that compute wrong results live in the experiments build only (make -C eamm_amd/csrc EXPERIMENTS=1)
_exp = os.path.join(os.path.dirname(_lib.LIB_PATH), "libeamm_hip_exp.so")
if os.path.exists(_exp):
    _lib.LIB_PATH = _exp

L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
tag = " ".join(f"{k}={os.environ[k]}" for k in ("EAMM_FINAL_MFMA4", "EAMM_COL7_DBG") if k in os.environ) or "default"
for B in [int(a) for a in sys.argv[1:]] or [8, 16]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 256, 256, 64, generator=g).to(dev)
    w = (torch.randn(3, 64, 7, 7, generator=g) * (2.0 / (64 * 49)) ** 0.5).contiguous()
    b = torch.randn(3, generator=g) * 0.1
    out = torch.empty(B, 3, 256, 256, device=dev)
    ms = C.c_float()
    _lib.check(L.eamm_op_conv(0, x.data_ptr(), 64, None, 0, B, 256, 256, 0, w.data_ptr(), b.data_ptr(), 3, 7, 7, 2, 0, None, 0, 4002,
                              out.data_ptr(), 20, C.byref(ms), st), None)
    ref = torch.sigmoid(torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).cpu(), w, b, padding=3))
    err = float((out[:1].cpu() - ref).abs().max())
    algo = 2.0 * B * 256 * 256 * 3 * 64 * 49
    print(f"[{tag}] final layer, {B} frames: {ms.value * 1e3:8.1f} us  {algo / ms.value / 1e9:6.1f} TF/s algorithmic  max|hip - torch| {err:.2e}", flush=True)
