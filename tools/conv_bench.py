#!/usr/bin/env python
"""Per-layer timing of the fp32-MFMA convolution kernel (eamm_op_conv timing mode) at the shapes of
the 256x256 / batch-16 workload.  Usage: python tools/conv_bench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None
SPLITK = int(os.environ.get("CONV_BENCH_SPLITK", "0"))   # explicit split-K (im2col kernels, polyphase patch kernel)
TILES = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]   # 0 auto, 1001.. dma tiles
# name, Hin, Win, C0, C1, Cout, ks(,kw), up, act, pool, resid
LAYERS = [
    ("bottleneck", 64, 64, 256, 0, 256, 3, 0, 0, 0, 1),
    ("up0", 64, 64, 256, 0, 128, 3, 1, 1, 0, 0),
    ("up1", 128, 128, 128, 0, 64, 3, 1, 1, 0, 0),
    ("final7x1", 256, 256, 64, 0, 21, (7, 1), 0, 0, 0, 0),
    ("final7x1n32", 256, 256, 64, 0, 32, (7, 1), 0, 0, 0, 0),
    ("hg_enc0", 64, 64, 64, 0, 128, 3, 0, 1, 1, 0),
    ("hg_enc1", 32, 32, 128, 0, 256, 3, 0, 1, 1, 0),
    ("hg_enc2", 16, 16, 256, 0, 512, 3, 0, 1, 1, 0),
    ("hg_enc3", 8, 8, 512, 0, 1024, 3, 0, 1, 1, 0),
    ("hg_enc4", 4, 4, 1024, 0, 1024, 3, 0, 1, 1, 0),
    ("hg_enc0_nopool", 64, 64, 64, 0, 128, 3, 0, 1, 0, 0),      # the same convolutions without the pooled epilogue
    ("hg_enc1_nopool", 32, 32, 128, 0, 256, 3, 0, 1, 0, 0),     # (what the Winograd op path accepts)
    ("hg_enc2_nopool", 16, 16, 256, 0, 512, 3, 0, 1, 0, 0),
    ("hg_enc3_nopool", 8, 8, 512, 0, 1024, 3, 0, 1, 0, 0),
    ("hg_dec0", 2, 2, 1024, 0, 1024, 3, 1, 1, 0, 0),
    ("hg_dec1", 4, 4, 1024, 1024, 512, 3, 1, 1, 0, 0),
    ("hg_dec2", 8, 8, 512, 512, 256, 3, 1, 1, 0, 0),
    ("hg_dec3", 16, 16, 256, 256, 128, 3, 1, 1, 0, 0),
    ("hg_dec4", 32, 32, 128, 128, 64, 3, 1, 1, 0, 0),
    ("head7x7", 64, 64, 64, 64, 12, 7, 0, 0, 0, 0),
]


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    total = 0.0
    for name, H, W, C0, C1, Cout, ks, up, act, pool, resid in LAYERS:
        if ONLY and name not in ONLY:
            continue
        ks, kw = ks if isinstance(ks, tuple) else (ks, ks)
        g = torch.Generator().manual_seed(1)
        in0 = torch.randn(B, H, W, C0, generator=g).to(dev)
        in1 = torch.randn(B, H, W, C1, generator=g).to(dev) if C1 else None
        cin = C0 + C1
        w = (torch.randn(Cout, cin, ks, kw, generator=g) * (2.0 / (cin * ks * kw)) ** 0.5).contiguous()
        b = torch.zeros(Cout)
        Ho, Wo = H << up, W << up
        res = torch.randn(B, Ho, Wo, Cout, generator=g).to(dev) if resid else None
        out = torch.empty(B, Ho >> pool, Wo >> pool, Cout, device=dev)
        flops = 2.0 * B * Ho * Wo * Cout * cin * ks * kw   # reference-algorithmic (un-collapsed) FLOPs
        for tile in TILES:
            if tile == 4000 and not (ks == 7 and kw == 1 and Cout == 32):
                continue
            if 1000 < tile < 4000 and (ks != 3 or kw != 3):
                continue
            if 2000 <= tile < 2200 and (up or pool or C1 or C0 % 64 or (tile >= 2100 and (H % 4 or W % 4))):
                continue
            if tile in (3000, 3003) and (not up or resid):
                continue
            if 1000 < tile < 2000 and {1001: 256, 1002: 128, 1003: 64, 1004: 128, 1005: 128}[tile] > max(Cout, 64) * 2:
                continue
            ms = C.c_float()
            rc = L.eamm_op_conv(0, in0.data_ptr(), C0, in1.data_ptr() if C1 else None, C1, B, H, W, up, w.data_ptr(),
                                b.data_ptr(), Cout, ks, kw, act, pool, res.data_ptr() if resid else None, SPLITK, tile,
                                out.data_ptr(), 20, C.byref(ms), st)
            _lib.check(rc, None)
            n = 12 if name == "bottleneck" else 1
            if tile == TILES[0]:
                total += ms.value * n
            print(f"{name:11s} tile={tile:4d} M={B*Ho*Wo:7d} N={Cout:4d} K={cin*ks*kw:5d}  {ms.value*1e3:8.1f} us  "
                  f"{flops/ms.value/1e9:7.1f} TF/s  ({flops/1e9:6.2f} GF)", flush=True)
    print(f"sum over a step (bottleneck x12): {total:.3f} ms")


if __name__ == "__main__":
    main()
