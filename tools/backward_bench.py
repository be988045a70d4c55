#!/usr/bin/env python
"""Timing of the backward kernels at the layer shapes of the 256x256 workload, through the C ABI:
eamm_op_conv_wgrad (weight + bias gradient), the data gradient (eamm_op_conv on the transposed, flipped filter) and
eamm_op_warp_backward.  HIP events on torch's current stream around `ITERS` back-to-back calls.
Usage: python tools/backward_bench.py [B]   (B = frames per launch, default 16)"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ITERS = 20
PEAK_TF, PEAK_GB = 157.3, 8000.0
# name, H, W, Cin, Cout, k : the generator's convolutions (reference modules/util.py:858-938, generator.py:14-48)
LAYERS = [
    ("bottleneck 3x3 256->256 @64", 64, 64, 256, 256, 3),
    ("down1 3x3 128->256 @128", 128, 128, 128, 256, 3),
    ("down0 3x3 64->128 @256", 256, 256, 64, 128, 3),
    ("up0 (after x2) 3x3 256->128 @128", 128, 128, 256, 128, 3),
    ("up1 (after x2) 3x3 128->64 @256", 256, 256, 128, 64, 3),
    ("hourglass enc3 3x3 512->1024 @8", 8, 8, 512, 1024, 3),
]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for name, H, W, cin, cout, k in LAYERS:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, H, W, cin, generator=g).to(dev)
        dy = torch.randn(B, H, W, cout, generator=g).to(dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        db = torch.empty(cout, device=dev)
        nwork = L.eamm_op_conv_wgrad_workspace_floats(B, H, W, cin, cout, k, k)
        work = torch.empty(nwork, device=dev)
        ms_w = timed(lambda: _lib.check(L.eamm_op_conv_wgrad(0, x.data_ptr(), dy.data_ptr(), B, H, W, cin, cout, k, k, dw.data_ptr(),
                                                             db.data_ptr(), work.data_ptr(), nwork, st), None))
        flops = 2.0 * B * H * W * cin * cout * k * k
        row = {"layer": name, "frames": B, "gflop": flops / 1e9, "wgrad_ms": ms_w, "wgrad_tflops": flops / ms_w / 1e9,
               "wgrad_frac": flops / ms_w / 1e9 / PEAK_TF}
        # data gradient: the forward kernels on grad_out with the transposed, flipped filter (algorithmic flops of a direct conv)
        wt = (torch.randn(cin, cout, k, k, generator=g) * (2.0 / (cout * k * k)) ** 0.5).contiguous()
        zb = torch.zeros(cin)
        dx = torch.empty(B, H, W, cin, device=dev)
        for label, tile in (("direct", 0), ("wino4", 2103)):
            if tile and (cout % 128 or H % 4 or W % 4):
                continue
            ms = C.c_float()
            _lib.check(L.eamm_op_conv(0, dy.data_ptr(), cout, None, 0, B, H, W, 0, wt.data_ptr(), zb.data_ptr(), cin, k, k, 0, 0, None,
                                      0, tile, dx.data_ptr(), ITERS, C.byref(ms), st), None)
            row[f"dgrad_{label}_ms"] = ms.value
            row[f"dgrad_{label}_tflops_algorithmic"] = flops / ms.value / 1e9
        rows.append(row)
        print(json.dumps(row))
    # warp backward at the generator's shape: [B,64,64,256] features of ONE source, per-frame flow and occlusion
    n, h, w, c = B, 64, 64, 256
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(1, h, w, c, generator=g).to(dev)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    defo = (torch.stack([xs, ys], -1)[None] + 0.05 * torch.randn(n, h, w, 2, generator=g)).contiguous().to(dev)
    occ = torch.rand(n, h, w, generator=g).to(dev)
    gout = torch.randn(n, h, w, c, generator=g).to(dev)
    for ns, label in ((1, "one source (frames' gradients meet on one map)"), (n, "per-frame source")):
        f = feat if ns == 1 else feat.expand(n, -1, -1, -1).contiguous()
        gf, gd, go = torch.empty_like(f), torch.empty_like(defo), torch.empty_like(occ)
        ms = timed(lambda: _lib.check(L.eamm_op_warp_backward(0, f.data_ptr(), defo.data_ptr(), occ.data_ptr(), gout.data_ptr(), n, ns,
                                                              h, w, c, gf.data_ptr(), gd.data_ptr(), go.data_ptr(), st), None))
        # algorithmic bytes: grad_out read once, the source map read once and its gradient written once, flow/occlusion in and out
        nbytes = 4.0 * (n * h * w * c + 2 * ns * h * w * c + n * h * w * 6)
        row = {"op": "warp_backward", "case": label, "frames": n, "ms": ms, "algorithmic_mb": nbytes / 1e6,
               "gbps": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / PEAK_GB}
        print(json.dumps(row))


if __name__ == "__main__":
    main()
