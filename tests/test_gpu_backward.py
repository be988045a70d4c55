"""Backward kernels of the path's operator kinds against torch-CPU autograd of the reference's
ops: F.grid_sample(...) * occlusion (reference modules/generator.py:50-57, 79-84) and F.conv2d (modules/util.py:858-938).
Through the C ABI (eamm_op_warp_backward, eamm_op_conv_wgrad, eamm_op_conv) via eamm_amd.autograd_ops."""
import pytest
import torch
import torch.nn.functional as F

from eamm_amd import _lib, autograd_ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _identity_grid(n, h, w):
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    return torch.stack([xs, ys], -1)[None].expand(n, -1, -1, -1)


def _warp_inputs(n, ns, h, w, c, occ, seed, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(ns, c, h, w, generator=g)
    defo = _identity_grid(n, h, w) + spread * torch.randn(n, h, w, 2, generator=g)   # some samples leave the map: zero padding
    om = torch.rand(n, 1, h, w, generator=g) if occ else None
    gout = torch.randn(n, c, h, w, generator=g)
    return feat, defo, om, gout


def _warp_reference(feat, defo, om, gout):
    f, d = feat.double().requires_grad_(), defo.double().requires_grad_()
    o = om.double().requires_grad_() if om is not None else None
    src = f if f.shape[0] == d.shape[0] else f.expand(d.shape[0], -1, -1, -1)
    y = F.grid_sample(src, d, mode="bilinear", padding_mode="zeros", align_corners=False)
    if o is not None:
        y = y * o
    y.backward(gout.double())
    return y.detach(), f.grad, d.grad, (o.grad if o is not None else None)


@pytest.mark.parametrize("name,kw", [
    ("one_source", dict(n=3, ns=1, h=16, w=16, c=64, occ=True, seed=0)),
    ("per_frame_source", dict(n=2, ns=2, h=12, w=20, c=32, occ=True, seed=1)),
    ("no_occlusion", dict(n=2, ns=1, h=16, w=16, c=256, occ=False, seed=2)),
    ("wide_channels", dict(n=1, ns=1, h=8, w=8, c=512, occ=True, seed=3)),          # 128 lanes per pixel: no in-wave reduction
    ("odd_channel_groups", dict(n=2, ns=2, h=10, w=6, c=24, occ=True, seed=4)),      # 6 lanes per pixel: not a power of two
    ("generator_shape", dict(n=2, ns=1, h=64, w=64, c=256, occ=True, seed=5, spread=0.1)),
])
def test_warp_backward_matches_autograd(name, kw):
    feat, defo, om, gout = _warp_inputs(**kw)
    want_y, want_f, want_d, want_o = _warp_reference(feat, defo, om, gout)
    f = feat.to(DEV).requires_grad_()
    d = defo.to(DEV).requires_grad_()
    o = om.to(DEV).requires_grad_() if om is not None else None
    y = autograd_ops.warp(f, d, o)
    y.backward(gout.to(DEV))
    torch.cuda.synchronize()

    def rel(got, want):
        return float((got.detach().cpu().double() - want).abs().max()) / max(1.0, float(want.abs().max()))

    errs = {"out": rel(y, want_y), "d_feat": rel(f.grad, want_f), "d_flow": rel(d.grad, want_d)}
    if om is not None:
        errs["d_occ"] = rel(o.grad, want_o)
    print(f"warp backward {name}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # fp32 kernel vs float64 autograd (the sample coordinate itself rounds at 2^-24 * W); d_flow carries the W/2 factor and a difference of corner values
    assert errs["out"] <= 1e-5 and errs["d_feat"] <= 5e-6 and errs["d_flow"] <= 2e-5, (name, errs)
    assert errs.get("d_occ", 0.0) <= 1e-5, (name, errs)


def test_warp_backward_of_out_of_range_samples_is_zero():
    # every sample outside the map by more than a pixel: zero output, zero gradients, nothing non-finite
    n, h, w, c = 2, 8, 8, 32
    feat = torch.randn(1, c, h, w)
    defo = torch.full((n, h, w, 2), 1.9)
    om = torch.rand(n, 1, h, w)
    f, d, o = feat.to(DEV).requires_grad_(), defo.to(DEV).requires_grad_(), om.to(DEV).requires_grad_()
    y = autograd_ops.warp(f, d, o)
    y.backward(torch.ones_like(y))
    for t in (y, f.grad, d.grad, o.grad):
        assert float(t.detach().abs().max()) == 0.0


def test_warp_backward_partial_gradients():
    # only the flow needs a gradient (the dense-motion branch with a detached source): the other outputs are not computed
    feat, defo, om, gout = _warp_inputs(2, 1, 16, 16, 64, True, 7)
    _, _, want_d, _ = _warp_reference(feat, defo, om, gout)
    d = defo.to(DEV).requires_grad_()
    y = autograd_ops.warp(feat.to(DEV), d, om.to(DEV))
    y.backward(gout.to(DEV))
    assert float((d.grad.cpu().double() - want_d).abs().max()) <= 2e-5 * max(1.0, float(want_d.abs().max()))


CONV_CASES = {
    # the generator's layer shapes at reduced extents
    "3x3_64_64":     dict(B=2, H=16, W=16, cin=64, cout=64, k=3, bias=True),
    "3x3_256_256":   dict(B=2, H=16, W=16, cin=256, cout=256, k=3, bias=True),        # bottleneck ResBlock2d
    "3x3_128_256":   dict(B=1, H=20, W=12, cin=128, cout=256, k=3, bias=True),        # ragged extents, DownBlock2d
    "3x3_96_32":     dict(B=3, H=9, W=7, cin=96, cout=32, k=3, bias=False),           # tile tails in both channel dims
    "7x7_64_32":     dict(B=1, H=24, W=24, cin=64, cout=32, k=7, bias=True),          # 7x7 (final / first layer kind)
    "3x3_pixels_not_multiple_of_32": dict(B=1, H=5, W=5, cin=32, cout=32, k=3, bias=True),
    # widths that are multiples of 32: the filter-row kernel (a K chunk is a segment of one image row)
    "row_3x3_w32":   dict(B=2, H=8, W=32, cin=64, cout=96, k=3, bias=True),
    "row_3x3_w64":   dict(B=3, H=5, W=64, cin=128, cout=64, k=3, bias=True),
    "row_7x7_w32":   dict(B=1, H=6, W=32, cin=32, cout=32, k=7, bias=True),           # taller filter than some maps' margin
    "row_7x7_w96":   dict(B=2, H=10, W=96, cin=64, cout=32, k=7, bias=False),
    # >= 2048 4x4 tiles: forward and data gradient on the F(4x4,3x3) Winograd kernels, filter transformed on the device
    "wino4_64_64":   dict(B=2, H=128, W=128, cin=64, cout=64, k=3, bias=True),
    "wino4_128_64":  dict(B=8, H=64, W=64, cin=128, cout=64, k=3, bias=True),          # transposed filter is not square
    # the generator's ACTUAL layer shapes at 256x256, 2 pairs: the bottleneck convolution (512 tiles: the Winograd
    # weight-gradient form and the split-group F(4x4) forward / data gradient), the last up block's convolution (after x2), the
    # 7x7 on 64 channels at full resolution (filter-row weight gradient, W % 32 == 0)
    "gen_bottleneck_2x64x64":   dict(B=2, H=64, W=64, cin=256, cout=256, k=3, bias=True),
    "gen_up1_2x256x256":        dict(B=2, H=256, W=256, cin=128, cout=64, k=3, bias=True),
    "gen_7x7_2x256x256_64_32":  dict(B=2, H=256, W=256, cin=64, cout=32, k=7, bias=True),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_backward_matches_autograd(name):
    kw = CONV_CASES[name]
    g = torch.Generator().manual_seed(100 + list(CONV_CASES).index(name))
    B, H, W, cin, cout, k = kw["B"], kw["H"], kw["W"], kw["cin"], kw["cout"], kw["k"]
    x = torch.randn(B, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = 0.1 * torch.randn(cout, generator=g) if kw["bias"] else None
    gout = torch.randn(B, cout, H, W, generator=g)

    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    br = b.double().requires_grad_() if b is not None else None
    yr = F.conv2d(xr, wr, br, padding=k // 2)
    yr.backward(gout.double())

    xd, wd = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if b is not None else None
    y = autograd_ops.conv2d_same(xd, wd, bd)
    y.backward(gout.to(DEV))
    torch.cuda.synchronize()

    def rel(got, want):
        return float((got.detach().cpu().double() - want).abs().max()) / max(1.0, float(want.abs().max()))

    errs = {"out": rel(y, yr.detach()), "dx": rel(xd.grad, xr.grad), "dw": rel(wd.grad, wr.grad)}
    if b is not None:
        errs["db"] = rel(bd.grad, br.grad)
    print(f"conv backward {name}: " + ", ".join(f"{k_} {v:.2e}" for k_, v in errs.items()))
    # fp32 MFMA accumulation vs float64: K = cin*k*k for out/dx, K = B*H*W pixels for dw
    assert all(v <= 2e-5 for v in errs.values()), (name, errs)


def test_conv_wgrad_is_deterministic_and_linear():
    # fixed-order split reduction: bit-identical across runs; linear in grad_out
    B, H, W, cin, cout, k = 2, 32, 32, 64, 64, 3
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    g1 = torch.randn(B, H, W, cout, generator=g).to(DEV)
    g2 = torch.randn(B, H, W, cout, generator=g).to(DEV)
    L = _lib.lib()
    nwork = L.eamm_op_conv_wgrad_workspace_floats(B, H, W, cin, cout, k, k)
    work = torch.empty(nwork, device=DEV)

    def wgrad(go):
        dw = torch.full((cout, cin, k, k), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        _lib.check(L.eamm_op_conv_wgrad(0, x.data_ptr(), go.data_ptr(), B, H, W, cin, cout, k, k, dw.data_ptr(), db.data_ptr(),
                                        work.data_ptr(), nwork, torch.cuda.current_stream().cuda_stream), None)
        torch.cuda.synchronize()
        return dw, db

    a1, b1 = wgrad(g1)
    a1b, b1b = wgrad(g1)
    assert torch.equal(a1, a1b) and torch.equal(b1, b1b)
    a2, _ = wgrad(g2)
    a12, _ = wgrad(g1 + g2)
    assert float((a12 - (a1 + a2)).abs().max()) <= 1e-4 * float(a12.abs().max())


def test_conv_wgrad_from_the_forwards_saved_transform_is_bit_identical():
    """eamm_op_conv_dev in its F(4x4,3x3) form leaves V = B^T x B in its workspace; eamm_op_conv_wgrad_saved starts the F(3x3,4x4)
    weight gradient from it instead of transforming x again -- the same sums in the same order as eamm_op_conv_wgrad."""
    B, H, W, cin, cout = 8, 64, 64, 64, 128
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    go = torch.randn(B, H, W, cout, generator=g).to(DEV)
    L = _lib.lib()
    off = L.eamm_op_conv_saved_transform_offset(B, H, W, cin, cout, 3, 3)
    assert off != 2 ** 64 - 1
    assert L.eamm_op_conv_saved_transform_offset(1, 16, 16, cin, cout, 3, 3) == 2 ** 64 - 1     # too few tiles: direct forward
    assert L.eamm_op_conv_saved_transform_offset(B, H, W, cin, cout, 7, 7) == 2 ** 64 - 1
    nfw = L.eamm_op_conv_dev_workspace_floats(B, H, W, cin, cout, 3, 3)
    fwork = torch.empty(nfw, device=DEV)
    out = torch.empty(B, H, W, cout, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.eamm_op_conv_dev(0, x.data_ptr(), B, H, W, cin, wt.data_ptr(), None, cout, 3, 3, 0, out.data_ptr(), fwork.data_ptr(), nfw, st), None)
    nwork = L.eamm_op_conv_wgrad_workspace_floats(B, H, W, cin, cout, 3, 3)
    work = torch.empty(nwork, device=DEV)
    dw0, db0 = torch.empty_like(wt), torch.empty(cout, device=DEV)
    dw1, db1 = torch.full_like(wt, float("nan")), torch.full((cout,), float("nan"), device=DEV)
    _lib.check(L.eamm_op_conv_wgrad(0, x.data_ptr(), go.data_ptr(), B, H, W, cin, cout, 3, 3, dw0.data_ptr(), db0.data_ptr(),
                                    work.data_ptr(), nwork, st), None)
    _lib.check(L.eamm_op_conv_wgrad_saved(0, fwork.data_ptr() + 4 * off, go.data_ptr(), B, H, W, cin, cout, dw1.data_ptr(), db1.data_ptr(),
                                          work.data_ptr(), nwork, st), None)
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    assert L.eamm_op_conv_wgrad_saved(0, fwork.data_ptr(), go.data_ptr(), 1, 16, 16, cin, cout, dw1.data_ptr(), None, work.data_ptr(), nwork, st) != 0


@pytest.mark.parametrize("H,W", [(12, 32), (9, 20)])
def test_conv_wgrad_7x1_filter(H, W):
    # the path's column convolutions (7x1): filter-row kernel with one tap per row / per-tap kernel, against autograd
    B, cin, cout = 2, 64, 32
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, cin, H, W, generator=g)
    gout = torch.randn(B, cout, H, W, generator=g)
    wr = torch.zeros(cout, cin, 7, 1, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, None, padding=(3, 0)).backward(gout.double())
    L = _lib.lib()
    xd, gd = x.permute(0, 2, 3, 1).contiguous().to(DEV), gout.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.full((cout, cin, 7, 1), float("nan"), device=DEV)
    nwork = L.eamm_op_conv_wgrad_workspace_floats(B, H, W, cin, cout, 7, 1)
    work = torch.empty(nwork, device=DEV)
    _lib.check(L.eamm_op_conv_wgrad(0, xd.data_ptr(), gd.data_ptr(), B, H, W, cin, cout, 7, 1, dw.data_ptr(), None, work.data_ptr(),
                                    nwork, torch.cuda.current_stream().cuda_stream), None)
    torch.cuda.synchronize()
    err = float((dw.cpu().double() - wr.grad).abs().max()) / max(1.0, float(wr.grad.abs().max()))
    assert err <= 2e-5, err


@pytest.mark.parametrize("relu,pool,sync", [(True, False, None), (True, True, None), (False, False, None), (True, True, True)])
def test_fused_nhwc_batchnorm_matches_autograd(relu, pool, sync):
    # [avgpool2x2](relu(BatchNorm(x))) on NHWC, training mode (reference modules/util.py:858-938 block tails): forward, running
    # statistics and every gradient against torch autograd in double; sync=True: the replicas' clamp formula on one rank
    from eamm_amd.sync_batchnorm import SynchronizedBatchNorm2d, _BatchNormNHWCFunction
    B, H, W, C = 3, 10, 12, 40
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    wt, bs = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    gout = torch.randn(B, C, H // 2 if pool else H, W // 2 if pool else W, generator=g)
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), bs.double().requires_grad_()
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    if sync:    # sync_batchnorm/batchnorm.py:110-125: clamp(var, eps) ** -0.5, running_var from the unbiased variance
        mean = xr.mean(dim=(0, 2, 3))
        var = ((xr - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        y = (xr - mean[None, :, None, None]) * (var.clamp(1e-5) ** -0.5 * wr)[None, :, None, None] + br[None, :, None, None]
        n = B * H * W
        rm, rv = 0.9 * rm + 0.1 * mean.detach(), 0.9 * rv + 0.1 * var.detach() * n / (n - 1)
    else:
        y = F.batch_norm(xr, rm, rv, wr, br, True, 0.1, 1e-5)
    y = F.relu(y) if relu else y
    y = F.avg_pool2d(y, 2) if pool else y
    y.backward(gout.double())

    mod = SynchronizedBatchNorm2d(C, sync=sync).to(DEV).train()
    with torch.no_grad():
        mod.weight.copy_(wt)
        mod.bias.copy_(bs)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    out = _BatchNormNHWCFunction.apply(xd, mod.weight, mod.bias, mod, relu, pool)
    out.backward(gout.permute(0, 2, 3, 1).contiguous().to(DEV))
    torch.cuda.synchronize()

    def rel(got, want):
        return float((got.detach().cpu().double() - want).abs().max()) / max(1.0, float(want.abs().max()))

    errs = {"y": rel(out.permute(0, 3, 1, 2), y.detach()), "dx": rel(xd.grad.permute(0, 3, 1, 2), xr.grad),
            "dw": rel(mod.weight.grad, wr.grad), "db": rel(mod.bias.grad, br.grad),
            "rm": rel(mod.running_mean, rm), "rv": rel(mod.running_var, rv)}
    assert all(v <= 5e-6 for v in errs.values()), errs


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_statistics_entries_equal_the_separate_ones_bit_for_bit(mode):
    """One replica: eamm_bn_nhwc_local_stats = eamm_bn_nhwc_local_sums + eamm_bn_finalize, eamm_bn_nhwc_backward_local =
    eamm_bn_nhwc_backward_sums + eamm_bn_backward_finalize (the finalize step inside the kernel that adds the slices up)."""
    L = _lib.lib()
    B, H, W, C = 4, 16, 24, 96
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).to(DEV)
    dy = torch.randn(B, H // 2, W // 2, C, generator=g).to(DEV)
    wt, bs = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    M = B * H * W
    work = torch.empty(L.eamm_bn_nhwc_workspace_floats(M, C), device=DEV)
    p = lambda t: t.data_ptr()
    new = lambda n: torch.full((n,), float("nan"), device=DEV)

    def forward(fused):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        sums, mean, scale, inv = new(6 * C + 2), new(C), new(C), new(C)
        if fused:
            _lib.check(L.eamm_bn_nhwc_local_stats(p(x), M, C, 1e-5, 0.1, mode, p(wt), p(rm), p(rv), p(sums), p(mean), p(scale), p(inv), p(work), st), None)
        else:
            _lib.check(L.eamm_bn_nhwc_local_sums(p(x), M, C, p(sums), p(work), st), None)
            _lib.check(L.eamm_bn_finalize(p(sums), C, 1e-5, 0.1, mode, p(wt), p(rm), p(rv), p(mean), p(scale), p(inv), st), None)
        return rm, rv, sums[:2 * C + 2].clone(), mean, scale, inv

    a, b = forward(False), forward(True)
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    mean, scale, inv = a[3], a[4], a[5]

    def backward(fused):
        sums, coef, dw, db = new(6 * C + 2), new(3 * C), new(C), new(C)
        if fused:
            _lib.check(L.eamm_bn_nhwc_backward_local(p(x), p(dy), p(mean), p(scale), p(bs), B, H, W, C, 1, 1, p(inv), p(wt), 1e-5, mode,
                                                     p(sums), p(dw), p(db), p(coef), p(work), st), None)
        else:
            _lib.check(L.eamm_bn_nhwc_backward_sums(p(x), p(dy), p(mean), p(scale), p(bs), B, H, W, C, 1, 1, p(sums), p(work), st), None)
            _lib.check(L.eamm_bn_backward_finalize(p(sums), p(sums), C, p(inv), p(wt), 1e-5, mode, p(dw), p(db), p(coef), st), None)
        return coef, dw, db

    a, b = backward(False), backward(True)
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) and bool(torch.isfinite(u).all()) for u, v in zip(a, b))
    assert L.eamm_bn_nhwc_local_stats(p(x), M, C, 1e-5, 0.1, 2, p(wt), p(wt), p(wt), p(work), p(wt), p(wt), p(wt), p(work), st) != 0   # mode 2: no statistics


def test_backward_entry_points_reject_bad_arguments():
    L = _lib.lib()
    t = torch.zeros(64, device=DEV)
    assert L.eamm_op_warp_backward(0, t.data_ptr(), t.data_ptr(), None, t.data_ptr(), 1, 1, 2, 2, 6, t.data_ptr(), None, None, None) != 0
    assert L.eamm_op_warp_backward(0, t.data_ptr(), t.data_ptr(), None, t.data_ptr(), 1, 1, 2, 2, 8, None, None, None, None) != 0
    assert L.eamm_op_conv_wgrad(0, t.data_ptr(), t.data_ptr(), 1, 2, 2, 4, 4, 2, 2, t.data_ptr(), None, t.data_ptr(), 64, None) != 0
    assert L.eamm_op_conv_wgrad(0, t.data_ptr(), t.data_ptr(), 1, 2, 2, 4, 4, 3, 3, t.data_ptr(), None, t.data_ptr(), 1, None) != 0   # workspace too small
    with pytest.raises(RuntimeError):
        autograd_ops.warp(torch.zeros(1, 8, 4, 4), torch.zeros(1, 4, 4, 2))   # CPU tensors: no fallback


# ---- the 7x7 layers with three channels on one side (reference modules/generator.py:26, 48), csrc/conv7_thin.hip ------------
def _thin4(t_nchw3):
    """[B,3,H,W] -> NHWC with four floats per pixel (fourth zero), on the GPU."""
    return F.pad(t_nchw3.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV)


@pytest.mark.parametrize("B,H,W,N", [(2, 16, 16, 64), (1, 21, 37, 32), (2, 64, 64, 64)])
def test_thin_7x7_layers_match_autograd(B, H, W, N):
    g = torch.Generator().manual_seed(31 + H)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nwork = L.eamm_op_conv7_thin_workspace_floats(B, H, W, N)
    work = torch.empty(nwork, device=DEV)

    def rel(got, want):
        return float((got.detach().cpu().double() - want).abs().max()) / max(1.0, float(want.abs().max()))

    # `first`: y = conv(x3, w [N,3,7,7]) + b: forward and weight gradient
    x3 = torch.randn(B, 3, H, W, generator=g)
    w1 = torch.randn(N, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    b1 = 0.1 * torch.randn(N, generator=g)
    gy = torch.randn(B, N, H, W, generator=g)
    wr = w1.double().requires_grad_()
    yr = F.conv2d(x3.double(), wr, b1.double(), padding=3)
    yr.backward(gy.double())
    x4 = _thin4(x3)
    y = torch.full((B, H, W, N), float("nan"), device=DEV)
    w1_d, b1_d = w1.to(DEV), b1.to(DEV)
    _lib.check(L.eamm_op_conv7_thin(0, x4.data_ptr(), w1_d.data_ptr(), b1_d.data_ptr(), B, H, W, N, 0, y.data_ptr(),
                                    work.data_ptr(), nwork, st), None)
    dw = torch.full((N, 3, 7, 7), float("nan"), device=DEV)
    gy_d = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    _lib.check(L.eamm_op_conv7_thin_wgrad(0, x4.data_ptr(), gy_d.data_ptr(), B, H, W, N, 1, dw.data_ptr(), work.data_ptr(), nwork, st), None)
    torch.cuda.synchronize()
    errs = {"first_fwd": rel(y.permute(0, 3, 1, 2), yr.detach()), "first_dw": rel(dw, wr.grad)}

    # `final`: y3 = conv(x [.,N], w [3,N,7,7]): data gradient and weight gradient from d y3
    x = torch.randn(B, N, H, W, generator=g)
    w2 = torch.randn(3, N, 7, 7, generator=g) * (2.0 / (49 * N)) ** 0.5
    g3 = torch.randn(B, 3, H, W, generator=g)
    xr, wr2 = x.double().requires_grad_(), w2.double().requires_grad_()
    F.conv2d(xr, wr2, None, padding=3).backward(g3.double())
    g4 = _thin4(g3)
    dx = torch.full((B, H, W, N), float("nan"), device=DEV)
    w2_d = w2.to(DEV)
    _lib.check(L.eamm_op_conv7_thin(0, g4.data_ptr(), w2_d.data_ptr(), None, B, H, W, N, 1, dx.data_ptr(), work.data_ptr(), nwork, st), None)
    dw2 = torch.full((3, N, 7, 7), float("nan"), device=DEV)
    x_d = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    _lib.check(L.eamm_op_conv7_thin_wgrad(0, g4.data_ptr(), x_d.data_ptr(), B, H, W, N, 0, dw2.data_ptr(), work.data_ptr(), nwork, st), None)
    torch.cuda.synchronize()
    errs.update({"final_dx": rel(dx.permute(0, 3, 1, 2), xr.grad), "final_dw": rel(dw2, wr2.grad)})
    # `final` forward with device parameters: sigmoid(conv + bias) -> NCHW, on the fused column-patch kernel
    b2 = 0.1 * torch.randn(3, generator=g)
    b2_d = b2.to(DEV)
    y3 = torch.full((B, 3, H, W), float("nan"), device=DEV)
    pw = torch.empty(7 * N * 32, device=DEV)
    _lib.check(L.eamm_op_final_conv_sigmoid(0, x_d.data_ptr(), w2_d.data_ptr(), b2_d.data_ptr(), B, H, W, N, y3.data_ptr(), pw.data_ptr(),
                                            pw.numel(), st), None)
    torch.cuda.synchronize()
    errs["final_fwd"] = rel(y3, torch.sigmoid(F.conv2d(x.double(), w2.double(), b2.double(), padding=3)))
    print(f"thin 7x7 {B}x{H}x{W}x{N}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(v <= 2e-5 for v in errs.values()), errs
