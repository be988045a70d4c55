"""The differentiable dense-motion operators (eamm_amd/motion_ops.py; csrc/motion.hip forward, csrc/motion_backward.hip backward;
VERDICT r03 item 7) one by one against what torch derives in DOUBLE from the reference's own statements
(modules/dense_motion.py:32-113, modules/util.py:815-855, 1005-1052): forward values and every gradient."""
import pytest
import torch
import torch.nn.functional as F

from eamm_amd.weights import antialias_kernel, synthetic_keypoints, synthetic_source

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grid(h, w, dtype):                                   # util.py:839-855
    xs = 2 * (torch.arange(w, dtype=dtype) / (w - 1)) - 1
    ys = 2 * (torch.arange(h, dtype=dtype) / (h - 1)) - 1
    return torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=2)


def _close(got, want, tol, what):
    err = float((got.detach().cpu().double() - want.detach().double()).abs().max())
    scale = max(1.0, float(want.detach().abs().max()))
    assert err <= tol * scale, (what, err, scale)
    return err / scale


@pytest.mark.parametrize("scale", [0.25, 1])
def test_antialias_down_forward_and_backward(scale):
    from eamm_amd import motion_ops
    src = synthetic_source(64, seed=3, batch=2)
    w = antialias_kernel(3)
    xr = src.double().requires_grad_()
    if scale == 1:
        ref = xr
    else:   # util.py:1044-1052
        ref = F.conv2d(F.pad(xr, (6, 6, 6, 6)), w.double(), groups=3)[:, :, ::4, ::4]
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    (ref * g.double()).sum().backward()
    xd = src.to(DEV).requires_grad_()
    out = motion_ops.antialias_down(xd, w.to(DEV) if scale != 1 else None, scale)
    assert out.shape == (2, ref.shape[2], ref.shape[3], 4) and float(out[..., 3].abs().max()) == 0.0
    _close(out[..., :3].permute(0, 3, 1, 2), ref, 2e-6, "antialias forward")
    (out[..., :3].permute(0, 3, 1, 2) * g.to(DEV)).sum().backward()
    _close(xd.grad, xr.grad, 2e-6, "antialias backward")


def _reference_front(kd, ks, small_nchw, variance, with_jac):
    """dense_motion.py:32-79, 93-94 in torch: hourglass input [n,(K+1)*4,h,w] (channel-major, as the reference) + sparse_deformed."""
    n, k = kd["value"].shape[:2]
    h, w = small_nchw.shape[2:]
    dt = small_nchw.dtype
    grid = _grid(h, w, dt)
    heat = lambda v: torch.exp(-0.5 * ((grid[None, None] - v[:, :, None, None, :]) ** 2).sum(-1) / variance)
    hm = heat(kd["value"]) - heat(ks["value"])
    hm = torch.cat([torch.zeros(n, 1, h, w, dtype=dt), hm], dim=1)
    rel = grid[None, None] - kd["value"][:, :, None, None, :]
    if with_jac:
        jac = torch.matmul(ks["jacobian"], torch.inverse(kd["jacobian"]))
        rel = torch.matmul(jac[:, :, None, None], rel[..., None])[..., 0]
    moved = rel + ks["value"][:, :, None, None, :]
    motions = torch.cat([grid[None, None].expand(n, 1, h, w, 2), moved], dim=1)
    rep = small_nchw[:, None].expand(n, k + 1, 3, h, w).reshape(n * (k + 1), 3, h, w)
    warped = F.grid_sample(rep, motions.reshape(n * (k + 1), h, w, 2), align_corners=False).view(n, k + 1, 3, h, w)
    hg = torch.cat([hm[:, :, None], warped], dim=2)          # [n,K+1,4,h,w]
    return hg, warped, motions


@pytest.mark.parametrize("with_jac", [True, False])
def test_kp_records_and_motion_front_against_torch_double(with_jac):
    from eamm_amd import motion_ops
    n, k, h, w, var = 3, 10, 16, 16, 0.01
    gen = torch.Generator().manual_seed(5)
    small = torch.rand(n, 3, h, w, generator=gen)
    kd, ks = synthetic_keypoints(n, k, seed=2, jacobian=with_jac), synthetic_keypoints(n, k, seed=0, jacobian=with_jac)
    kdr = {a: b.double().requires_grad_() for a, b in kd.items()}
    ksr = {a: b.double().requires_grad_() for a, b in ks.items()}
    sr = small.double().requires_grad_()
    hg_ref, sd_ref, _ = _reference_front(kdr, ksr, sr, var, with_jac)
    g_hg = torch.randn(hg_ref.shape, generator=gen)
    g_sd = torch.randn(sd_ref.shape, generator=gen)
    ((hg_ref * g_hg.double()).sum() + (sd_ref * g_sd.double()).sum()).backward()

    kdd = {a: b.to(DEV).requires_grad_() for a, b in kd.items()}
    ksd = {a: b.to(DEV).requires_grad_() for a, b in ks.items()}
    sd_in = F.pad(small.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV).requires_grad_()
    rec = motion_ops.kp_records(kdd, ksd)
    hg, sdef = motion_ops.motion_front(rec, sd_in, var, 64)
    assert hg.shape == (n, h, w, 64) and float(hg[..., 4 * (k + 1):].abs().max()) == 0.0
    got_hg = hg[..., :4 * (k + 1)].reshape(n, h, w, k + 1, 4).permute(0, 3, 4, 1, 2)
    _close(got_hg, hg_ref, 1e-5, "hourglass input")
    _close(sdef, sd_ref, 1e-5, "sparse_deformed")
    ((got_hg * g_hg.to(DEV)).sum() + (sdef * g_sd.to(DEV)).sum()).backward()
    for a in kd:
        e1 = _close(kdd[a].grad, kdr[a].grad, 2e-5, "d kp_driving." + a)
        e2 = _close(ksd[a].grad, ksr[a].grad, 2e-5, "d kp_source." + a)
        print(f"front backward (jacobians {with_jac}) {a}: driving {e1:.1e} source {e2:.1e}")
    _close(sd_in.grad[..., :3].permute(0, 3, 1, 2), sr.grad, 2e-5, "d small")
    assert float(sd_in.grad[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("with_occ", [True, False, "stacked"])
def test_motion_head_against_torch_double(with_occ):
    """"stacked": one convolution produced both heads -- the occlusion logit is channel K + 1 of the mask logits' rows."""
    from eamm_amd import motion_ops
    n, k, h, w = 2, 10, 16, 16
    gen = torch.Generator().manual_seed(9)
    kd, ks = synthetic_keypoints(n, k, seed=4), synthetic_keypoints(n, k, seed=6)
    lm = torch.randn(n, h, w, 32, generator=gen)
    lo = torch.randn(n, h, w, 32, generator=gen) if with_occ else None
    stacked = with_occ == "stacked"
    if stacked:
        lo[..., 0] = lm[..., k + 1]
    kdr = {a: b.double().requires_grad_() for a, b in kd.items()}
    ksr = {a: b.double().requires_grad_() for a, b in ks.items()}
    lmr = lm.double().requires_grad_()
    lor = lo.double().requires_grad_() if with_occ else None
    _, _, motions = _reference_front(kdr, ksr, torch.zeros(n, 3, h, w, dtype=torch.float64), 0.01, True)
    mask_ref = F.softmax(lmr[..., :k + 1].permute(0, 3, 1, 2), dim=1)                        # dense_motion.py:98-99
    def_ref = (motions.permute(0, 1, 4, 2, 3) * mask_ref[:, :, None]).sum(dim=1).permute(0, 2, 3, 1)   # :101-104
    g_m, g_d = torch.randn(mask_ref.shape, generator=gen), torch.randn(def_ref.shape, generator=gen)
    loss = (mask_ref * g_m.double()).sum() + (def_ref * g_d.double()).sum()
    if with_occ:
        occ_ref = torch.sigmoid(lor[..., 0])
        g_o = torch.randn(occ_ref.shape, generator=gen)
        loss = loss + (occ_ref * g_o.double()).sum()
    loss.backward()

    kdd = {a: b.to(DEV).requires_grad_() for a, b in kd.items()}
    ksd = {a: b.to(DEV).requires_grad_() for a, b in ks.items()}
    lmd = lm.to(DEV).requires_grad_()
    lod = lo.to(DEV).requires_grad_() if (with_occ and not stacked) else None
    rec = motion_ops.kp_records(kdd, ksd)
    mask, defo, occ = motion_ops.motion_head(lmd, lod, rec, stacked=stacked)
    _close(mask, mask_ref, 2e-6, "mask")
    _close(defo, def_ref, 2e-6, "deformation")
    out = (mask * g_m.to(DEV)).sum() + (defo * g_d.to(DEV)).sum()
    if with_occ:
        _close(occ, occ_ref, 2e-6, "occlusion")
        out = out + (occ * g_o.to(DEV)).sum()
    else:
        assert occ is None
    out.backward()
    _close(lmd.grad[..., :k + 1], lmr.grad[..., :k + 1], 1e-5, "d mask logits")
    if stacked:
        _close(lmd.grad[..., k + 1], lor.grad[..., 0], 1e-5, "d occlusion logit (stacked)")
        assert float(lmd.grad[..., k + 2:].abs().max()) == 0.0
    else:
        assert float(lmd.grad[..., k + 1:].abs().max()) == 0.0
    if with_occ and not stacked:
        _close(lod.grad[..., 0], lor.grad[..., 0], 1e-5, "d occlusion logits")
        assert float(lod.grad[..., 1:].abs().max()) == 0.0
    for a in kd:
        _close(kdd[a].grad, kdr[a].grad, 2e-5, "d kp_driving." + a)
        _close(ksd[a].grad, ksr[a].grad, 2e-5, "d kp_source." + a)


def test_kp_records_raise_on_a_singular_driving_jacobian():
    from eamm_amd import motion_ops
    kd, ks = synthetic_keypoints(2, 10, seed=2), synthetic_keypoints(2, 10, seed=0)
    kd["jacobian"][1, 3] = torch.tensor([[1.0, 2.0], [2.0, 4.0]])
    with pytest.raises(RuntimeError, match="singular"):
        motion_ops.kp_records({a: b.to(DEV) for a, b in kd.items()}, {a: b.to(DEV) for a, b in ks.items()})
