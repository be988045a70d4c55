"""Differentiable forward of OcclusionAwareGenerator -- SURVEY.md section 8f row N4, the backward half: what
``loss.backward()`` needs in the reference's fine-tuning loop (train.py:133; the generator stays in training mode there and the
loss reaches both the generator's parameters and, through the flow, the audio-driven key points).  ``.eval()`` with gradients
enabled takes the same composition with every BatchNorm on its running statistics (sync_batchnorm/batchnorm.py:48-53), so the
output is differentiable in evaluation mode as the reference module's is.

The reference composes ``nn.Conv2d`` / ``SynchronizedBatchNorm2d`` / ``F.grid_sample`` modules and lets autograd differentiate
them (modules/generator.py:59-97, dense_motion.py:32-113, util.py:858-1002).  Here the same composition is built from this
package's operators, each a ``torch.autograd.Function`` whose forward AND backward are libeamm_hip.so kernels:

* every convolution          ``autograd_ops.conv2d_same_nhwc``  (fp32-MFMA forward / data gradient / weight gradient); the two 7x7
                             layers with three channels on one side ``first_conv7`` / ``final_conv7_sigmoid`` (no padding of
                             the three channels to 32)
* every BatchNorm + ReLU     ``sync_batchnorm._BatchNormNHWCFunction``  (batch statistics, replicas' all-reduce, the block's ReLU
                             and DownBlock2d's 2x2 average fused; backward with the mask recomputed)
* every bilinear warp        ``autograd_ops.warp_nhwc``     (feature warp x occlusion, the K+1 sparse warps, ``deformed``)

Activations stay NHWC -- the kernels' layout -- from the source image to the prediction; the remaining steps are element-wise
or a few hundred floats (nearest x2, residual add, softmax over the K+1 motions, sigmoid, heat-maps, 2x2 jacobian algebra,
channel padding to the kernels' 32-channel granule, anti-aliasing as two banded GEMMs) and stay torch-ROCm ops with their
own autograd.  This is the OP-LEVEL composition: it exists so that the backward kernels are exercised and verified in
the generator's real data flow (tests/test_train_backward.py: gradients against the reference's autograd fixtures; the
operators one by one: tests/test_gpu_backward.py); it
re-packs filters per call and is not the tuned path -- the inference engine and the resumable training forward are.
GPU only: no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import autograd_ops
from .sync_batchnorm import SynchronizedBatchNorm2d, _BatchNormNHWCFunction

_G = 32   # channel granule of the convolution kernels


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _pad_last(x: torch.Tensor, mult: int) -> torch.Tensor:
    c = x.shape[-1]
    return x if c % mult == 0 else F.pad(x, (0, _round_up(c, mult) - c))


def conv(x: torch.Tensor, mod: torch.nn.Conv2d, feeds_norm: bool = False, keep_padded: bool = False) -> torch.Tensor:
    """``mod(x)`` for the path's stride-1 "same" 3x3 / 7x7 convolutions on NHWC [B,H,W,Cin] -> NHWC [B,H,W,Cout]; channels
    zero-padded to the kernels' granule (the padded filter rows / columns are zeros and the slice drops their gradient).  An input
    that already carries zero channels up to the granule is taken as it is; ``keep_padded`` returns the granule-padded output
    (channels >= Cout are exactly zero... plus nothing: their filters and biases are zeros)."""
    return conv_wb(x, mod.weight, mod.bias, feeds_norm, keep_padded)


def conv_wb(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], feeds_norm: bool = False, keep_padded: bool = False) -> torch.Tensor:
    """``conv`` on a filter / bias pair that is not a module's (the flow head's two stacked convolutions)."""
    cout, cin = w.shape[:2]
    cin_p, cout_p = _round_up(cin, _G), _round_up(cout, _G)
    if cin_p != cin:
        x = _pad_last(x, _G)
        w = F.pad(w, (0, 0, 0, 0, 0, cin_p - cin))
    if cout_p != cout:
        w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_p - cout))
        b = F.pad(b, (0, cout_p - cout)) if b is not None else None
    # feeds_norm: the output goes straight into a batch-statistics BatchNorm -- the bias gradient is exactly zero
    y = autograd_ops.conv2d_same_nhwc(x, w, b, bias_grad_is_zero=feeds_norm)
    return y[..., :cout] if (cout_p != cout and not keep_padded) else y


def warp(features: torch.Tensor, deformation: torch.Tensor, occlusion=None) -> torch.Tensor:
    """generator.py:50-57 (+ :79-84 with an occlusion map) on NHWC features; flow [n,h',w',2] and occlusion [n,h',w'] are
    resized bilinearly to the features first, as the reference does."""
    h, w = features.shape[1:3]
    if deformation.shape[1:3] != (h, w):
        deformation = F.interpolate(deformation.permute(0, 3, 1, 2), size=(h, w), mode="bilinear",
                                    align_corners=False).permute(0, 2, 3, 1)
    if occlusion is not None and occlusion.shape[1:3] != (h, w):
        occlusion = F.interpolate(occlusion[:, None], size=(h, w), mode="bilinear", align_corners=False)[:, 0]
    c = features.shape[3]
    out = autograd_ops.warp_nhwc(_pad_last(features, 8), deformation, occlusion)
    return out[..., :c] if c % 8 else out


def _upsample2(x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(scale_factor=2), nearest (util.py:896), on NHWC: one broadcast copy."""
    b, h, w, c = x.shape
    return x[:, :, None, :, None, :].expand(b, h, 2, w, 2, c).reshape(b, 2 * h, 2 * w, c)


class _Graph:
    """One forward's bookkeeping: the generator (parameter holder), its BatchNorm adapters, the replicas' settings.
    Activations are NHWC [B,H,W,C] between the operators -- the convolution and warp kernels' own layout."""

    def __init__(self, gen):
        self.gen = gen
        self.training = bool(gen.training)

    def norm_relu(self, x: torch.Tensor, holder: torch.nn.BatchNorm2d, pool: bool = False) -> torch.Tensor:
        """[avgpool2x2](relu(BatchNorm(x))) in one fused operator (every BatchNorm of the generator is followed by a ReLU)."""
        if x.shape[3] != holder.num_features:
            raise RuntimeError(f"expected {holder.num_features} channels, got {x.shape[3]}")
        if not self.training:
            # evaluation mode (batchnorm.py:48-53): a per-channel affine map from the running statistics -- a few element-wise
            # torch ops with their own autograd (differentiable in x, weight and bias; the statistics are buffers)
            scale = holder.weight * torch.rsqrt(holder.running_var + holder.eps)
            y = torch.relu(x * scale + (holder.bias - holder.running_mean * scale))
            if pool:
                b, h, w, c = y.shape
                y = y.view(b, h // 2, 2, w // 2, 2, c).mean(dim=(2, 4))
            return y
        # the holder owns the tensors (state_dict names of the reference); the adapter lends them to the Function.  The
        # adapter lives ON the holder (not in a table keyed by id(holder): ids go stale after deepcopy / are reused)
        a = holder.__dict__.get("_eamm_adapter")
        if a is None or a.num_features != holder.num_features:
            a = SynchronizedBatchNorm2d(holder.num_features, eps=holder.eps, momentum=holder.momentum)
            holder.__dict__["_eamm_adapter"] = a
        a.eps, a.momentum = holder.eps, holder.momentum
        a._parameters["weight"], a._parameters["bias"] = holder.weight, holder.bias
        a._buffers["running_mean"], a._buffers["running_var"] = holder.running_mean, holder.running_var
        a.training = True
        a.process_group, a.sync = self.gen.process_group, self.gen.sync_batchnorm
        a._ops.check(x, a)
        if a._replicas() == 1 and x.numel() // x.shape[3] <= 1:      # batchnorm.py:112
            raise AssertionError("BatchNorm computes unbiased standard-deviation, which requires size > 1.")
        return _BatchNormNHWCFunction.apply(x.contiguous(), holder.weight, holder.bias, a, True, pool)

    # ---- blocks: modules/util.py:858-938 --------------------------------------------------------------------------------
    # (feeds_norm -- "the bias gradient is exactly zero" -- holds for BATCH statistics only: training mode)
    def same_block(self, x, blk):
        return self.norm_relu(conv(x, blk.conv, feeds_norm=self.training), blk.norm)

    def down_block(self, x, blk):
        return self.norm_relu(conv(x, blk.conv, feeds_norm=self.training), blk.norm, pool=True)

    def up_block(self, x, blk):
        return self.same_block(_upsample2(x), blk)

    def res_block(self, x, blk):
        y = conv(self.norm_relu(x, blk.norm1), blk.conv1, feeds_norm=self.training)      # conv1 -> norm2
        y = conv(self.norm_relu(y, blk.norm2), blk.conv2)
        return y + x

    def hourglass(self, x, hg):                                    # util.py:941-1002
        skips = [x]
        for blk in hg.encoder.down_blocks:
            skips.append(self.down_block(skips[-1], blk))
        out = skips.pop()
        for blk in hg.decoder.up_blocks:
            out = torch.cat([self.up_block(out, blk), skips.pop()], dim=3)
        return out


def _grid(h: int, w: int, like: torch.Tensor) -> torch.Tensor:
    """util.py:839-855: [h,w,2], last dim (x, y) in [-1, 1]."""
    xs = 2 * (torch.arange(w, device=like.device).to(like.dtype) / (w - 1)) - 1
    ys = 2 * (torch.arange(h, device=like.device).to(like.dtype) / (h - 1)) - 1
    return torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=2)


def _heatmaps(value: torch.Tensor, grid: torch.Tensor, variance: float) -> torch.Tensor:
    d = grid[None, None] - value[:, :, None, None, :]              # util.py:815-836
    return torch.exp(-0.5 * (d * d).sum(-1) / variance)


_RANK_ONE: Dict[tuple, bool] = {}


def _band(taps: torch.Tensor, n_in: int, step: int) -> torch.Tensor:
    """[n_in, n_out] matrix of a 1-D correlation with `taps`, zero padding len(taps) // 2, every step-th output kept."""
    ka = taps.numel() // 2
    n_out = (n_in + step - 1) // step
    i = torch.arange(n_in, device=taps.device)[:, None]
    o = torch.arange(n_out, device=taps.device)[None, :]
    t = i - step * o + ka
    ok = (t >= 0) & (t < taps.numel())
    return torch.where(ok, taps[t.clamp(0, taps.numel() - 1)], torch.zeros((), dtype=taps.dtype, device=taps.device))


def _antialias_down(x: torch.Tensor, weight: torch.Tensor, scale: float) -> torch.Tensor:
    """AntiAliasInterpolation2d (util.py:1005-1052): zero-pad, depth-wise Gaussian, keep every (1/scale)-th row and column.
    The Gaussian is an outer product (util.py:1024-1036), so the filter-and-subsample is two small dense matrix products
    (rows, then columns) -- rocBLAS GEMMs with their own autograd instead of a 169-tap depth-wise convolution that computes
    15/16 outputs only to drop them.  A buffer that is not rank one (never the reference's) takes the convolution."""
    step = int(1 / scale)
    k = weight[0, 0]
    gy, gx = k.sum(dim=1), k.sum(dim=0)
    key = (weight.data_ptr(), weight._version, weight.device)
    if key not in _RANK_ONE:     # checked once per buffer (a host read)
        _RANK_ONE.clear()
        same = weight.shape[0] == 1 or bool(torch.equal(weight[0], weight[-1]))
        _RANK_ONE[key] = same and float((torch.outer(gy, gx) - k).abs().max()) <= 1e-6 * float(k.abs().max())
    if not _RANK_ONE[key]:
        ka = weight.shape[-1] // 2
        return F.conv2d(F.pad(x, (ka, ka, ka, ka)), weight, groups=x.shape[1])[:, :, ::step, ::step]
    my, mx = _band(gy, x.shape[2], step), _band(gx, x.shape[3], step)
    return torch.matmul(my.t(), torch.matmul(x, mx))


def _dense_motion(g: _Graph, source_image, kp_driving, kp_source):
    """dense_motion.py:32-113.  source_image NCHW as the caller passed it.  Round 4: anti-aliasing, key-point records, heat-maps +
    sparse motions + the K+1 warps, and the softmax / flow / sigmoid head are HIP operators with HIP backward (``motion_ops``); the
    torch composition below remains for configurations those kernels do not cover (scale_factor other than 0.25 / 1, more than
    three image channels)."""
    dm = g.gen.dense_motion_network
    inv = 1.0 / dm.scale_factor
    if os.environ.get("EAMM_MOTION_TORCH") != "1" and source_image.shape[1] == 3 and abs(inv - round(inv)) < 1e-6 and int(round(inv)) in (1, 4) and dm.num_kp <= 31 and \
            source_image.shape[2] % int(round(inv)) == 0 and source_image.shape[3] % int(round(inv)) == 0:
        from . import motion_ops
        small = motion_ops.antialias_down(source_image, dm.down.weight if dm.scale_factor != 1 else None, dm.scale_factor)
        rec = motion_ops.kp_records(kp_driving, kp_source)
        # hourglass input [B,h,w,64]: 4 (K+1) real channels + zeros up to the kernels' granule -- the convolutions take it as it is
        hg_in, sparse = motion_ops.motion_front(rec, small, dm.kp_variance, _round_up(4 * (dm.num_kp + 1), _G))
        if dm.hourglass.decoder.up_blocks[-1].conv.weight.shape[0] % _G:   # the last concatenation would not end on the granule
            hg_in = hg_in[..., :4 * (dm.num_kp + 1)]
        feat = g.hourglass(hg_in, dm.hourglass)
        if dm.occlusion is not None and dm.mask.bias is not None and dm.occlusion.bias is not None:
            # dense_motion.py:98,110: the mask and occlusion 7x7 convolutions read the same features -- run as ONE convolution with
            # the filters stacked (K + 2 of the granule's 32 output channels instead of K + 1 and 1 of 32 each; the engine does the
            # same); autograd splits the stacked filter's gradient back onto the two modules
            logits = conv_wb(feat, torch.cat([dm.mask.weight, dm.occlusion.weight], dim=0), torch.cat([dm.mask.bias, dm.occlusion.bias]),
                             keep_padded=True)
            mask, deformation, occ = motion_ops.motion_head(logits, None, rec, stacked=True)
        else:
            lo = conv(feat, dm.occlusion, keep_padded=True) if dm.occlusion is not None else None
            mask, deformation, occ = motion_ops.motion_head(conv(feat, dm.mask, keep_padded=True), lo, rec)
        out = {"sparse_deformed": sparse, "mask": mask, "deformation": deformation}
        if occ is not None:
            out["occlusion_map"] = occ
        return out
    return _dense_motion_torch(g, source_image, kp_driving, kp_source)


def _dense_motion_torch(g: _Graph, source_image, kp_driving, kp_source):
    """The same stage as torch-ROCm ops with torch's own autograd (rounds 3's composition)."""
    dm = g.gen.dense_motion_network
    src = _antialias_down(source_image, dm.down.weight, dm.scale_factor) if dm.scale_factor != 1 else source_image
    b, c, h, w = src.shape
    k = dm.num_kp
    grid = _grid(h, w, src)
    heat = _heatmaps(kp_driving["value"], grid, dm.kp_variance) - _heatmaps(kp_source["value"], grid, dm.kp_variance)
    heat = torch.cat([torch.zeros_like(heat[:, :1]), heat], dim=1)                         # [B,K+1,h,w]
    rel = grid[None, None] - kp_driving["value"][:, :, None, None, :]                      # dense_motion.py:47-67
    if "jacobian" in kp_driving:
        jac = torch.matmul(kp_source["jacobian"], torch.inverse(kp_driving["jacobian"]))   # [B,K,2,2]
        j = jac[:, :, None, None]                                                           # 2x2 times a vector, element-wise
        rel = torch.stack([j[..., 0, 0] * rel[..., 0] + j[..., 0, 1] * rel[..., 1],
                           j[..., 1, 0] * rel[..., 0] + j[..., 1, 1] * rel[..., 1]], dim=-1)
    moved = rel + kp_source["value"][:, :, None, None, :]
    motions = torch.cat([grid[None, None].expand(b, 1, h, w, 2), moved], dim=1)            # [B,K+1,h,w,2]
    src_nhwc = src.permute(0, 2, 3, 1)                                                      # dense_motion.py:69-79
    rep = src_nhwc[:, None].expand(b, k + 1, h, w, c).reshape(b * (k + 1), h, w, c)
    warped = warp(rep, motions.reshape(b * (k + 1), h, w, 2)).view(b, k + 1, h, w, c)      # NHWC per (pair, motion)
    # hourglass input: channel (k, j) with j = 0 the heat-map, 1..c the warped source (dense_motion.py:93-94)
    hg_in = torch.cat([heat.permute(0, 2, 3, 1)[..., None], warped.permute(0, 2, 3, 1, 4)], dim=4).reshape(b, h, w, (k + 1) * (c + 1))
    feat = g.hourglass(hg_in, dm.hourglass)
    mask = F.softmax(conv(feat, dm.mask), dim=3)                                           # [B,h,w,K+1]  dense_motion.py:98-99
    deformation = (motions * mask.permute(0, 3, 1, 2)[..., None]).sum(dim=1)               # [B,h,w,2]    :101-104
    out = {"sparse_deformed": warped.permute(0, 1, 4, 2, 3), "mask": mask.permute(0, 3, 1, 2), "deformation": deformation}
    if dm.occlusion is not None:
        out["occlusion_map"] = torch.sigmoid(conv(feat, dm.occlusion))[..., 0]             # [B,h,w]
    return out


def forward_train(gen, source_image: torch.Tensor, kp_driving, kp_source) -> Dict[str, torch.Tensor]:
    """OcclusionAwareGenerator.forward WITH an autograd graph (generator.py:59-97); NCHW in and out.  ``gen.training`` picks
    batch statistics (+ running-statistics update) or running statistics in every BatchNorm."""
    if source_image.device.type != "cuda":
        raise RuntimeError("eamm_amd.OcclusionAwareGenerator runs only on a ROCm GPU (there is no CPU fallback for this path)")
    g = _Graph(gen)
    src_nhwc = source_image.permute(0, 2, 3, 1)
    fc = gen.first.conv                                                                    # generator.py:61-63
    if fc.weight.shape[1] == 3 and fc.weight.shape[0] in (32, 64) and fc.bias is not None:
        out = g.norm_relu(autograd_ops.first_conv7(F.pad(src_nhwc, (0, 1)), fc.weight, fc.bias, bias_grad_is_zero=g.training), gen.first.norm)
    else:
        out = g.same_block(src_nhwc, gen.first)
    for blk in gen.down_blocks:
        out = g.down_block(out, blk)
    result = {}
    if gen.dense_motion_network is not None:                                               # generator.py:64-86
        dmo = _dense_motion(g, source_image, kp_driving, kp_source)
        result["mask"], result["sparse_deformed"] = dmo["mask"].contiguous(), dmo["sparse_deformed"].contiguous()
        occ = dmo.get("occlusion_map")
        if occ is not None:
            result["occlusion_map"] = occ[:, None]
        out = warp(out, dmo["deformation"], occ)
        result["deformed"] = warp(src_nhwc, dmo["deformation"]).permute(0, 3, 1, 2).contiguous()
    for blk in gen.bottleneck:                                                             # generator.py:89-93
        out = g.res_block(out, blk)
    for blk in gen.up_blocks:
        out = g.up_block(out, blk)
    fin = gen.final
    if fin.weight.shape[0] == 3 and out.shape[3] in (32, 64) and fin.bias is not None:
        result["prediction"] = autograd_ops.final_conv7_sigmoid(out, fin.weight, fin.bias)
    else:
        result["prediction"] = torch.sigmoid(conv(out, fin)).permute(0, 3, 1, 2).contiguous()
    return result
