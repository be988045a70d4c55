"""Differentiable dense-motion front end and flow head (round 4) -- ``torch.autograd.Function``s
whose forward AND backward are libeamm_hip.so kernels (csrc/motion.hip, csrc/motion_backward.hip; C ABI ``eamm_op_antialias_down``,
``eamm_op_kp_records``, ``eamm_op_motion_front``, ``eamm_op_motion_head`` and their ``_backward`` entries).

They replace, inside the differentiable generator forward (``train_graph``), the torch-ROCm ops that used to carry the
reference's modules/dense_motion.py:32-113 and modules/util.py:1005-1052 with torch's own autograd: the anti-alias
down-sampling (two rocBLAS GEMMs), the heat-maps and 2x2 jacobian algebra (ATen element-wise kernels + torch.inverse), the K+1
sparse warps, the softmax / flow combine / sigmoid.  GPU only: no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .autograd_ops import _need_gpu, _ptr, _stream


def _check(code):
    _lib.check(code, None)


class _AntiAliasDownFunction(torch.autograd.Function):
    """source NCHW [B,3,H,W], buffer [3,1,13,13] -> NHWC [B,H/s,W/s,4] (RGB + a zero channel)."""

    @staticmethod
    def forward(ctx, source, weight, inv_scale):
        b, _, hh, ww = source.shape
        small = torch.empty(b, hh // inv_scale, ww // inv_scale, 4, dtype=torch.float32, device=source.device)
        with torch.cuda.device(source.device):
            _check(_lib.lib().eamm_op_antialias_down(source.device.index, _ptr(source), _ptr(weight), b, hh, ww, inv_scale, _ptr(small),
                                                     _stream(source.device)))
        ctx.save_for_backward(weight)
        ctx.shape, ctx.inv_scale = (b, hh, ww), inv_scale
        return small

    @staticmethod
    def backward(ctx, grad_small):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        (weight,) = ctx.saved_tensors
        b, hh, ww = ctx.shape
        grad_small = grad_small.contiguous()
        grad = torch.empty(b, 3, hh, ww, dtype=torch.float32, device=grad_small.device)
        with torch.cuda.device(grad.device):
            _check(_lib.lib().eamm_op_antialias_down_backward(grad.device.index, _ptr(grad_small), _ptr(weight), b, hh, ww, ctx.inv_scale,
                                                              _ptr(grad), _stream(grad.device)))
        return grad, None, None


def antialias_down(source: torch.Tensor, weight: Optional[torch.Tensor], scale_factor: float) -> torch.Tensor:
    """AntiAliasInterpolation2d (reference modules/util.py:1005-1052) on an RGB batch -> NHWC [B,h,w,4]; ``scale_factor`` 0.25 (the
    13x13 Gaussian of every shipped configuration) or 1 (util.py:1047-1048 returns the input)."""
    _need_gpu(source, "antialias_down")
    inv = int(round(1.0 / scale_factor))
    if source.dim() != 4 or source.shape[1] != 3 or inv not in (1, 4) or source.shape[2] % inv or source.shape[3] % inv:
        raise ValueError(f"antialias_down: RGB NCHW input with sides multiples of 1/scale_factor, scale_factor 0.25 or 1; got "
                         f"{tuple(source.shape)}, {scale_factor}")
    if inv == 4 and (weight is None or tuple(weight.shape) != (3, 1, 13, 13)):
        raise ValueError("antialias_down: the module's [3,1,13,13] Gaussian buffer is required")
    return _AntiAliasDownFunction.apply(source.contiguous(), None if weight is None else weight.contiguous(), inv)


_SINGULAR = "kp_records: a driving jacobian is singular (torch.inverse would raise, dense_motion.py:56)"


class _KpRecordsFunction(torch.autograd.Function):
    """(kp_driving value, jacobian, kp_source value, jacobian) -> records [n,K,8]: kd.xy, ks.xy, J = Js inverse(Jd)."""

    @staticmethod
    def forward(ctx, kd_val, kd_jac, ks_val, ks_jac):
        n, k = kd_val.shape[:2]
        rec = torch.empty(n, k, 8, dtype=torch.float32, device=kd_val.device)
        flag = torch.zeros(1, dtype=torch.int32, device=kd_val.device)
        with torch.cuda.device(kd_val.device):
            _check(_lib.lib().eamm_op_kp_records(kd_val.device.index, _ptr(kd_val), _ptr(kd_jac), _ptr(ks_val), _ptr(ks_jac), n, k,
                                                 _ptr(rec), _ptr(flag), _stream(kd_val.device)))
        # torch.inverse raises (and synchronises) on a singular matrix too -- but a host read is illegal while the stream is being
        # captured into a graph: there the flag stays on the device and backward() (or the replay's caller) reads it
        ctx.flag = None
        if kd_jac is not None:
            if torch.cuda.is_current_stream_capturing():
                ctx.flag = flag
            elif int(flag.item()):
                raise RuntimeError(_SINGULAR)
        ctx.save_for_backward(kd_jac, ks_jac)
        return rec

    @staticmethod
    def backward(ctx, grad_rec):
        kd_jac, ks_jac = ctx.saved_tensors
        if ctx.flag is not None and not torch.cuda.is_current_stream_capturing() and int(ctx.flag.item()):
            raise RuntimeError(_SINGULAR)
        n, k = grad_rec.shape[:2]
        grad_rec = grad_rec.contiguous()
        need = ctx.needs_input_grad
        dev = grad_rec.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        g_kdv = new(n, k, 2) if need[0] else None
        g_kdj = new(n, k, 2, 2) if (need[1] and kd_jac is not None) else None
        g_ksv = new(n, k, 2) if need[2] else None
        g_ksj = new(n, k, 2, 2) if (need[3] and ks_jac is not None) else None
        if g_kdv is None and g_kdj is None and g_ksv is None and g_ksj is None:
            return None, None, None, None
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_op_kp_records_backward(dev.index, _ptr(kd_jac), _ptr(ks_jac), _ptr(grad_rec), n, k, _ptr(g_kdv),
                                                          _ptr(g_ksv), _ptr(g_kdj), _ptr(g_ksj), _stream(dev)))
        return g_kdv, g_kdj, g_ksv, g_ksj


def kp_records(kp_driving: dict, kp_source: dict) -> torch.Tensor:
    """dense_motion.py:47-67: the per key point affine map T_k(z) = J_k (z - kp_driving_k) + kp_source_k as records [n,K,8]."""
    kd_val, ks_val = kp_driving["value"], kp_source["value"]
    _need_gpu(kd_val, "kp_records")
    kd_jac = kp_driving.get("jacobian") if "jacobian" in kp_driving else None
    ks_jac = kp_source.get("jacobian") if kd_jac is not None else None
    if kd_val.shape != ks_val.shape or kd_val.dim() != 3 or kd_val.shape[2] != 2:
        raise ValueError(f"kp_records: key-point values must both be [n,K,2], got {tuple(kd_val.shape)} / {tuple(ks_val.shape)}")
    cont = lambda t: None if t is None else t.contiguous()
    return _KpRecordsFunction.apply(cont(kd_val), cont(kd_jac), cont(ks_val), cont(ks_jac))


def _workspace(n, k, h, w, dev):
    return torch.empty(max(1, _lib.lib().eamm_op_motion_workspace_floats(n, k, h, w)), dtype=torch.float32, device=dev)


class _MotionFrontFunction(torch.autograd.Function):
    """records [n,K,8], small NHWC [n,h,w,4] -> hourglass input NHWC [n,h,w,Cpad], sparse_deformed [n,K+1,3,h,w]."""

    @staticmethod
    def forward(ctx, rec, small, variance, cpad):
        n, k = rec.shape[:2]
        h, w = small.shape[1:3]
        dev = rec.device
        hg = torch.empty(n, h, w, cpad, dtype=torch.float32, device=dev)
        sd = torch.empty(n, k + 1, 3, h, w, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_op_motion_front(dev.index, _ptr(rec), _ptr(small), n, k, h, w, float(variance), cpad, _ptr(hg), _ptr(sd),
                                                   _stream(dev)))
        ctx.save_for_backward(rec, small)
        ctx.variance, ctx.cpad = float(variance), cpad
        return hg, sd

    @staticmethod
    def backward(ctx, grad_hg, grad_sd):
        rec, small = ctx.saved_tensors
        n, k = rec.shape[:2]
        h, w = small.shape[1:3]
        dev = rec.device
        if grad_hg is None and grad_sd is None:
            return None, None, None, None
        grad_hg = None if grad_hg is None else grad_hg.contiguous()
        grad_sd = None if grad_sd is None else grad_sd.contiguous()
        g_rec = torch.empty_like(rec)
        g_small = torch.empty_like(small) if ctx.needs_input_grad[1] else None
        ws = _workspace(n, k, h, w, dev)
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_op_motion_front_backward(dev.index, _ptr(rec), _ptr(small), n, k, h, w, ctx.variance, ctx.cpad,
                                                            _ptr(grad_hg), _ptr(grad_sd), _ptr(g_small), _ptr(g_rec), _ptr(ws), ws.numel(),
                                                            _stream(dev)))
        return (g_rec if ctx.needs_input_grad[0] else None), g_small, None, None


def motion_front(records: torch.Tensor, small: torch.Tensor, kp_variance: float, cpad: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """dense_motion.py:32-45, 47-79, 93-94: heat-maps, sparse motions and the K+1 warps of the down-sampled source, laid out as the
    hourglass input (channel 4k = heat-map k, 4k+1..3 = RGB warped by motion k; zero padded to ``cpad``) + ``sparse_deformed``."""
    _need_gpu(records, "motion_front")
    n, k = records.shape[:2]
    if tuple(small.shape[:1]) != (n,) or small.dim() != 4 or small.shape[3] != 4 or cpad < 4 * (k + 1) or cpad % 4:
        raise ValueError(f"motion_front: records {tuple(records.shape)} / small {tuple(small.shape)} / cpad {cpad} mismatch")
    return _MotionFrontFunction.apply(records.contiguous(), small.contiguous(), kp_variance, int(cpad))


class _MotionHeadFunction(torch.autograd.Function):
    """mask logits NHWC [n,h,w,ld], occlusion logits NHWC [n,h,w,ldo] or None, records -> mask [n,K+1,h,w], deformation [n,h,w,2],
    occlusion [n,h,w] (or None)."""

    @staticmethod
    def forward(ctx, lm, lo, rec, stacked=False):
        """stacked: the occlusion logit is channel K + 1 of ``lm`` (one convolution produced both); ``lo`` is None then."""
        n, h, w, ld = lm.shape
        k = rec.shape[1]
        dev = lm.device
        mask = torch.empty(n, k + 1, h, w, dtype=torch.float32, device=dev)
        defo = torch.empty(n, h, w, 2, dtype=torch.float32, device=dev)
        occ = torch.empty(n, h, w, dtype=torch.float32, device=dev) if (lo is not None or stacked) else None
        lo_ptr = C.c_void_p(lm.data_ptr() + 4 * (k + 1)) if stacked else _ptr(lo)
        ldo = ld if stacked else (0 if lo is None else lo.shape[3])
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_op_motion_head(dev.index, _ptr(lm), ld, lo_ptr, ldo, _ptr(rec), n, k, h, w,
                                                  _ptr(mask), _ptr(defo), _ptr(occ), _stream(dev)))
        ctx.save_for_backward(mask, occ, rec)
        ctx.ld, ctx.ldo, ctx.stacked = ld, ldo, bool(stacked)
        return mask, defo, occ

    @staticmethod
    def backward(ctx, g_mask, g_defo, g_occ):
        mask, occ, rec = ctx.saved_tensors
        n, k1, h, w = mask.shape
        k = k1 - 1
        dev = mask.device
        cont = lambda t: None if t is None else t.contiguous()
        g_mask, g_defo, g_occ = cont(g_mask), cont(g_defo), cont(g_occ)
        g_lm = torch.empty(n, h, w, ctx.ld, dtype=torch.float32, device=dev)
        g_lo = torch.empty(n, h, w, ctx.ldo, dtype=torch.float32, device=dev) if (occ is not None and not ctx.stacked) else None
        g_lo_ptr = C.c_void_p(g_lm.data_ptr() + 4 * (k + 1)) if ctx.stacked else _ptr(g_lo)
        g_rec = torch.empty_like(rec)
        ws = _workspace(n, k, h, w, dev)
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_op_motion_head_backward(dev.index, _ptr(mask), _ptr(occ), _ptr(rec), n, k, h, w, _ptr(g_mask), _ptr(g_defo),
                                                           _ptr(g_occ), _ptr(g_lm), ctx.ld, g_lo_ptr, ctx.ldo, _ptr(g_rec), _ptr(ws),
                                                           ws.numel(), _stream(dev)))
        return g_lm, g_lo, (g_rec if ctx.needs_input_grad[2] else None), None


def motion_head(mask_logits: torch.Tensor, occlusion_logits: Optional[torch.Tensor], records: torch.Tensor, stacked: bool = False):
    """dense_motion.py:98-111 on the two 7x7 convolutions' NHWC outputs (channels 0..K of ``mask_logits``, channel 0 of
    ``occlusion_logits``): mask [n,K+1,h,w], deformation [n,h,w,2], occlusion [n,h,w] or None.  ``stacked``: ONE convolution with
    the two filters stacked produced both -- the occlusion logit is channel K + 1 of ``mask_logits``, ``occlusion_logits`` is None."""
    _need_gpu(mask_logits, "motion_head")
    n, k = records.shape[:2]
    if mask_logits.dim() != 4 or mask_logits.shape[0] != n or mask_logits.shape[3] < k + 1 + int(stacked):
        raise ValueError(f"motion_head: mask logits {tuple(mask_logits.shape)} do not hold {k + 1} motions of {n} frames")
    if stacked and occlusion_logits is not None:
        raise ValueError("motion_head: stacked logits carry the occlusion logit in channel K + 1; pass occlusion_logits=None")
    if occlusion_logits is not None and tuple(occlusion_logits.shape[:3]) != tuple(mask_logits.shape[:3]):
        raise ValueError("motion_head: occlusion logits and mask logits differ in shape")
    return _MotionHeadFunction.apply(mask_logits.contiguous(), None if occlusion_logits is None else occlusion_logits.contiguous(),
                                     records.contiguous(), stacked)
