"""Differentiable op-level wrappers of the path's two operator kinds -- SURVEY.md section 8f row N4 (backward of the warp and
convolution kernels).  The reference gets these gradients from autograd (train.py:133 ``loss.backward()`` through
modules/generator.py:50-57, 79-84 and the ``nn.Conv2d`` of modules/util.py:858-938); here each is a ``torch.autograd.Function``
whose forward AND backward run in libeamm_hip.so:

* ``warp(features, deformation, occlusion)``  =  ``F.grid_sample(features, deformation) * occlusion``
  forward ``eamm_op_warp``, backward ``eamm_op_warp_backward`` (gradients of all three inputs);
* ``conv2d_same(x, weight, bias)``  =  ``F.conv2d(x, weight, bias, padding=k // 2)`` for the path's 3x3 / 7x7 filters
  forward ``eamm_op_conv_dev`` (parameters stay in HBM, the filter is packed on the device; F(4x4,3x3) Winograd where the
  bottleneck's kernel applies, else the implicit GEMM); data gradient = the same entry on the output gradient with the filter
  read transposed and flipped; weight / bias gradient ``eamm_op_conv_wgrad``.

Together with ``eamm_amd.SynchronizedBatchNorm2d`` (forward and backward) these are every kernel kind a backward pass of the
generator needs; ``eamm_amd/train_graph.py`` composes them into the generator's differentiable ``.train()`` forward.  They are
OP-LEVEL: every call packs its filter again (on the device) and allocates its outputs -- DESIGN.md section 8.  GPU only: no
CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

_CONV_BK = 32   # channel granule of the implicit-GEMM loaders (conv_common.h CONV_BK)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t: torch.Tensor, what: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"eamm_amd.autograd_ops.{what} runs only on a ROCm GPU (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what}: float32 tensors only, got {t.dtype}")


class _WarpFunction(torch.autograd.Function):
    """NHWC features [ns,h,w,C] (ns = 1 or n), flow [n,h,w,2], occlusion [n,h,w] or None -> NHWC [n,h,w,C]."""

    @staticmethod
    def forward(ctx, feat, deformation, occlusion):
        n, h, w = deformation.shape[:3]
        ns, c = feat.shape[0], feat.shape[3]
        out = torch.empty(n, h, w, c, dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().eamm_op_warp(feat.device.index, _ptr(feat), _ptr(deformation), _ptr(occlusion), n, ns, h, w, c, h, w,
                                               _ptr(out), 0, None, _stream(feat.device)), None)
        ctx.save_for_backward(feat, deformation, occlusion)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feat, deformation, occlusion = ctx.saved_tensors
        n, h, w = deformation.shape[:3]
        ns, c = feat.shape[0], feat.shape[3]
        grad_out = grad_out.contiguous()
        need = ctx.needs_input_grad
        gf = torch.empty_like(feat) if need[0] else None
        gd = torch.empty_like(deformation) if need[1] else None
        go = torch.empty_like(occlusion) if (occlusion is not None and need[2]) else None
        if gf is None and gd is None and go is None:
            return None, None, None
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().eamm_op_warp_backward(feat.device.index, _ptr(feat), _ptr(deformation), _ptr(occlusion),
                                                        _ptr(grad_out), n, ns, h, w, c, _ptr(gf), _ptr(gd), _ptr(go),
                                                        _stream(feat.device)), None)
        return gf, gd, go


def warp(features: torch.Tensor, deformation: torch.Tensor, occlusion: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.grid_sample(features, deformation) * occlusion`` (reference modules/generator.py:50-57, 79-84), differentiable.

    features NCHW [n,C,h,w] or [1,C,h,w] (one source for all frames); deformation [n,h,w,2]; occlusion [n,1,h,w] or None;
    flow and occlusion at the features' resolution (the reference's bilinear resize of a coarser flow stays a torch op in
    front of this one).  Returns NCHW [n,C,h,w]."""
    _need_gpu(features, "warp")
    n, h, w, two = deformation.shape
    if two != 2 or features.shape[2:] != (h, w) or features.shape[0] not in (1, n) or features.shape[1] % 8:
        raise ValueError(f"warp: features {tuple(features.shape)} / deformation {tuple(deformation.shape)} mismatch "
                         "(same resolution, channels a multiple of 8)")
    occ = None
    if occlusion is not None:
        if tuple(occlusion.shape) != (n, 1, h, w):
            raise ValueError(f"warp: occlusion must be [{n},1,{h},{w}], got {tuple(occlusion.shape)}")
        occ = occlusion.reshape(n, h, w).contiguous()
    out = _WarpFunction.apply(features.permute(0, 2, 3, 1).contiguous(), deformation.contiguous(), occ)
    return out.permute(0, 3, 1, 2)


class _Conv2dSameFunction(torch.autograd.Function):
    """NHWC x [B,H,W,Cin], OIHW weight, bias -> NHWC [B,H,W,Cout]."""

    @staticmethod
    def _conv(x, weight, bias, transposed=False):
        """eamm_op_conv_dev: parameters stay on the device, the filter is packed there.  transposed: `weight` is the FORWARD
        filter [Cin,Cout,k,k] of the convolution whose data gradient this call computes (x = grad_out)."""
        b, h, w, cin = x.shape
        cout = weight.shape[1] if transposed else weight.shape[0]
        kh, kw = weight.shape[2:]
        L = _lib.lib()
        out = torch.empty(b, h, w, cout, dtype=torch.float32, device=x.device)
        nwork = L.eamm_op_conv_dev_workspace_floats(b, h, w, cin, cout, kh, kw)
        if nwork == 0:
            raise ValueError(f"conv2d_same: unsupported convolution {tuple(x.shape)} x {tuple(weight.shape)}")
        work = torch.empty(nwork, dtype=torch.float32, device=x.device)
        wt = weight.detach().contiguous()
        bt = bias.detach().contiguous() if bias is not None else None
        with torch.cuda.device(x.device):
            _lib.check(L.eamm_op_conv_dev(x.device.index, _ptr(x), b, h, w, cin, _ptr(wt), _ptr(bt), cout, kh, kw, int(transposed),
                                          _ptr(out), _ptr(work), nwork, _stream(x.device)), None)
        return out

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _Conv2dSameFunction._conv(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, h, w, cin = x.shape
        cout, _, kh, kw = weight.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dX = "same" correlation of dY with the filter transposed over (out, in) and flipped over (y, x)
            gx = _Conv2dSameFunction._conv(grad_out, weight, None, transposed=True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            L = _lib.lib()
            gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nwork = L.eamm_op_conv_wgrad_workspace_floats(b, h, w, cin, cout, kh, kw)
            work = torch.empty(max(1, nwork), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(L.eamm_op_conv_wgrad(x.device.index, _ptr(x), _ptr(grad_out), b, h, w, cin, cout, kh, kw, _ptr(gw), _ptr(gb),
                                                _ptr(work), nwork, _stream(x.device)), None)
        return gx, gw, gb


def conv2d_same_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``conv2d_same`` on the kernels' own layout: x NHWC [B,H,W,Cin] -> NHWC [B,H,W,Cout] (no layout copies)."""
    _need_gpu(x, "conv2d_same_nhwc")
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((3, 3), (7, 7)) or x.shape[3] != cin or cin % _CONV_BK or cout % _CONV_BK:
        raise ValueError(f"conv2d_same_nhwc: weight {tuple(weight.shape)} / input {tuple(x.shape)} unsupported "
                         f"(3x3 or 7x7, channels multiples of {_CONV_BK})")
    if weight.device != x.device or (bias is not None and bias.device != x.device):
        raise RuntimeError("conv2d_same_nhwc: parameters and input are on different devices")
    return _Conv2dSameFunction.apply(x.contiguous(), weight, bias)


def warp_nhwc(features: torch.Tensor, deformation: torch.Tensor, occlusion: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``warp`` on NHWC features [n|1,h,w,C] (C a multiple of 8), flow [n,h,w,2], occlusion [n,h,w] or None -> NHWC [n,h,w,C]."""
    _need_gpu(features, "warp_nhwc")
    n, h, w, two = deformation.shape
    if two != 2 or features.shape[1:3] != (h, w) or features.shape[0] not in (1, n) or features.shape[3] % 8:
        raise ValueError(f"warp_nhwc: features {tuple(features.shape)} / deformation {tuple(deformation.shape)} mismatch")
    if occlusion is not None and tuple(occlusion.shape) != (n, h, w):
        raise ValueError(f"warp_nhwc: occlusion must be [{n},{h},{w}], got {tuple(occlusion.shape)}")
    return _WarpFunction.apply(features.contiguous(), deformation.contiguous(), None if occlusion is None else occlusion.contiguous())


def conv2d_same(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.conv2d(x, weight, bias, padding=k // 2)`` for the path's square 3x3 / 7x7 filters (reference modules/util.py:858-938),
    differentiable in x, weight and bias.  x NCHW [B,Cin,H,W], weight [Cout,Cin,k,k] on the GPU; Cin and Cout multiples of 32
    (the granule of the forward kernels, which also compute the data gradient).  Returns NCHW."""
    _need_gpu(x, "conv2d_same")
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((3, 3), (7, 7)) or x.shape[1] != cin or cin % _CONV_BK or cout % _CONV_BK:
        raise ValueError(f"conv2d_same: weight {tuple(weight.shape)} / input {tuple(x.shape)} unsupported "
                         f"(3x3 or 7x7, channels multiples of {_CONV_BK})")
    if weight.device != x.device or (bias is not None and bias.device != x.device):
        raise RuntimeError("conv2d_same: parameters and input are on different devices")
    out = _Conv2dSameFunction.apply(x.permute(0, 2, 3, 1).contiguous(), weight, bias)
    return out.permute(0, 3, 1, 2)
