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
_NO_OFFSET = C.c_size_t(-1).value
_SAVE_TRANSFORM = __import__("os").environ.get("EAMM_SAVE_TRANSFORM", "1") != "0"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# The weight gradient of a convolution and its data gradient are independent given (x, dY).  EAMM_WGRAD_STREAM=1 runs the weight
# gradient on a side stream beside the data gradient (the current stream waits for it before the backward returns, so autograd
# sees both results in stream order; buffers are allocated on the CURRENT stream and released only after that wait, so the
# caching allocator's stream-order reuse holds).  MEASURED AND OFF (profiles/r04_experiments.txt section 14): 8 pairs at 256x256
# backward 12.1-12.2 -> 12.4-12.5 ms, 16 pairs 21.2 -> 21.7 ms -- both launch sequences already fill the chip, the overlap only
# adds contention and two event hand-overs per layer.
_SIDE_STREAMS = {}


def _wgrad_stream(dev: torch.device):
    import os
    if os.environ.get("EAMM_WGRAD_STREAM", "0") != "1" or torch.cuda.is_current_stream_capturing():
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


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
    def _conv(x, weight, bias, transposed=False, keep_work=False):
        """eamm_op_conv_dev: parameters stay on the device, the filter is packed there.  transposed: `weight` is the FORWARD
        filter [Cin,Cout,k,k] of the convolution whose data gradient this call computes (x = grad_out).  keep_work: also return
        the call's workspace (it holds the transformed input the weight gradient can start from)."""
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
        return (out, work) if keep_work else out

    @staticmethod
    def forward(ctx, x, weight, bias, bias_grad_is_zero=False):
        ctx.has_bias = bias is not None
        ctx.bias_grad_is_zero = bool(bias_grad_is_zero)
        # Where the forward ran the F(4x4,3x3) form and the weight gradient takes its F(3x3,4x4) form, both start from the same
        # transformed input: the forward's workspace is kept for the backward (2.25 x the activation; EAMM_SAVE_TRANSFORM=0: not kept)
        b, h, w, cin = x.shape
        cout, _, kh, kw = weight.shape
        off = _lib.lib().eamm_op_conv_saved_transform_offset(b, h, w, cin, cout, kh, kw) if _SAVE_TRANSFORM and ctx.needs_input_grad[1] else _NO_OFFSET
        ctx.v_offset = off
        if off != _NO_OFFSET:
            out, work = _Conv2dSameFunction._conv(x, weight, bias, keep_work=True)
            ctx.save_for_backward(x, weight, work)
            return out
        ctx.save_for_backward(x, weight)
        return _Conv2dSameFunction._conv(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors[:2]
        saved_work = ctx.saved_tensors[2] if ctx.v_offset != _NO_OFFSET else None
        grad_out = grad_out.contiguous()
        b, h, w, cin = x.shape
        cout, _, kh, kw = weight.shape
        gx = gw = gb = side = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            L = _lib.lib()
            gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            # a bias that feeds a batch-statistics BatchNorm directly has a gradient of exactly zero (the mean is subtracted
            # again): the caller says so and the sum over the pixels is not computed (autograd's own value there is rounding noise)
            zero_gb = ctx.has_bias and ctx.bias_grad_is_zero
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if (ctx.has_bias and not zero_gb) else None
            nwork = L.eamm_op_conv_wgrad_workspace_floats(b, h, w, cin, cout, kh, kw)
            work = torch.empty(max(1, nwork), dtype=torch.float32, device=x.device)
            if zero_gb:
                gb0 = torch.zeros(cout, dtype=torch.float32, device=x.device)
            side = _wgrad_stream(x.device) if ctx.needs_input_grad[0] else None
            cur = torch.cuda.current_stream(x.device)
            if side is not None:
                side.wait_stream(cur)            # x, dY and the buffers above are ready in the current stream's order
            with torch.cuda.device(x.device):
                if saved_work is not None:
                    v = C.c_void_p(saved_work.data_ptr() + 4 * ctx.v_offset)
                    _lib.check(L.eamm_op_conv_wgrad_saved(x.device.index, v, _ptr(grad_out), b, h, w, cin, cout, _ptr(gw), _ptr(gb),
                                                          _ptr(work), nwork, C.c_void_p((side or cur).cuda_stream)), None)
                else:
                    _lib.check(L.eamm_op_conv_wgrad(x.device.index, _ptr(x), _ptr(grad_out), b, h, w, cin, cout, kh, kw, _ptr(gw), _ptr(gb),
                                                    _ptr(work), nwork, C.c_void_p((side or cur).cuda_stream)), None)
            if zero_gb:
                gb = gb0
        if ctx.needs_input_grad[0]:
            # dX = "same" correlation of dY with the filter transposed over (out, in) and flipped over (y, x)
            gx = _Conv2dSameFunction._conv(grad_out, weight, None, transposed=True)
        if side is not None:
            cur.wait_stream(side)
        return gx, gw, gb, None


class _FirstConv7Function(torch.autograd.Function):
    """The generator's first layer (reference modules/generator.py:26: 7x7, 3 -> N): x4 NHWC [B,H,W,4] (three channels + a zero),
    weight [N,3,7,7], bias [N] -> NHWC [B,H,W,N].  eamm_op_conv7_thin / _wgrad: the three channels are part of the GEMM's K
    index instead of being padded to 32."""

    @staticmethod
    def forward(ctx, x4, weight, bias, bias_grad_is_zero=False):
        ctx.bias_grad_is_zero = bool(bias_grad_is_zero)
        b, h, w, _ = x4.shape
        n = weight.shape[0]
        L = _lib.lib()
        out = torch.empty(b, h, w, n, dtype=torch.float32, device=x4.device)
        nwork = L.eamm_op_conv7_thin_workspace_floats(b, h, w, n)
        work = torch.empty(nwork, dtype=torch.float32, device=x4.device)
        wt, bt = weight.detach().contiguous(), bias.detach().contiguous()
        with torch.cuda.device(x4.device):
            _lib.check(L.eamm_op_conv7_thin(x4.device.index, _ptr(x4), _ptr(wt), _ptr(bt), b, h, w, n, 0, _ptr(out), _ptr(work), nwork,
                                            _stream(x4.device)), None)
        ctx.save_for_backward(x4, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x4, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, h, w, _ = x4.shape
        n = weight.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:   # the source image rarely needs a gradient: the general kernels on the padded filter
            wp = torch.nn.functional.pad(weight.detach(), (0, 0, 0, 0, 0, _CONV_BK - 3))              # [N,32,7,7]
            gx = _Conv2dSameFunction._conv(grad_out, wp, None, transposed=True)[..., :4].contiguous()
            gx[..., 3] = 0
        if ctx.needs_input_grad[1]:
            L = _lib.lib()
            gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            nwork = L.eamm_op_conv7_thin_workspace_floats(b, h, w, n)
            work = torch.empty(nwork, dtype=torch.float32, device=x4.device)
            with torch.cuda.device(x4.device):
                _lib.check(L.eamm_op_conv7_thin_wgrad(x4.device.index, _ptr(x4), _ptr(grad_out), b, h, w, n, 1, _ptr(gw), _ptr(work), nwork,
                                                      _stream(x4.device)), None)
        if ctx.needs_input_grad[2]:
            gb = torch.zeros_like(weight[:, 0, 0, 0]) if ctx.bias_grad_is_zero else grad_out.sum(dim=(0, 1, 2))
        return gx, gw, gb, None


class _FinalConvSigmoidFunction(torch.autograd.Function):
    """The generator's last layer (reference modules/generator.py:92-93): sigmoid(conv7x7(x NHWC [B,H,W,N], weight [3,N,7,7]) + bias)
    -> NCHW [B,3,H,W].  Forward: the evaluation path's fused column-patch kernel on a device-packed filter
    (eamm_op_final_conv_sigmoid); backward: eamm_op_conv7_thin (data gradient) and eamm_op_conv7_thin_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        b, h, w, n = x.shape
        L = _lib.lib()
        y = torch.empty(b, 3, h, w, dtype=torch.float32, device=x.device)
        nwork = 7 * n * 32
        work = torch.empty(nwork, dtype=torch.float32, device=x.device)
        wt, bt = weight.detach().contiguous(), bias.detach().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(L.eamm_op_final_conv_sigmoid(x.device.index, _ptr(x), _ptr(wt), _ptr(bt), b, h, w, n, _ptr(y), _ptr(work), nwork,
                                                    _stream(x.device)), None)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight, y = ctx.saved_tensors
        b, h, w, n = x.shape
        g = grad_y * y * (1.0 - y)                                                          # through the sigmoid, NCHW [B,3,H,W]
        g4 = torch.nn.functional.pad(g.permute(0, 2, 3, 1), (0, 1)).contiguous()             # NHWC, four floats per pixel
        L = _lib.lib()
        nwork = L.eamm_op_conv7_thin_workspace_floats(b, h, w, n)
        work = torch.empty(nwork, dtype=torch.float32, device=x.device)
        gx = gw = gb = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                wt = weight.detach().contiguous()
                _lib.check(L.eamm_op_conv7_thin(x.device.index, _ptr(g4), _ptr(wt), None, b, h, w, n, 1, _ptr(gx), _ptr(work), nwork,
                                                _stream(x.device)), None)
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
                work2 = torch.empty(nwork, dtype=torch.float32, device=x.device)
                _lib.check(L.eamm_op_conv7_thin_wgrad(x.device.index, _ptr(g4), _ptr(x), b, h, w, n, 0, _ptr(gw), _ptr(work2), nwork,
                                                      _stream(x.device)), None)
        if ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3))
        return gx, gw, gb


def first_conv7(x4: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, bias_grad_is_zero: bool = False) -> torch.Tensor:
    """7x7 "same" convolution of a three-channel image (NHWC [B,H,W,4], fourth channel zero) to N = 32 | 64 channels."""
    _need_gpu(x4, "first_conv7")
    if x4.shape[3] != 4 or tuple(weight.shape[1:]) != (3, 7, 7) or weight.shape[0] not in (32, 64) or bias is None:
        raise ValueError(f"first_conv7: input {tuple(x4.shape)} / weight {tuple(weight.shape)} unsupported")
    return _FirstConv7Function.apply(x4.contiguous(), weight, bias, bias_grad_is_zero)


def final_conv7_sigmoid(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """sigmoid(7x7 "same" convolution of NHWC [B,H,W,N], N = 32 | 64, to three channels + bias) -> NCHW [B,3,H,W]."""
    _need_gpu(x, "final_conv7_sigmoid")
    if tuple(weight.shape) != (3, x.shape[3], 7, 7) or x.shape[3] not in (32, 64) or bias is None:
        raise ValueError(f"final_conv7_sigmoid: input {tuple(x.shape)} / weight {tuple(weight.shape)} unsupported")
    return _FinalConvSigmoidFunction.apply(x.contiguous(), weight, bias)


def conv2d_same_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     bias_grad_is_zero: bool = False) -> torch.Tensor:
    """``conv2d_same`` on the kernels' own layout: x NHWC [B,H,W,Cin] -> NHWC [B,H,W,Cout] (no layout copies).
    ``bias_grad_is_zero``: the output feeds a batch-statistics BatchNorm directly, so the bias gradient is exactly zero and is
    returned as zeros without being computed."""
    _need_gpu(x, "conv2d_same_nhwc")
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((3, 3), (7, 7)) or x.shape[3] != cin or cin % _CONV_BK or cout % _CONV_BK:
        raise ValueError(f"conv2d_same_nhwc: weight {tuple(weight.shape)} / input {tuple(x.shape)} unsupported "
                         f"(3x3 or 7x7, channels multiples of {_CONV_BK})")
    if weight.device != x.device or (bias is not None and bias.device != x.device):
        raise RuntimeError("conv2d_same_nhwc: parameters and input are on different devices")
    return _Conv2dSameFunction.apply(x.contiguous(), weight, bias, bias_grad_is_zero)


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
    out = _Conv2dSameFunction.apply(x.permute(0, 2, 3, 1).contiguous(), weight, bias, False)
    return out.permute(0, 3, 1, 2)
