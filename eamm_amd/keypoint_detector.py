"""Drop-in ``KPDetector`` / ``KPDetector_a`` whose forward passes run in libeamm_hip.so.

Mirrors reference modules/keypoint_detector.py: same constructor keywords (``demo.py:59-72`` splats
``kp_detector_params`` with ``common_params`` / ``audio_params``), same ``state_dict`` keys, same output dict
(``value`` [B,K,2], ``heatmap`` [B,K,h-6,w-6], ``jacobian`` [B,K,2,2]).  ``KPDetector.forward(image)`` runs
anti-alias down-sampling + hourglass + heads (once per clip on the source, demo.py:206); ``KPDetector_a.
forward(feature_map)`` runs only the heads on the audio-driven feature map (once per frame, demo.py:219) -- its
``predictor`` hourglass is part of the checkpoint but unused by the reference's forward, and here too.
The sub-modules only hold parameters; there is no PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from .generator import _AntiAlias, _ConvNorm, _Stack
from .weights import hourglass_channels



def _refuse_silent_detach(mod: nn.Module, x: torch.Tensor):
    """These modules are inference-only (no backward kernels: SURVEY.md 8f N1 / N3 are forward rows; their parameters are
    frozen at construction).  If the caller nevertheless asks for a gradient -- gradients enabled and the input or a
    re-enabled parameter requires one -- the reference would return a differentiable tensor; returning a detached one
    would silently train nothing, so refuse instead."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod.parameters())):
        raise RuntimeError(f"eamm_amd.{type(mod).__name__} is inference-only (forward kernels, no backward): call it under "
                           "torch.no_grad() / with frozen parameters and an input that does not require grad; it never "
                           "returns a silently detached tensor")


class _KPBase(nn.Module):
    _with_predictor = True

    def __init__(self, block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                 estimate_jacobian=False, scale_factor=1, single_jacobian_map=False, pad=0, in_features=None,
                 max_batch=16):
        super().__init__()
        in_features = num_channels if in_features is None else in_features
        enc, dec, out_filters = hourglass_channels(block_expansion, in_features, num_blocks, max_features)
        pred = nn.Module()
        pred.encoder = _Stack("down_blocks", [_ConvNorm(ci, co, 3) for ci, co in enc])
        pred.decoder = _Stack("up_blocks", [_ConvNorm(ci, co, 3) for ci, co in dec])
        self.predictor = pred
        self.kp = nn.Conv2d(out_filters, num_kp, kernel_size=(7, 7), padding=pad)
        if estimate_jacobian:
            self.num_jacobian_maps = 1 if single_jacobian_map else num_kp
            self.jacobian = nn.Conv2d(out_filters, 4 * self.num_jacobian_maps, kernel_size=(7, 7), padding=pad)
            self.jacobian.weight.data.zero_()      # reference init: identity jacobians (keypoint_detector.py:27-28)
            self.jacobian.bias.data.copy_(torch.tensor([1, 0, 0, 1] * self.num_jacobian_maps, dtype=torch.float))
        else:
            self.jacobian = None
        self.temperature, self.scale_factor = temperature, scale_factor
        if scale_factor != 1:
            self.down = _AntiAlias(num_channels)
        self._cfg = dict(num_kp=num_kp, num_channels=num_channels, in_features=in_features,
                         block_expansion=block_expansion, max_features=max_features, num_blocks=num_blocks,
                         temperature=float(temperature), estimate_jacobian=int(bool(estimate_jacobian)),
                         single_jacobian_map=int(bool(single_jacobian_map)),
                         inv_scale=int(round(1.0 / scale_factor)), pad=int(pad))
        self.out_filters = out_filters
        self.max_batch = int(max_batch)
        self._ctx: Optional[C.c_void_p] = None
        self._key = None
        for p in self.parameters():
            p.requires_grad_(False)

    # -- handle management ----------------------------------------------------------------------------------
    def _close(self):
        if self._ctx is not None:
            _lib.lib().eamm_kp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self._close()
        except Exception:
            pass

    def _ensure(self, height: int, width: int, batch: int):
        dev = self.kp.weight.device
        if dev.type != "cuda":
            raise RuntimeError(f"eamm_amd.{type(self).__name__} runs only on a ROCm GPU (no CPU fallback): call .cuda()")
        if self.training:
            raise RuntimeError("inference-only: call .eval() (BatchNorm uses running statistics)")
        ver = tuple(t._version for t in self.state_dict(keep_vars=True).values())
        key = (dev, height, width, ver)
        if self._ctx is not None and self._key == key and batch <= self._cap:
            return
        self._close()
        L = _lib.lib()
        cs = _lib.EammKpConfig()
        for k, v in self._cfg.items():
            setattr(cs, k, v)
        cs.height, cs.width = int(height), int(width)
        self._cap = max(batch, self.max_batch)
        cs.max_batch = self._cap
        cs.with_predictor = int(self._with_predictor)
        ctx = C.c_void_p()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(L.eamm_kp_create(C.byref(cs), idx, C.byref(ctx)), None, kp=True)
        self._ctx = ctx
        needed = ("kp.", "jacobian.") + (("predictor.", "down.") if self._with_predictor else ())
        for name, t in self.state_dict().items():
            if name.endswith("num_batches_tracked") or not name.startswith(needed):
                continue
            host = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, host.dim()))(*host.shape)
            _lib.check(L.eamm_kp_load_tensor(ctx, name.encode(), C.c_void_p(host.data_ptr()), shape, host.dim()), ctx, kp=True)
        with torch.cuda.device(dev):
            _lib.check(L.eamm_kp_finalize_weights(ctx), ctx, kp=True)
        self._key = key

    _want_heatmap = True      # detect(..., heatmap=False) clears it for one call: the clip harness never reads the heat-maps (demo.py:219-281)

    def detect(self, x, heatmap: bool = True) -> Dict[str, torch.Tensor]:
        """``forward`` with the heat-maps optional: ``heatmap=False`` returns {'value', 'jacobian'} only and skips the heat-map pass
        (K x 58 x 58 floats per frame written for nobody: ``demo.py`` never reads them; ``driving_keypoints`` calls this)."""
        self._want_heatmap = bool(heatmap)
        try:
            return self(x)
        finally:
            self._want_heatmap = True

    def _run(self, x: torch.Tensor, fn_name: str, hm_h: int, hm_w: int) -> Dict[str, torch.Tensor]:
        dev, b, k = x.device, x.shape[0], self._cfg["num_kp"]
        out = {"value": torch.empty(b, k, 2, device=dev)}
        o = _lib.EammKpOutputs(value=out["value"].data_ptr())
        if self._want_heatmap:
            out["heatmap"] = torch.empty(b, k, hm_h, hm_w, device=dev)
            o.heatmap = out["heatmap"].data_ptr()
        if self.jacobian is not None:
            out["jacobian"] = torch.empty(b, k, 2, 2, device=dev)
            o.jacobian = out["jacobian"].data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(getattr(_lib.lib(), fn_name)(self._ctx, C.c_void_p(x.data_ptr()), b, C.byref(o), stream),
                       self._ctx, kp=True)
        return out


class KPDetector(_KPBase):
    """reference keypoint_detector.py:7-105"""

    _with_predictor = True

    def __init__(self, block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                 estimate_jacobian=False, scale_factor=1, single_jacobian_map=False, pad=0, max_batch=16):
        super().__init__(block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                         estimate_jacobian, scale_factor, single_jacobian_map, pad, None, max_batch)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        _refuse_silent_detach(self, x)
        with torch.no_grad():
            return self._forward(x)

    def _forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        nc = self._cfg["num_channels"]
        if x.dim() != 4 or x.shape[1] != nc or x.dtype != torch.float32:
            raise RuntimeError(f"expected a float32 [B,{nc},H,W] image batch, got {tuple(x.shape)} {x.dtype}")
        b, _, hh, ww = x.shape
        self._ensure(hh, ww, b)
        inv, pad = self._cfg["inv_scale"], self._cfg["pad"]
        return self._run(x.contiguous(), "eamm_kp_detect", hh // inv - 6 + 2 * pad, ww // inv - 6 + 2 * pad)


class KPDetector_a(_KPBase):
    """reference keypoint_detector.py:110-205 (forward takes the feature map, not an image)"""

    _with_predictor = False

    def __init__(self, block_expansion, num_kp, num_channels, num_channels_a, max_features, num_blocks, temperature,
                 estimate_jacobian=False, scale_factor=1, single_jacobian_map=False, pad=0, max_batch=16):
        super().__init__(block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                         estimate_jacobian, scale_factor, single_jacobian_map, pad, num_channels_a, max_batch)

    def forward(self, feature_map) -> Dict[str, torch.Tensor]:
        """``feature_map``: the reference's float32 [B, C, h, w] tensor, or (round 6) the ``SplitFeatureMap`` ``DeconvTail.forward_split``
        hands over (same values, the layout the heads read)."""
        _refuse_silent_detach(self, feature_map if torch.is_tensor(feature_map) else feature_map.wide)
        with torch.no_grad():
            return self._forward(feature_map)

    def accepts_split(self, height: int, width: int, batch: int = 1) -> int:
        """``wide`` channel count of the split feature map these heads read directly (``DeconvTail.forward_split``), 0 if they do
        not run in the wide + thin form (round 6: the private NHWC hand-over).  Builds the handle for (height, width) maps."""
        inv = self._cfg["inv_scale"]
        self._ensure(height * inv, width * inv, batch)
        return int(_lib.lib().eamm_kp_split_channels(self._ctx))

    def _forward(self, feature_map) -> Dict[str, torch.Tensor]:
        if not torch.is_tensor(feature_map) and hasattr(feature_map, "wide") and hasattr(feature_map, "thin"):
            return self._forward_split(feature_map)
        if feature_map.dim() != 4 or feature_map.shape[1] != self.out_filters or feature_map.dtype != torch.float32:
            raise RuntimeError(f"expected a float32 [B,{self.out_filters},h,w] feature map, got "
                               f"{tuple(feature_map.shape)} {feature_map.dtype}")
        b, _, h, w = feature_map.shape
        inv, pad = self._cfg["inv_scale"], self._cfg["pad"]
        self._ensure(h * inv, w * inv, b)
        return self._run(feature_map.contiguous(), "eamm_kp_detect_features", h - 6 + 2 * pad, w - 6 + 2 * pad)

    def _forward_split(self, fm) -> Dict[str, torch.Tensor]:
        b, c, h, w = fm.shape
        if c != self.out_filters:
            raise RuntimeError(f"expected a split [B,{self.out_filters},h,w] feature map, got {tuple(fm.shape)}")
        inv, pad = self._cfg["inv_scale"], self._cfg["pad"]
        self._ensure(h * inv, w * inv, b)
        if int(_lib.lib().eamm_kp_split_channels(self._ctx)) != c - 3:
            return self._forward(fm.to_nchw())          # heads not in the wide + thin form: the reference's tensor
        wide, thin = fm.wide.contiguous(), fm.thin.contiguous()
        dev, k = wide.device, self._cfg["num_kp"]
        out = {"value": torch.empty(b, k, 2, device=dev)}
        o = _lib.EammKpOutputs(value=out["value"].data_ptr())
        if self._want_heatmap:
            out["heatmap"] = torch.empty(b, k, h - 6 + 2 * pad, w - 6 + 2 * pad, device=dev)
            o.heatmap = out["heatmap"].data_ptr()
        if self.jacobian is not None:
            out["jacobian"] = torch.empty(b, k, 2, 2, device=dev)
            o.jacobian = out["jacobian"].data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().eamm_kp_detect_features_split(self._ctx, C.c_void_p(wide.data_ptr()), C.c_void_p(thin.data_ptr()), b,
                                                               C.byref(o), stream), self._ctx, kp=True)
        return out
