"""Drop-in for the ``decon`` tail of the audio-to-feature network.

The reference's ``AT_net2.decon`` (modules/util.py:559-576) is an ``nn.Sequential`` of five ConvTranspose2d layers
(+ BatchNorm2d + ReLU) that turns one LSTM output vector into the [35,64,64] feature maps ``KPDetector_a`` reads; the
reference calls it once per frame with batch 1 (util.py:604-607).  ``DeconvTail`` is an ``nn.Sequential`` with the same
children indices -- so ``audio_feature.decon = DeconvTail()`` before ``audio_feature.load_state_dict(...)``
(demo.py:93) picks up the checkpoint's ``decon.N.*`` entries unchanged -- whose forward runs in libeamm_hip.so for
any batch, so a whole clip's frames go through in one call.  The children only hold parameters; there is no PyTorch
fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib
from .keypoint_detector import _refuse_silent_detach
from .weights import DECONV_CHANNELS


class SplitFeatureMap:
    """A [B, 32 m + 3, S, S] feature map in the layout ``KPDetector_a``'s heads read: ``wide`` NHWC [B,S,S,32 m] and ``thin``
    [B,S,S,4] (channels 32 m .. 32 m + 2 and a zero).  Produced by ``DeconvTail.forward_split``, accepted by ``KPDetector_a``."""
    __slots__ = ("wide", "thin")

    def __init__(self, wide: torch.Tensor, thin: torch.Tensor):
        if (wide.dim() != 4 or thin.dim() != 4 or thin.shape[-1] != 4 or wide.shape[:3] != thin.shape[:3] or wide.dtype != torch.float32
                or thin.dtype != torch.float32 or wide.device != thin.device):
            raise RuntimeError(f"SplitFeatureMap: wide {tuple(wide.shape)} / thin {tuple(thin.shape)} do not belong together")
        self.wide, self.thin = wide, thin

    @property
    def shape(self):
        b, h, w, c = self.wide.shape
        return (b, c + 3, h, w)

    @property
    def device(self):
        return self.wide.device

    def to_nchw(self) -> torch.Tensor:
        """The reference's tensor (util.py:604-607 `deco_out[:, t]`): [B, 32 m + 3, S, S]."""
        return torch.cat([self.wide, self.thin[..., :3]], dim=-1).permute(0, 3, 1, 2).contiguous()


class DeconvTail(nn.Sequential):
    """MI355X-native stand-in for reference modules/util.py:559-576 (``AT_net2.decon``)."""

    def __init__(self, channels: Sequence[int] = DECONV_CHANNELS, max_batch: int = 64):
        layers = []
        n = len(channels) - 1
        if not 2 <= n <= 8:
            raise ValueError("between 2 and 8 layers")
        for i in range(n):
            layers.append(nn.ConvTranspose2d(channels[i], channels[i + 1], kernel_size=6 if i == 0 else 4, stride=2,
                                             padding=1, bias=True))
            if i + 1 < n:
                layers += [nn.BatchNorm2d(channels[i + 1]), nn.ReLU(True)]
        super().__init__(*layers)
        self.channels = tuple(int(c) for c in channels)
        self.max_batch = int(max_batch)
        self._ctx: Optional[C.c_void_p] = None
        self._key = None
        self._cap = 0
        for p in self.parameters():
            p.requires_grad_(False)

    def _close(self):
        if self._ctx is not None:
            _lib.lib().eamm_deconv_destroy(self._ctx)
            self._ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self._close()
        except Exception:
            pass

    def _ensure(self, batch: int):
        dev = self[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("eamm_amd.DeconvTail runs only on a ROCm GPU (no CPU fallback): call .cuda()")
        if self.training:
            raise RuntimeError("inference-only: call .eval() (BatchNorm uses running statistics)")
        ver = tuple(t._version for t in self.state_dict(keep_vars=True).values())
        key = (dev, ver)
        if self._ctx is not None and self._key == key and batch <= self._cap:
            return
        self._close()
        L = _lib.lib()
        cs = _lib.EammDeconvConfig()
        cs.num_layers = len(self.channels) - 1
        for i, c in enumerate(self.channels):
            cs.channels[i] = c
        self._cap = max(batch, self.max_batch)
        cs.max_batch = self._cap
        ctx = C.c_void_p()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(L.eamm_deconv_create(C.byref(cs), idx, C.byref(ctx)), None, deconv=True)
        self._ctx = ctx
        for name, t in self.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            host = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, host.dim()))(*host.shape)
            _lib.check(L.eamm_deconv_load_tensor(ctx, name.encode(), C.c_void_p(host.data_ptr()), shape, host.dim()),
                       ctx, deconv=True)
        with torch.cuda.device(dev):
            _lib.check(L.eamm_deconv_finalize_weights(ctx), ctx, deconv=True)
        self._key = key

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B,C0,1,1] (the reference's call, util.py:603-607) or [B,C0]; returns [B,C_last,S,S]."""
        _refuse_silent_detach(self, x)
        with torch.no_grad():
            return self._forward(x)

    def forward_split(self, x: torch.Tensor) -> "SplitFeatureMap":
        """The same layers, the result in the PRIVATE layout ``KPDetector_a``'s heads read (round 6): ``wide`` NHWC
        [B,S,S,C_last - 3] + ``thin`` [B,S,S,4] (the last three channels and a zero) instead of the reference's NCHW
        [B,C_last,S,S] (util.py:604-607) -- both ends of that tensor live in this library, so the clip harness
        (``driving_keypoints``) skips the NCHW round trip.  Only when C_last = 32 m + 3 (the shipped 35); the values are
        those of ``forward`` bit for bit (``SplitFeatureMap.to_nchw()`` gives the reference's tensor back)."""
        _refuse_silent_detach(self, x)
        with torch.no_grad():
            return self._forward(x, split=True)

    def split_channels(self) -> int:
        """``wide`` channel count of ``forward_split`` (0: this tail's output has no 32 m + 3 split)."""
        wide = self.channels[-1] - 3
        return wide if wide >= 32 and wide % 32 == 0 else 0

    def _forward(self, x: torch.Tensor, split: bool = False):
        if x.dim() == 4 and x.shape[2:] == (1, 1):
            x = x.flatten(1)
        if x.dim() != 2 or x.shape[1] != self.channels[0] or x.dtype != torch.float32:
            raise RuntimeError(f"expected float32 [B,{self.channels[0]}(,1,1)], got {tuple(x.shape)} {x.dtype}")
        b = x.shape[0]
        self._ensure(b)
        x = x.contiguous()
        side = 4 << (len(self.channels) - 2)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if split:
            wide_c = self.split_channels()
            if not wide_c:
                raise RuntimeError(f"forward_split needs 32 m + 3 output channels, this tail has {self.channels[-1]}")
            wide = torch.empty(b, side, side, wide_c, device=x.device)
            thin = torch.empty(b, side, side, 4, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().eamm_deconv_forward_split(self._ctx, C.c_void_p(x.data_ptr()), b, C.c_void_p(wide.data_ptr()),
                                                                C.c_void_p(thin.data_ptr()), stream), self._ctx, deconv=True)
            return SplitFeatureMap(wide, thin)
        out = torch.empty(b, self.channels[-1], side, side, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().eamm_deconv_forward(self._ctx, C.c_void_p(x.data_ptr()), b, C.c_void_p(out.data_ptr()),
                                                      stream), self._ctx, deconv=True)
        return out
