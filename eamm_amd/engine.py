"""Python handle around the C ABI (include/eamm_hip.h): one ``Engine`` = one ``eamm_ctx``.

PyTorch is plumbing here -- it owns device memory (tensors), the current HIP stream and, for the
multi-GPU clip pipeline, ``torch.distributed``; every FLOP of the path runs in libeamm_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional

import torch

from . import _lib

_OUTPUT_KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed", "deformation")


def config_struct(cfg: dict, height: int, width: int, max_frames: int, max_sources: int) -> _lib.EammConfig:
    """Constructor kwargs of OcclusionAwareGenerator (reference generator.py:14-15) -> eamm_config."""
    dm = cfg.get("dense_motion_params")
    if dm is None:   # a generator without a motion network (generator.py:22-23): dm_num_blocks = 0 tells the library
        dm = {"block_expansion": 0, "max_features": 0, "num_blocks": 0, "scale_factor": 1}
    scale = dm.get("scale_factor", 1)
    inv = int(round(1.0 / scale))
    if abs(inv * scale - 1.0) > 1e-6:
        raise ValueError(f"scale_factor={scale} is not 1/integer")
    s = _lib.EammConfig()
    s.num_channels = cfg["num_channels"]
    s.num_kp = cfg["num_kp"]
    s.block_expansion = cfg["block_expansion"]
    s.max_features = cfg["max_features"]
    s.num_down_blocks = cfg["num_down_blocks"]
    s.num_bottleneck_blocks = cfg["num_bottleneck_blocks"]
    # (without a motion network the reference never reads estimate_occlusion_map: the map is that network's output)
    s.estimate_occlusion_map = int(bool(cfg.get("estimate_occlusion_map", False)) and cfg.get("dense_motion_params") is not None)
    s.dm_block_expansion = dm["block_expansion"]
    s.dm_max_features = dm["max_features"]
    s.dm_num_blocks = dm["num_blocks"]
    s.dm_inv_scale = inv
    s.kp_variance = float(dm.get("kp_variance", 0.01))
    s.height, s.width = int(height), int(width)
    s.max_frames, s.max_sources = int(max_frames), int(max_sources)
    return s


def _dev_ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def library_knobs() -> dict:
    """Every EAMM_* tuning knob the library has read so far in this process: {name: {"value": v, "set": 0|1}}."""
    import json
    L = _lib.lib()
    need = L.eamm_knobs_json(None, 0)
    buf = C.create_string_buffer(need + 1)
    L.eamm_knobs_json(buf, len(buf))
    return json.loads(buf.value.decode())


class Engine:
    def __init__(self, cfg: dict, height: int, width: int, max_frames: int = 16, max_sources: int = 1,
                 device: Optional[torch.device] = None, training: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("eamm_amd.Engine needs a ROCm GPU: the path has no CPU fallback")
        self.cfg = dict(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"eamm_amd.Engine needs a GPU device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.height, self.width = int(height), int(width)
        self.max_frames, self.max_sources = int(max_frames), int(max_sources)
        self.num_kp = cfg["num_kp"]
        self.num_channels = cfg["num_channels"]   # 1..6 (include/eamm_hip.h: one or two run as the equivalent three-channel network, four to six as two channel groups)
        self._L = _lib.lib()
        self._cs = config_struct(cfg, height, width, max_frames, max_sources)
        self.inv_scale = self._cs.dm_inv_scale
        self.h, self.w = self.height // self.inv_scale, self.width // self.inv_scale
        self.has_occlusion = bool(self._cs.estimate_occlusion_map)
        self.has_motion = self._cs.dm_num_blocks > 0
        ctx = C.c_void_p()
        _lib.check(self._L.eamm_create(C.byref(self._cs), self.device.index, C.byref(ctx)), None)
        self._ctx = ctx
        self._finalized = False
        # training-mode handle (eamm_set_training): raw convolution weights, BatchNorm as separate kernels with batch
        # statistics -- only train_forward() may be used on it
        self.training = bool(training)
        if self.training:
            _lib.check(self._L.eamm_set_training(self._ctx, 1), self._ctx)
        self.ns_cached = 0
        # bumped by every encode / import: lets a caller that caches "my source is the encoded one" (the module's
        # forward()) notice that somebody else (the clip pipeline) has replaced the engine's source cache since
        self.cache_generation = 0

    # -- lifecycle -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.eamm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]):
        """checkpoint['generator'] (reference demo.py:91) -> folded, MFMA-packed device weights."""
        for key, t in state_dict.items():
            if key.endswith("num_batches_tracked"):
                continue
            host = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, host.dim()))(*host.shape)
            _lib.check(self._L.eamm_load_tensor(self._ctx, key.encode(), C.c_void_p(host.data_ptr()), shape,
                                                host.dim()), self._ctx)
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_finalize_weights(self._ctx), self._ctx)
        self._finalized = True

    # -- forward ---------------------------------------------------------------------------------
    def _check_dev(self, t: torch.Tensor, name: str, shape_tail):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if t.device != self.device:
            raise RuntimeError(f"{name} is on {t.device}, the generator is on {self.device}")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32, got {t.dtype}")
        if tuple(t.shape[1:]) != tuple(shape_tail):
            raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected [*,{','.join(map(str, shape_tail))}]")
        return t.contiguous()

    def encode_source(self, source: torch.Tensor) -> int:
        if self.training:
            raise RuntimeError("a training-mode engine has no BatchNorm folded into its convolutions: use train_forward()")
        src = self._check_dev(source, "source_image", (self.num_channels, self.height, self.width))
        ns = src.shape[0]
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_encode_source(self._ctx, _dev_ptr(src), ns, self._stream()), self._ctx)
        self.ns_cached = ns
        self.cache_generation += 1
        return ns

    # -- training-mode forward (include/eamm_hip.h, "N4, second slice") ------------------------------------------------
    def train_sites(self):
        """State-dict prefixes of the BatchNorm sites of one forward, in execution order."""
        return [self._L.eamm_train_site_name(self._ctx, i).decode() for i in range(self._L.eamm_train_num_sites(self._ctx))]

    def train_forward(self, source: torch.Tensor, kp_driving: dict, kp_source: dict, norms: Dict[str, torch.nn.Module],
                      outputs: Iterable[str] = ("prediction",), sync: bool = False, reduce=None) -> Dict[str, torch.Tensor]:
        """One forward with batch statistics.  ``norms``: BatchNorm module per site name (weight, bias and -- updated in
        place -- running_mean / running_var are read through their device pointers).  ``reduce(t)``: all-reduce (sum) of the
        float tensor ``t`` over the replicas, called once per site between the statistics and the normalisation; None on
        one replica."""
        if not self.training:
            raise RuntimeError("train_forward needs an Engine(training=True)")
        K, H, W, h, w = self.num_kp, self.height, self.width, self.h, self.w
        src = self._check_dev(source, "source_image", (self.num_channels, H, W))
        n = src.shape[0]
        kd = ks = kdj = ksj = None
        if self.has_motion:
            kd = self._check_dev(kp_driving["value"], "kp_driving['value']", (K, 2))
            ks = self._check_dev(kp_source["value"], "kp_source['value']", (K, 2))
            if kd.shape[0] != n or ks.shape[0] != n:
                raise RuntimeError("key-point batch size does not match source_image batch size")
            if "jacobian" in kp_driving:
                kdj = self._check_dev(kp_driving["jacobian"], "kp_driving['jacobian']", (K, 2, 2))
                ksj = self._check_dev(kp_source["jacobian"], "kp_source['jacobian']", (K, 2, 2))
        want = set(outputs) | {"prediction"}
        if "occlusion_map" in want and not self.has_occlusion:
            want.discard("occlusion_map")
        ch = self.num_channels
        shapes = {"prediction": (n, ch, H, W), "mask": (n, K + 1, h, w), "sparse_deformed": (n, K + 1, ch, h, w),
                  "occlusion_map": (n, 1, h, w), "deformed": (n, ch, H, W), "deformation": (n, h, w, 2)}
        res = {k: torch.empty(shapes[k], dtype=torch.float32, device=self.device) for k in _OUTPUT_KEYS if k in want}
        o = _lib.EammOutputs()
        for k, t in res.items():
            setattr(o, k, t.data_ptr())
        names = self.train_sites()
        sites = (_lib.EammBnSite * len(names))()
        momentum = eps = None
        for i, name in enumerate(names):
            m = norms[name]
            for t in (m.weight, m.bias, m.running_mean, m.running_var):
                if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                    raise RuntimeError(f"BatchNorm tensors of {name} must be contiguous float32 on {self.device}")
            sites[i].weight, sites[i].bias = m.weight.data_ptr(), m.bias.data_ptr()
            sites[i].running_mean, sites[i].running_var = m.running_mean.data_ptr(), m.running_var.data_ptr()
            if momentum is None:
                momentum, eps = float(m.momentum), float(m.eps)
            elif (float(m.momentum), float(m.eps)) != (momentum, eps):
                raise RuntimeError("all BatchNorm sites must share momentum and eps (they do in the reference)")
        sums = torch.empty(6 * self._L.eamm_train_max_channels(self._ctx) + 2, dtype=torch.float32, device=self.device)
        nfl = C.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_train_begin(self._ctx, _dev_ptr(src), n, _dev_ptr(kd), _dev_ptr(kdj), _dev_ptr(ks), _dev_ptr(ksj),
                                                sites, len(names), momentum, eps, int(bool(sync)), _dev_ptr(sums), C.byref(o),
                                                self._stream()), self._ctx)
            while True:
                rc = self._L.eamm_train_next(self._ctx, C.byref(nfl))
                if rc < 0:
                    _lib.check(rc, self._ctx)
                if rc == 0:
                    break
                if reduce is not None:
                    reduce(sums[:nfl.value])
        self.ns_cached = 0
        self.cache_generation += 1
        return res

    def forward_frames(self, kp_driving: dict, kp_source: dict, outputs: Iterable[str] = ("prediction",),
                       uint8_frames: bool = False) -> Dict[str, torch.Tensor]:
        if self.training:
            raise RuntimeError("a training-mode engine has no BatchNorm folded into its convolutions: use train_forward()")
        K, H, W, h, w = self.num_kp, self.height, self.width, self.h, self.w
        kd = self._check_dev(kp_driving["value"], "kp_driving['value']", (K, 2))
        ks = self._check_dev(kp_source["value"], "kp_source['value']", (K, 2))
        n = kd.shape[0]
        kdj = ksj = None
        if "jacobian" in kp_driving:  # dense_motion.py:55
            kdj = self._check_dev(kp_driving["jacobian"], "kp_driving['jacobian']", (K, 2, 2))
            ksj = self._check_dev(kp_source["jacobian"], "kp_source['jacobian']", (K, 2, 2))
            if kdj.shape[0] != n or ksj.shape[0] != ks.shape[0]:
                raise RuntimeError("jacobian batch size does not match value batch size")
        if ks.shape[0] != self.ns_cached:
            raise RuntimeError(f"kp_source has {ks.shape[0]} sets but {self.ns_cached} source(s) are encoded")
        want = set(outputs) | {"prediction"}
        unknown = want - set(_OUTPUT_KEYS)
        if unknown:
            raise KeyError(f"unknown output(s) {sorted(unknown)}")
        if "occlusion_map" in want and not self.has_occlusion:
            want.discard("occlusion_map")
        if not self.has_motion and want != {"prediction"}:
            raise KeyError(f"this generator has no motion network: only 'prediction' exists, not {sorted(want - {'prediction'})}")
        ch = self.num_channels
        shapes = {"prediction": (n, ch, H, W), "mask": (n, K + 1, h, w), "sparse_deformed": (n, K + 1, ch, h, w),
                  "occlusion_map": (n, 1, h, w), "deformed": (n, ch, H, W), "deformation": (n, h, w, 2)}
        res = {k: torch.empty(shapes[k], dtype=torch.float32, device=self.device) for k in _OUTPUT_KEYS if k in want}
        o = _lib.EammOutputs()
        for k, t in res.items():
            setattr(o, k, t.data_ptr())
        if uint8_frames and self.num_channels != 3:
            raise RuntimeError("uint8 RGB frames need num_channels == 3")
        if uint8_frames:
            res["frames_u8"] = torch.empty((n, H, W, 3), dtype=torch.uint8, device=self.device)
            o.frames_u8 = res["frames_u8"].data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_forward_frames(self._ctx, n, _dev_ptr(kd), _dev_ptr(kdj), _dev_ptr(ks),
                                                   _dev_ptr(ksj), C.byref(o), self._stream()), self._ctx)
        return res

    def check_numeric(self):
        """Synchronise and raise if a singular key-point jacobian was met (torch.inverse semantics)."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_check_numeric(self._ctx, self._stream()), self._ctx)

    # -- source cache (multi-GPU broadcast payload) -------------------------------------------------
    def source_cache_numel(self, ns: int = 1) -> int:
        return self._L.eamm_source_cache_bytes(self._ctx, ns) // 4

    def export_source_cache(self, ns: Optional[int] = None) -> torch.Tensor:
        ns = self.ns_cached if ns is None else ns
        blob = torch.empty(self.source_cache_numel(ns), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_export_source_cache(self._ctx, _dev_ptr(blob), ns, self._stream()), self._ctx)
        return blob

    def import_source_cache(self, blob: torch.Tensor, ns: int = 1):
        if blob.device != self.device or blob.dtype != torch.float32 or blob.numel() != self.source_cache_numel(ns):
            raise RuntimeError("source cache blob has the wrong device, dtype or size")
        with torch.cuda.device(self.device):
            _lib.check(self._L.eamm_import_source_cache(self._ctx, _dev_ptr(blob.contiguous()), ns, self._stream()),
                       self._ctx)
        self.ns_cached = ns
        self.cache_generation += 1

    # -- stage timing (HIP events inside the library, on the stream the kernels run on) -----------------
    STAGES = ("front", "hg_enc", "hg_dec", "head", "warp", "bneck_transform", "bneck_conv", "up", "final", "bneck_gemm_kernel",
              "bneck_union", "bneck_windows", "exec_gflop", "bneck_exec_gflop",   # these four: chip-level accounting (eamm_hip.h)
              "gf_front", "gf_hg_enc", "gf_hg_dec", "gf_head", "gf_warp", "gf_bneck", "gf_up", "gf_final")   # executed GFLOP per stage

    def profile(self, on: bool = True):
        _lib.check(self._L.eamm_profile_enable(self._ctx, int(on)), self._ctx)

    def profile_read(self, reset: bool = True) -> dict:
        ms = (C.c_double * len(self.STAGES))()
        calls, frames = C.c_int64(), C.c_int64()
        _lib.check(self._L.eamm_profile_read(self._ctx, ms, len(self.STAGES), C.byref(calls), C.byref(frames),
                                             int(reset)), self._ctx)
        return {"calls": calls.value, "frames": frames.value, "ms": dict(zip(self.STAGES, list(ms)))}

    # -- accounting --------------------------------------------------------------------------------
    @property
    def flops_per_frame(self) -> float:
        return self._L.eamm_flops_per_frame(self._ctx)

    def bottleneck_form(self, frames: int) -> int:
        """0 = direct, 2 = Winograd F(2x2,3x3), 4 = Winograd F(4x4,3x3) for a call of ``frames`` frames."""
        return self._L.eamm_bottleneck_form(self._ctx, int(frames))

    def bottleneck_chains(self, frames: int) -> int:
        """Concurrent launch sequences (streams) the bottleneck of a call of ``frames`` frames is split into."""
        return self._L.eamm_bottleneck_chains(self._ctx, int(frames))

    def pass_chains(self, frames: int) -> int:
        """Of those, the chains that run the whole per-frame pass as independent launch sequences (1 = none)."""
        return self._L.eamm_pass_chains(self._ctx, int(frames))

    def last_stream_set(self) -> int:
        """Side streams of the last forward_frames call: 0 the device's shared pool, 1 the handle's private streams (another
        thread held the pool), 2 private because the caller's stream was being captured (include/eamm_hip.h)."""
        return self._L.eamm_last_stream_set(self._ctx)

    def describe_plan(self, frames: int) -> dict:
        """The launch plan the library picks for a call of ``frames`` frames (chains, forms, kernels), for benchmark records."""
        import json
        buf = C.create_string_buffer(2048)
        n = self._L.eamm_describe_plan(self._ctx, int(frames), buf, len(buf))
        if n < 0:
            raise ValueError(f"no plan for a call of {frames} frames (max_frames={self.max_frames})")
        return json.loads(buf.value.decode())

    @property
    def encode_flops(self) -> float:
        return self._L.eamm_encode_flops(self._ctx)
