"""Drop-in ``OcclusionAwareGenerator`` whose forward pass runs in libeamm_hip.so.

Mirrors the reference module's interface for this path (reference modules/generator.py:8-97):
same constructor keywords (``demo.py:54-55`` splats the YAML sections into it), same
``forward(source_image, kp_driving, kp_source) -> dict`` contract and the same ``state_dict`` key
set, so ``generator.load_state_dict(checkpoint['generator'])`` / ``.cuda()`` / ``.eval()``
(demo.py:56-57, 91, 105) work unchanged.  The sub-modules below only HOLD parameters under the
reference's names; they are never called.  All computation happens in the HIP library; there is no
PyTorch/CPU fallback and ``forward`` raises when the module is not on a GPU.  ``.train()`` is supported (batch statistics in
every BatchNorm, running statistics updated, replicas' statistics all-reduced -- SURVEY.md 8f row N4).

Autograd semantics are the reference module's: parameters keep PyTorch's default ``requires_grad=True``, so
``torch.optim.Adam(generator.parameters())`` + ``loss.backward()`` (train.py:133-136) works straight after construction.
Whenever gradients are enabled and anything that reaches the output requires one -- a parameter, the key points, the source --
``forward`` is the composition of differentiable HIP operators of ``train_graph`` (batch statistics in ``.train()``, running
statistics in ``.eval()``) and its outputs carry the graph, exactly where the reference's would.  Under ``torch.no_grad()``
(demo.py:195) it is the graph-free engine: the folded fast path in ``.eval()``, the resumable batch-statistics pass in
``.train()``.  ``forward`` never returns a silently detached output.  (The clip interface below -- ``encode_source`` /
``forward_frames`` and everything built on it -- is the INFERENCE path by contract: it has no reference counterpart that could
carry a graph, always runs graph-free, and refuses an input that requires a gradient rather than detaching it.)

Beyond the reference interface the module exposes the two halves of forward separately
(``encode_source`` / ``forward_frames``) so that a clip can reuse the frame-invariant source
encoder (the reference recomputes it per frame, generator.py:61-63 inside demo.py:279's loop).
"""
from __future__ import annotations

import operator
import os
import warnings
from typing import Dict, Iterable, Optional

import torch
from torch import nn

from .engine import Engine
from .weights import antialias_kernel, generator_channels, hourglass_channels


_VERSION_OF = operator.attrgetter("_version")

class _Epoch:
    """Structure epoch of ONE generator: bumped whenever a module of its tree registers, replaces or deletes a parameter,
    buffer or sub-module.  The generator's cached list of tensor slots is rebuilt when the epoch has moved, so a replaced
    sub-module (``gen.bottleneck[0] = ...``) or a tensor registered later is seen -- at the cost of one integer compare per
    forward.  (Round 4 counted with torch's process-global module-registration hooks, installed at import and never removed:
    every module of the process paid for them.  Now the counting is done by the tree's own classes.)"""
    __slots__ = ("n",)

    def __init__(self):
        self.n = 0


class _Tracked:
    """Mixin (in front of an nn.Module class in the MRO): structural changes bump the owning generator's epoch.  A module that
    belongs to no generator (the key-point detectors use the holders too) has no cell and pays one dict lookup."""

    def _bump(self):
        cell = self.__dict__.get("_eamm_epoch")
        if cell is not None:
            cell.n += 1

    def __setattr__(self, name, value):
        d = self.__dict__
        if (isinstance(value, (nn.Parameter, nn.Module)) or name in ("_parameters", "_buffers", "_modules")
                or name in d.get("_parameters", ()) or name in d.get("_buffers", ()) or name in d.get("_modules", ())):
            self._bump()
        super().__setattr__(name, value)

    def __delattr__(self, name):
        self._bump()
        super().__delattr__(name)

    def register_parameter(self, name, param):
        self._bump()
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        self._bump()
        super().register_buffer(name, tensor, persistent=persistent)

    def add_module(self, name, module):
        self._bump()
        super().add_module(name, module)

    def register_module(self, name, module):
        self._bump()
        super().add_module(name, module)

    def _apply(self, fn, *args, **kwargs):
        # `.cuda()` / `.double()` / `.half()` on a SUB-module (`gen.first.double()`): `param.data = fn(param.data)` changes neither
        # the tensor's identity nor its version counter, so without the bump the engine would keep the old weights
        self._bump()
        return super()._apply(fn, *args, **kwargs)


class _TrackedContainer(_Tracked):
    """nn.ModuleList / nn.Sequential mutators that write ``self._modules`` directly (no registration call to intercept)."""

    def insert(self, index, module):
        self._bump()
        return super().insert(index, module)

    def pop(self, key):
        self._bump()
        return super().pop(key)

    def extend(self, modules):
        self._bump()
        return super().extend(modules)

    def append(self, module):
        self._bump()
        return super().append(module)

    def __setitem__(self, idx, module):
        self._bump()
        return super().__setitem__(idx, module)

    def __delitem__(self, idx):
        self._bump()
        return super().__delitem__(idx)


class _Conv2d(_Tracked, nn.Conv2d):
    pass


class _BatchNorm2d(_Tracked, nn.BatchNorm2d):
    pass


class _ModuleList(_TrackedContainer, nn.ModuleList):
    pass


class _Sequential(_TrackedContainer, nn.Sequential):
    pass


class _Holder(_Tracked, nn.Module):
    pass


class _ConvNorm(_Holder):
    """Parameter holder named like DownBlock2d / UpBlock2d / SameBlock2d (util.py:883-938)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = _Conv2d(cin, cout, kernel_size=k, padding=k // 2)
        self.norm = _BatchNorm2d(cout, affine=True)


class _ResHolder(_Holder):
    """Parameter holder named like ResBlock2d (util.py:858-870)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = _Conv2d(c, c, kernel_size=3, padding=1)
        self.conv2 = _Conv2d(c, c, kernel_size=3, padding=1)
        self.norm1 = _BatchNorm2d(c, affine=True)
        self.norm2 = _BatchNorm2d(c, affine=True)


class _Stack(_Holder):
    def __init__(self, name, blocks):
        super().__init__()
        setattr(self, name, _ModuleList(blocks))


class _AntiAlias(_Holder):
    """Holder of the 13x13 Gaussian buffer of AntiAliasInterpolation2d (util.py:1005-1042)."""

    def __init__(self, channels):
        super().__init__()
        self.register_buffer("weight", antialias_kernel(channels))


class _DenseMotionHolder(_Holder):
    """Parameters of DenseMotionNetwork under the reference's names (dense_motion.py:12-30)."""

    def __init__(self, block_expansion, num_blocks, max_features, num_kp, num_channels,
                 estimate_occlusion_map=False, scale_factor=1, kp_variance=0.01):
        super().__init__()
        enc, dec, out_filters = hourglass_channels(block_expansion, (num_kp + 1) * (num_channels + 1), num_blocks,
                                                   max_features)
        hg = _Holder()
        hg.encoder = _Stack("down_blocks", [_ConvNorm(ci, co, 3) for ci, co in enc])
        hg.decoder = _Stack("up_blocks", [_ConvNorm(ci, co, 3) for ci, co in dec])
        self.hourglass = hg
        self.mask = _Conv2d(out_filters, num_kp + 1, kernel_size=(7, 7), padding=(3, 3))
        self.occlusion = _Conv2d(out_filters, 1, kernel_size=(7, 7), padding=(3, 3)) if estimate_occlusion_map else None
        if scale_factor != 1:
            self.down = _AntiAlias(num_channels)
        self.num_kp, self.scale_factor, self.kp_variance = num_kp, scale_factor, kp_variance


class OcclusionAwareGenerator(_Tracked, nn.Module):
    """MI355X-native stand-in for reference modules/generator.py:OcclusionAwareGenerator."""

    def __init__(self, num_channels, num_kp, block_expansion, max_features, num_down_blocks,
                 num_bottleneck_blocks, estimate_occlusion_map=False, dense_motion_params=None,
                 estimate_jacobian=False, max_frames=16, cache_source=True):
        super().__init__()
        self._cfg = dict(num_channels=num_channels, num_kp=num_kp, block_expansion=block_expansion,
                         max_features=max_features, num_down_blocks=num_down_blocks,
                         num_bottleneck_blocks=num_bottleneck_blocks, estimate_occlusion_map=estimate_occlusion_map,
                         dense_motion_params=None if dense_motion_params is None else dict(dense_motion_params),
                         estimate_jacobian=estimate_jacobian)
        if dense_motion_params is not None:                      # generator.py:18-23
            self.dense_motion_network = _DenseMotionHolder(num_kp=num_kp, num_channels=num_channels,
                                                           estimate_occlusion_map=estimate_occlusion_map,
                                                           **dense_motion_params)
        else:
            self.dense_motion_network = None
        down, up, bott = generator_channels(self._cfg)
        self.first = _ConvNorm(num_channels, block_expansion, 7)
        self.down_blocks = _ModuleList([_ConvNorm(ci, co, 3) for ci, co in down])
        self.up_blocks = _ModuleList([_ConvNorm(ci, co, 3) for ci, co in up])
        self.bottleneck = _Sequential()
        for i in range(num_bottleneck_blocks):
            self.bottleneck.add_module("r" + str(i), _ResHolder(bott))
        self.final = _Conv2d(block_expansion, num_channels, kernel_size=(7, 7), padding=(3, 3))
        self.estimate_occlusion_map = estimate_occlusion_map
        self.num_channels = num_channels
        self.max_frames = int(max_frames)
        # forward() is called once per driving frame with the SAME source tensor (demo.py:279); re-encoding it is
        # skipped when the very same tensor object (identity, not address) is passed again unmodified (same autograd
        # version counter, which every in-place write bumps).  cache_source=False restores encode-per-call.
        self.cache_source = bool(cache_source)
        self._src_ref = None
        self._src_version = -1
        self._src_engine = None
        self._src_generation = -1
        self._engine: Optional[Engine] = None
        self._engine_key = None
        self._train_engine: Optional[Engine] = None
        self._train_key = None
        self._train_weight_changes = 0     # graph-free .train() forwards that found the convolution weights changed since the last
        # route of the graph-free (torch.no_grad()) .train() forward: "engine" = the library's resumable batch-statistics pass
        # (eamm_train_begin / eamm_train_next: one handle, saved plan, direct convolutions: 15.7 ms per 16 pairs at 256x256),
        # "operators" = the differentiable operator composition under no_grad (Winograd forms, fused BatchNorm: 9.6 ms; packs its
        # filters per call).  EAMM_TRAIN_ROUTE overrides the default.
        self.train_route = os.environ.get("EAMM_TRAIN_ROUTE", "engine")
        # widths that are not multiples of the kernels' 32-channel granule: the inference engine pads the state_dict into the
        # equivalent wider network at load time; the training-mode ENGINE does not (its BatchNorm kernels read the module's own
        # statistics tensors), so graph-free .train() forwards of such a generator take the operator composition
        widths = [c for pair in down + up for c in pair] + [bott]
        if dense_motion_params is not None:
            enc, dec, _ = hourglass_channels(dense_motion_params["block_expansion"], (num_kp + 1) * (num_channels + 1),
                                             dense_motion_params["num_blocks"], dense_motion_params["max_features"])
            widths += [co for _, co in enc] + [co for _, co in dec]
        if any(c % 32 for c in widths if c != num_channels) or num_channels != 3:
            self.train_route = "operators"
        # .train() mode: replicas for the BatchNorm statistics (None: the world group when torch.distributed runs with more
        # than one rank -- the analogue of DataParallel replicating the reference module); sync_batchnorm forces the
        # replicas' formula on or off (sync_batchnorm/batchnorm.py:48-53 vs :55-125)
        self.process_group = None
        self.sync_batchnorm: Optional[bool] = None
        # (parameters keep requires_grad=True as in the reference: the optimiser of train.py:136 sees every one of them)

    def __getstate__(self):
        """copy.deepcopy / pickling: the library handles (and the caches keyed on them) belong to THIS object; a copy builds its
        own at its first forward."""
        state = dict(self.__dict__)
        for k in ("_engine", "_engine_key", "_train_engine", "_train_key", "_src_ref", "_src_engine", "_slots", "_slots_epoch", "_fast_key"):
            if k in state:
                state[k] = None
        state["_src_version"] = state["_src_generation"] = -1
        return state

    # -- engine management ---------------------------------------------------------------------------
    def _epoch(self) -> "_Epoch":
        cell = self.__dict__.get("_eamm_epoch")
        if cell is None:
            cell = self.__dict__["_eamm_epoch"] = _Epoch()
        return cell

    def _adopt(self):
        """Hand every tracked module of the tree this generator's epoch cell; note whether the tree holds modules of other
        classes (a user's replacement block): their internal changes raise no epoch, so the fast path then also compares
        tensor identities."""
        cell = self._epoch()
        foreign = False
        for mod in self.modules():
            if isinstance(mod, _Tracked):
                mod.__dict__["_eamm_epoch"] = cell
            else:
                foreign = True
        self.__dict__["_foreign"] = foreign

    def _tensor_slots(self):
        """(owner dict, key, state_dict name) of every parameter and buffer, collected once: reading the LIVE dicts each call
        sees replaced tensors (``.cuda()``, ``load_state_dict(assign=True)``, attribute assignment) without rebuilding a
        ``state_dict`` per forward (340 us for the 196 tensors of the shipped configuration -- a third of a one-frame call)."""
        slots = self.__dict__.get("_slots")
        epoch = self._epoch()
        if slots is None or self.__dict__.get("_slots_epoch") != epoch.n:
            self._adopt()
            slots = []
            for prefix, mod in self.named_modules():
                for store, skip in ((mod._parameters, ()), (mod._buffers, mod._non_persistent_buffers_set)):
                    for key in store:            # (a slot that holds None today is kept: it may be assigned later)
                        if key not in skip:
                            slots.append((store, key, (prefix + "." if prefix else "") + key))
            self.__dict__["_slots"] = slots
            self.__dict__["_slots_epoch"] = epoch.n
        return slots

    def _weights_version(self):
        """Identity AND version of every parameter / buffer (a replaced tensor may carry the same version counter as the one it
        replaces).  The full tuple costs 40 us for the 224 tensors of the shipped configuration -- 4 % of a one-frame call -- so
        the hot path (``_weights_unchanged``) only adds up the version counters of the tensors seen last time: they never
        decrease, so an equal sum means no in-place write, and a REPLACED tensor shows as a changed structure epoch (attribute
        assignment in any module of the tree, ``load_state_dict`` / ``_apply``: overridden below to bump it) or a changed identity
        on the slow path."""
        ts = [t for t in (store[key] for store, key, _ in self._tensor_slots()) if t is not None]
        return tuple(map(id, ts)) + tuple(map(_VERSION_OF, ts))

    def _weights_unchanged(self) -> bool:
        fast = self.__dict__.get("_fast_key")
        if fast is None or fast[0] != self._epoch().n or sum(map(_VERSION_OF, fast[1])) != fast[2]:
            return False
        if self.__dict__.get("_foreign"):      # modules of other classes in the tree: their tensors may have been swapped silently
            live = [t for t in (store[key] for store, key, _ in self.__dict__["_slots"]) if t is not None]
            return len(live) == len(fast[1]) and all(a is b for a, b in zip(live, fast[1]))
        return True

    def _remember_weights(self):
        ts = [t for t in (store[key] for store, key, _ in self._tensor_slots()) if t is not None]
        self.__dict__["_fast_key"] = (self._epoch().n, ts, sum(map(_VERSION_OF, ts)))

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._bump()           # assign=True swaps buffers without a registration call; copies bump the version counters anyway
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._epoch().n += 1   # .cuda() / .to() / .float(): `param.data = fn(param.data)` changes neither identity nor version
        return out

    def _ensure_engine(self, height: int, width: int, frames: int, sources: int) -> Engine:
        dev = self.final.weight.device      # (next(self.parameters()) walks the module tree: 16 us per call)
        if dev.type != "cuda":
            raise RuntimeError("eamm_amd.OcclusionAwareGenerator runs only on a ROCm GPU: move the module with "
                               ".cuda() first (there is no CPU fallback for this path)")
        e = self._engine
        if (e is not None and self._engine_key is not None and self._engine_key[:3] == (dev, height, width) and
                e.max_frames >= frames and e.max_sources >= sources and self._weights_unchanged()):
            return e
        key = (dev, height, width, self._weights_version())
        self._remember_weights()
        if e is None or self._engine_key != key or e.max_frames < frames or e.max_sources < sources:
            if e is not None:
                e.close()
            e = Engine(self._cfg, height, width, max_frames=max(frames, self.max_frames),
                       max_sources=max(sources, 1), device=dev)
            e.load_state_dict(self.state_dict())
            self._engine, self._engine_key = e, key
        return e

    # -- .train(): batch statistics in every BatchNorm (sync_batchnorm/batchnorm.py:55-125) -----------------------------
    def _norm_modules(self):
        return {name: mod for name, mod in self.named_modules() if isinstance(mod, nn.BatchNorm2d)}

    def _ensure_train_engine(self, height: int, width: int, frames: int) -> Engine:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("eamm_amd.OcclusionAwareGenerator runs only on a ROCm GPU: move the module with "
                               ".cuda() first (there is no CPU fallback for this path)")
        # convolution weights only: the BatchNorm tensors are read (and the running statistics written) in place
        conv_version = tuple((id(store[key]), store[key]._version) for store, key, name in self._tensor_slots()
                             if ".norm" not in name and store[key] is not None)
        e = self._train_engine
        key = (dev, height, width, conv_version)
        if e is None or self._train_key != key or e.max_frames < frames:
            if e is not None:
                e.close()
            n = max(frames, self.max_frames)
            e = Engine(self._cfg, height, width, max_frames=n, max_sources=n, device=dev, training=True)
            e.load_state_dict(self.state_dict())
            self._train_engine, self._train_key = e, key
        return e

    def _replicas(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1

    def _all_reduce(self, t: torch.Tensor):
        import torch.distributed as dist
        if dist.get_backend(self.process_group) == "nccl":       # RCCL: device tensor, in place, over xGMI
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
        else:                                                    # gloo (tests): staged through the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.process_group)
            t.copy_(h)

    def _train_weights_keep_changing(self, height, width) -> bool:
        """A fine-tuning loop that mixes optimiser steps with graph-free .train() forwards (validation passes in training mode)
        would rebuild the training engine -- allocation, state_dict copy, filter repack: hundreds of milliseconds -- after every
        step.  From the second such change on, those forwards take the operator composition under no_grad instead
        (it packs filters per call on the device, 6-17 us per layer) and the idle engine is freed."""
        e = self._train_engine
        if e is None or self._train_key is None or self._train_key[1:3] != (height, width):
            return False
        conv_version = tuple((id(store[key]), store[key]._version) for store, key, name in self._tensor_slots()
                             if ".norm" not in name and store[key] is not None)
        if conv_version == self._train_key[3]:
            return False
        self._train_weight_changes += 1
        if self._train_weight_changes < 2:
            return False
        e.close()
        self._train_engine = self._train_key = None
        return True

    def _forward_train(self, source_image, kp_driving, kp_source):
        b, _, hh, ww = source_image.shape
        if self.train_route == "operators" or self._train_weight_changes >= 2 or self._train_weights_keep_changing(hh, ww):
            from . import train_graph
            out = train_graph.forward_train(self, source_image, kp_driving, kp_source)      # (the caller holds torch.no_grad())
            self._bump_running_stats()
            self._src_ref = None
            return {k: out[k] for k in ("mask", "sparse_deformed", "occlusion_map", "deformed", "prediction") if k in out}
        e = self._ensure_train_engine(hh, ww, b)
        world = self._replicas()
        sync = world > 1 if self.sync_batchnorm is None else bool(self.sync_batchnorm)
        want = ["prediction"]
        if self.dense_motion_network is not None:
            want += ["mask", "sparse_deformed", "deformed"]
            if self.estimate_occlusion_map:
                want.append("occlusion_map")
            kd = {k: kp_driving[k] for k in ("value", "jacobian") if k in kp_driving}
            ks = {k: kp_source[k] for k in ("value", "jacobian") if k in kp_source}
        else:
            kd = ks = None
        norms = self._norm_modules()
        out = e.train_forward(source_image, kd, ks, norms, outputs=want, sync=sync,
                              reduce=self._all_reduce if (sync and world > 1) else None)
        self._bump_running_stats()
        self._src_ref = None
        e.check_numeric()
        return {k: out[k] for k in ("mask", "sparse_deformed", "occlusion_map", "deformed", "prediction") if k in out}

    # -- the reference contract -----------------------------------------------------------------------
    def _any_parameter_requires_grad(self) -> bool:
        """``any(p.requires_grad ...)`` over the cached slot list (the list is rebuilt only when the structure epoch moves; the
        FLAGS are re-read on every call -- ``requires_grad_()`` changes neither a version counter nor the structure, so they
        cannot be cached -- which costs a walk over ~200 parameters, only when gradients are enabled)."""
        ps = [t for t in (store[key] for store, key, _ in self._tensor_slots()) if isinstance(t, nn.Parameter)]
        return any(p.requires_grad for p in ps)

    def _wants_graph(self, source_image, kp_driving, kp_source) -> bool:
        if not torch.is_grad_enabled():
            return False
        tensors = [source_image] + [v for kp in (kp_driving, kp_source) if kp for v in kp.values() if torch.is_tensor(v)]
        return any(t.requires_grad for t in tensors) or self._any_parameter_requires_grad()

    @staticmethod
    def _refuse_grad_inputs(what, *tensors):
        """The clip interface is graph-free by contract; an input that asks for a gradient would be silently detached."""
        if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
            raise RuntimeError(f"eamm_amd.OcclusionAwareGenerator.{what} is the graph-free inference path: an input requires a "
                               "gradient, which it cannot carry -- call forward() for a differentiable pass, or detach the input")

    def forward(self, source_image, kp_driving, kp_source):
        """Reference generator.py:59-97: batch of independent (source, kp_source, kp_driving) triples.  In ``.train()`` mode
        every BatchNorm normalises with the statistics of the batch and updates its running statistics, as the reference's
        blocks do (modules/util.py:858-938).  The outputs carry an autograd graph exactly when the reference's would: gradients
        enabled and a parameter (``requires_grad`` is PyTorch's default True) or an input that requires one -- then the forward
        is the composition of differentiable HIP operators of ``train_graph`` (train.py:133's ``loss.backward()``), with batch
        statistics in ``.train()`` and running statistics in ``.eval()``.  Under ``torch.no_grad()`` (demo.py:195) it is the
        graph-free engine."""
        if source_image.dim() != 4:
            raise RuntimeError(f"source_image must be [B,3,H,W], got {tuple(source_image.shape)}")
        if self._wants_graph(source_image, kp_driving, kp_source):
            from . import train_graph
            if not self.training and not self.__dict__.get("_warned_eval_graph"):
                self.__dict__["_warned_eval_graph"] = True
                warnings.warn("eamm_amd.OcclusionAwareGenerator: building an autograd graph in .eval() mode (gradients are enabled "
                              "and a parameter or an input requires one) -- the differentiable operator composition, not the folded "
                              "inference engine; wrap inference in torch.no_grad() as demo.py:195 does", stacklevel=2)
            out = train_graph.forward_train(self, source_image, kp_driving, kp_source)
            if self.training:
                self._bump_running_stats()
            self._src_ref = None
            return {k: out[k] for k in ("mask", "sparse_deformed", "occlusion_map", "deformed", "prediction") if k in out}
        with torch.no_grad():
            return self._forward_no_grad(source_image, kp_driving, kp_source)

    def _bump_running_stats(self):
        # the library wrote the running statistics through raw pointers: bump the tensors' version counters so that every
        # cache keyed on them (the evaluation engine's folded weights) sees the change
        stats = [t for m in self._norm_modules().values() for t in (m.running_mean, m.running_var)]
        try:
            torch._C._increment_version(stats)
        except (AttributeError, TypeError):   # older / newer torch without the list form: a no-op in-place write
            for t in stats:
                t.add_(0)

    def _forward_no_grad(self, source_image, kp_driving, kp_source):
        if self.training:
            return self._forward_train(source_image, kp_driving, kp_source)
        b, _, hh, ww = source_image.shape
        e = self._ensure_engine(hh, ww, b, b)
        fresh = not (self.cache_source and self._src_ref is source_image and self._src_version == source_image._version
                     and e.ns_cached == b and self._src_engine is e and self._src_generation == e.cache_generation)
        if fresh:
            e.encode_source(source_image)
            self._src_ref, self._src_version, self._src_engine = source_image, source_image._version, e
            self._src_generation = e.cache_generation
        want = ["prediction"]
        if self.dense_motion_network is not None:                # generator.py:64-86
            want += ["mask", "sparse_deformed", "deformed"]
            if self.estimate_occlusion_map:
                want.append("occlusion_map")
        if self.dense_motion_network is None:    # the reference never looks at the key points then (generator.py:64)
            kd = ks = {"value": torch.zeros(b, self._cfg["num_kp"], 2, dtype=torch.float32, device=source_image.device)}
        else:
            kd = {k: kp_driving[k] for k in ("value", "jacobian") if k in kp_driving}
            ks = {k: kp_source[k] for k in ("value", "jacobian") if k in kp_source}
        if kd["value"].shape[0] != b or ks["value"].shape[0] != b:
            raise RuntimeError("key-point batch size does not match source_image batch size")
        out = e.forward_frames(kd, ks, outputs=want)
        e.check_numeric()  # torch.inverse raises (and synchronises) on a singular jacobian, dense_motion.py:56
        return {k: out[k] for k in ("mask", "sparse_deformed", "occlusion_map", "deformed", "prediction") if k in out}

    # -- clip interface: encoder hoisted out of the frame loop ------------------------------------------
    def encode_source(self, source_image: torch.Tensor, max_frames: Optional[int] = None) -> Engine:
        self._refuse_grad_inputs("encode_source", source_image)
        with torch.no_grad():
            return self._encode_source(source_image, max_frames)

    def _encode_source(self, source_image: torch.Tensor, max_frames: Optional[int] = None) -> Engine:
        if self.training:
            raise RuntimeError("the clip interface (encode_source / forward_frames) is the inference path: BatchNorm uses "
                               "running statistics (sync_batchnorm/batchnorm.py:48-53); call .eval() as demo.py:105 does")
        ns, _, hh, ww = source_image.shape
        e = self._ensure_engine(hh, ww, max_frames or self.max_frames, ns)
        e.encode_source(source_image)
        self._src_ref = None   # the engine's cache no longer belongs to forward()'s last source
        return e

    def forward_frames(self, kp_driving: Dict[str, torch.Tensor], kp_source: Dict[str, torch.Tensor],
                       outputs: Iterable[str] = ("prediction",), uint8_frames: bool = False):
        self._refuse_grad_inputs("forward_frames", *kp_driving.values(), *kp_source.values())
        if self._engine is None or self._engine.ns_cached < 1:
            raise RuntimeError("call encode_source(source_image) first")
        with torch.no_grad():
            return self._engine.forward_frames(kp_driving, kp_source, outputs=outputs, uint8_frames=uint8_frames)

    @property
    def engine(self) -> Optional[Engine]:
        return self._engine
