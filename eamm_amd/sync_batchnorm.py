"""Drop-in ``SynchronizedBatchNorm2d`` whose forward AND backward run in libeamm_hip.so -- SURVEY.md section 8f row N4:
the training-mode forward (batch statistics, cross-replica reduction, running-statistics update), the evaluation forward,
and (round 3) the gradient of both as a ``torch.autograd.Function`` -- the output is differentiable with respect to the
input, ``weight`` and ``bias`` exactly as the reference module's is.

Mirrors reference sync_batchnorm/batchnorm.py:38-125 (``_SynchronizedBatchNorm``): same constructor, same parameter and
buffer names (it IS a ``torch.nn.modules.batchnorm._BatchNorm``), same three behaviours --

* evaluation, or training on a single replica: ``F.batch_norm`` semantics (batchnorm.py:48-53) -- biased variance + eps
  under the square root, running statistics updated with the unbiased variance;
* training on several replicas (batchnorm.py:55-125): every replica computes per-channel sum and sum of squares, they
  are added over the replicas, mean / ``clamp(biased variance, eps) ** -0.5`` come from the totals, the running statistics
  are updated once from the global statistics, every replica normalises its own shard.

The reference's replicas are ``nn.DataParallel`` threads exchanging tensors through a master (``SyncMaster``,
``ReduceAddCoalesced``, ``Broadcast``); here a replica is a process of a ``torch.distributed`` group (one per GPU) and the
exchange is ONE all-reduce of 2C + 2 floats over RCCL/xGMI (the two extra floats carry the per-channel element count, so
unequal shards work as they do in the reference).  Every rank applies the same running-statistics update, so the buffers
stay identical on all ranks without a broadcast (the reference keeps them on the master only).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist
from torch.nn.modules.batchnorm import _BatchNorm

from . import _lib

BN_SYNC, BN_SINGLE, BN_EVAL = 0, 1, 2


def _check(code: int):
    if code != _lib.EAMM_OK:
        msg = _lib.lib().eamm_bn_last_error()
        raise _lib.EammError(code, msg.decode() if msg else "?")


class HipBatchNormOps:
    """The three device steps of the forward (C ABI: eamm_bn_local_sums / eamm_bn_finalize / eamm_bn_apply)."""

    def check(self, input: torch.Tensor, mod: "SynchronizedBatchNorm2d"):
        if input.device.type != "cuda":
            raise RuntimeError("eamm_amd.SynchronizedBatchNorm2d runs only on a ROCm GPU (there is no CPU fallback)")
        if input.dtype != torch.float32:
            raise RuntimeError(f"input must be float32, got {input.dtype}")
        if mod.running_mean is None or mod.running_mean.device != input.device:
            raise RuntimeError("module buffers and input are on different devices (move the module with .cuda())")

    def local_sums(self, x: torch.Tensor) -> torch.Tensor:
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        L = _lib.lib()
        sums = torch.empty(6 * c + 2, dtype=torch.float32, device=x.device)   # [2C+2] exchanged floats + 2C doubles
        work = torch.empty(max(1, L.eamm_bn_workspace_floats(n, c, hw)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _check(L.eamm_bn_local_sums(C.c_void_p(x.data_ptr()), n, c, hw, C.c_void_p(sums.data_ptr()),
                                        C.c_void_p(work.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return sums

    def finalize(self, sums: Optional[torch.Tensor], mod: "SynchronizedBatchNorm2d", mode: int):
        c, dev = mod.num_features, mod.running_mean.device
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        scale = torch.empty(c, dtype=torch.float32, device=dev)
        inv_std = torch.empty(c, dtype=torch.float32, device=dev)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_bn_finalize(ptr(sums), c, float(mod.eps), float(mod.momentum), mode, ptr(mod.weight),
                                               ptr(mod.running_mean), ptr(mod.running_var), ptr(mean), ptr(scale), ptr(inv_std),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return mean, scale, inv_std

    # ---- backward: eamm_bn_backward_sums / _finalize / _apply ------------------------------------------------------
    def backward_sums(self, x: torch.Tensor, dy: torch.Tensor, mean: torch.Tensor) -> torch.Tensor:
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        L = _lib.lib()
        sums = torch.empty(6 * c + 2, dtype=torch.float32, device=x.device)
        work = torch.empty(max(1, L.eamm_bn_workspace_floats(n, c, hw)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _check(L.eamm_bn_backward_sums(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(mean.data_ptr()), n, c, hw,
                                           C.c_void_p(sums.data_ptr()), C.c_void_p(work.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return sums

    def backward_finalize(self, local: torch.Tensor, reduced: torch.Tensor, inv_std: torch.Tensor, weight: Optional[torch.Tensor],
                          eps: float, mode: int, want_wb: bool):
        c = inv_std.numel()
        coef = torch.empty(3 * c, dtype=torch.float32, device=inv_std.device)
        dw = torch.empty(c, dtype=torch.float32, device=inv_std.device) if want_wb else None
        db = torch.empty(c, dtype=torch.float32, device=inv_std.device) if want_wb else None
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(inv_std.device):
            _check(_lib.lib().eamm_bn_backward_finalize(ptr(local), ptr(reduced), c, ptr(inv_std), ptr(weight), float(eps), mode,
                                                        ptr(dw), ptr(db), ptr(coef), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return coef, dw, db

    def backward_apply(self, x: torch.Tensor, dy: torch.Tensor, mean: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(_lib.lib().eamm_bn_backward_apply(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(mean.data_ptr()),
                                                     C.c_void_p(coef.data_ptr()), n, c, hw, C.c_void_p(dx.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return dx

    def apply(self, x: torch.Tensor, mean: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(_lib.lib().eamm_bn_apply(C.c_void_p(x.data_ptr()), C.c_void_p(mean.data_ptr()), C.c_void_p(scale.data_ptr()),
                                            None if bias is None else C.c_void_p(bias.data_ptr()), n, c, hw,
                                            C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return y


class _BatchNormFunction(torch.autograd.Function):
    """y = BatchNorm(x) with the statistics chosen by the module's state; forward and backward are the library's kernels.

    The gradient is the one autograd derives from the reference's forward (sync_batchnorm/batchnorm.py:61-79, 110-125), whose
    ReduceAddCoalesced / Broadcast carry it across the replicas: dx needs sum dy and sum dy * xhat over ALL replicas -- one
    all-reduce of 2C + 2 floats, like the forward's -- while dweight / dbias are this replica's sums (the reference adds the
    replicas' parameter gradients on the DataParallel master; with one process per replica that addition is the
    gradient all-reduce of DistributedDataParallel or of the training loop)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        ops = mod._ops
        if not mod.training:                                            # batchnorm.py:48-53, eval branch
            mode = BN_EVAL
            mean, scale, inv_std = ops.finalize(None, mod, BN_EVAL)
        else:
            world = mod._replicas()
            parallel = world > 1 if mod.sync is None else bool(mod.sync)
            sums = ops.local_sums(x)                                     # batchnorm.py:61-64
            if world > 1 and parallel:                                   # batchnorm.py:66-70, 102-105
                mod._all_reduce(sums[:2 * mod.num_features + 2])
            mode = BN_SYNC if parallel else BN_SINGLE
            mean, scale, inv_std = ops.finalize(sums, mod, mode)         # batchnorm.py:110-125
        y = ops.apply(x, mean, scale, bias)                              # batchnorm.py:72-79
        ctx.mod, ctx.mode = mod, mode
        ctx.reduce = mode == BN_SYNC and mod._replicas() > 1
        ctx.save_for_backward(x, mean, inv_std, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, inv_std, weight = ctx.saved_tensors
        mod, ops = ctx.mod, ctx.mod._ops
        dy = dy.contiguous()
        local = ops.backward_sums(x, dy, mean)
        reduced = local
        if ctx.reduce:
            reduced = local.clone()
            mod._all_reduce(reduced[:2 * mod.num_features + 2])
        want_wb = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        coef, dw, db = ops.backward_finalize(local, reduced, inv_std, weight, mod.eps, ctx.mode, want_wb)
        dx = ops.backward_apply(x, dy, mean, coef) if ctx.needs_input_grad[0] else None
        return dx, (dw if ctx.needs_input_grad[1] else None), (db if ctx.needs_input_grad[2] else None), None


class _BatchNormNHWCFunction(torch.autograd.Function):
    """The same module on an NHWC activation [B,H,W,C], fused with the tail of the reference's blocks (modules/util.py:858-938):
    y = [avgpool2x2](relu(BatchNorm(x))) -- ``eamm_bn_nhwc_*``.  Training mode only (the differentiable generator forward of
    ``train_graph``); statistics, running-statistics update, the replicas' two all-reduces and the parameter gradients are
    ``_BatchNormFunction``'s.  Only x is kept for the backward: the ReLU mask is recomputed from it."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, relu, pool):
        L = _lib.lib()
        b, h, w, c = x.shape
        m = b * h * w
        dev = x.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        world = mod._replicas()
        parallel = world > 1 if mod.sync is None else bool(mod.sync)
        sums = torch.empty(6 * c + 2, dtype=torch.float32, device=dev)
        work = torch.empty(max(1, L.eamm_bn_nhwc_workspace_floats(m, c)), dtype=torch.float32, device=dev)
        mode = BN_SYNC if parallel else BN_SINGLE
        with torch.cuda.device(dev):
            if world > 1 and parallel:
                _check(L.eamm_bn_nhwc_local_sums(ptr(x), m, c, ptr(sums), ptr(work), st))
                mod._all_reduce(sums[:2 * c + 2])
                mean, scale, inv_std = mod._ops.finalize(sums, mod, mode)
            else:   # one replica: nothing to exchange -- the finalize step runs inside the kernel that adds the slices up
                mean, scale, inv_std = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
                _check(L.eamm_bn_nhwc_local_stats(ptr(x), m, c, float(mod.eps), float(mod.momentum), mode, ptr(mod.weight),
                                                  ptr(mod.running_mean), ptr(mod.running_var), ptr(sums), ptr(mean), ptr(scale),
                                                  ptr(inv_std), ptr(work), st))
            y = torch.empty((b, h // 2, w // 2, c) if pool else (b, h, w, c), dtype=torch.float32, device=dev)
            _check(L.eamm_bn_nhwc_apply(ptr(x), ptr(mean), ptr(scale), ptr(bias), b, h, w, c, int(relu), int(pool), ptr(y), st))
        ctx.mod, ctx.mode, ctx.relu, ctx.pool = mod, mode, bool(relu), bool(pool)
        ctx.reduce = mode == BN_SYNC and world > 1
        ctx.save_for_backward(x, mean, scale, inv_std, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, scale, inv_std, weight, bias = ctx.saved_tensors
        mod = ctx.mod
        L = _lib.lib()
        b, h, w, c = x.shape
        dev = x.device
        dy = dy.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        local = torch.empty(6 * c + 2, dtype=torch.float32, device=dev)
        work = torch.empty(max(1, L.eamm_bn_nhwc_workspace_floats(b * h * w, c)), dtype=torch.float32, device=dev)
        want_wb = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        with torch.cuda.device(dev):
            if ctx.reduce:
                _check(L.eamm_bn_nhwc_backward_sums(ptr(x), ptr(dy), ptr(mean), ptr(scale), ptr(bias), b, h, w, c, int(ctx.relu),
                                                    int(ctx.pool), ptr(local), ptr(work), st))
                reduced = local.clone()
                mod._all_reduce(reduced[:2 * c + 2])
                coef, dw, db = mod._ops.backward_finalize(local, reduced, inv_std, weight, mod.eps, ctx.mode, want_wb)
            else:   # one replica: sums and their finalize step in one call
                coef = torch.empty(3 * c, dtype=torch.float32, device=dev)
                dw = torch.empty(c, dtype=torch.float32, device=dev) if want_wb else None
                db = torch.empty(c, dtype=torch.float32, device=dev) if want_wb else None
                _check(L.eamm_bn_nhwc_backward_local(ptr(x), ptr(dy), ptr(mean), ptr(scale), ptr(bias), b, h, w, c, int(ctx.relu),
                                                     int(ctx.pool), ptr(inv_std), ptr(weight), float(mod.eps), ctx.mode, ptr(local),
                                                     ptr(dw), ptr(db), ptr(coef), ptr(work), st))
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(L.eamm_bn_nhwc_backward_apply(ptr(x), ptr(dy), ptr(mean), ptr(scale), ptr(bias), ptr(coef), b, h, w, c,
                                                     int(ctx.relu), int(ctx.pool), ptr(dx), st))
        return dx, (dw if ctx.needs_input_grad[1] else None), (db if ctx.needs_input_grad[2] else None), None, None, None


class SynchronizedBatchNorm2d(_BatchNorm):
    """MI355X-native stand-in for reference sync_batchnorm/batchnorm.py:SynchronizedBatchNorm2d (forward and backward).

    ``process_group``: the replicas (default: the world group when ``torch.distributed`` is initialised with more than one
    rank -- the analogue of the reference's ``_is_parallel`` flag, set when ``DataParallel`` replicates the module).
    ``sync=True`` forces the replicas' formula (``clamp(var, eps) ** -0.5``) on a single rank too.
    """

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, process_group=None, sync: Optional[bool] = None):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        if momentum is None:
            raise ValueError("momentum=None (cumulative average) is not part of the reference module")
        self.process_group = process_group
        self.sync = sync
        self._ops = HipBatchNormOps()

    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError("expected 4D input (got {}D input)".format(input.dim()))

    def _replicas(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        self._check_input_dim(input)
        if input.shape[1] != self.num_features:
            raise RuntimeError(f"expected {self.num_features} channels, got {input.shape[1]}")
        self._check_device(input)
        return _BatchNormFunction.apply(input.contiguous(), self.weight, self.bias, self)

    def _check_device(self, input: torch.Tensor):
        self._ops.check(input, self)
        # batchnorm.py:112 asserts size > 1 on the reduced count; with R >= 2 non-empty shards it cannot fire, on one
        # replica the count is known from the shape (no host read of the device-side count, no stream synchronisation)
        if self.training and self._replicas() == 1 and input.numel() // input.shape[1] <= 1:
            raise AssertionError("BatchNorm computes unbiased standard-deviation, which requires size > 1.")

    def _all_reduce(self, sums: torch.Tensor):
        if dist.get_backend(self.process_group) == "nccl":               # RCCL: device tensor, in place, over xGMI
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.process_group)
        else:                                                            # gloo (tests): staged through the host
            h = sums.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.process_group)
            sums.copy_(h)                                                # (a view of the first 2C+2 floats: in place)
