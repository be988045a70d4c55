"""Drop-in ``SynchronizedBatchNorm2d`` whose forward runs in libeamm_hip.so -- SURVEY.md section 8f row N4, FIRST SLICE:
the training-mode forward (batch statistics, cross-replica reduction, running-statistics update) and the evaluation
forward.  The backward pass is not built: outputs carry no autograd graph.

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
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(_lib.lib().eamm_bn_finalize(ptr(sums), c, float(mod.eps), float(mod.momentum), mode, ptr(mod.weight),
                                               ptr(mod.running_mean), ptr(mod.running_var), ptr(mean), ptr(scale),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return mean, scale

    def apply(self, x: torch.Tensor, mean: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(_lib.lib().eamm_bn_apply(C.c_void_p(x.data_ptr()), C.c_void_p(mean.data_ptr()), C.c_void_p(scale.data_ptr()),
                                            None if bias is None else C.c_void_p(bias.data_ptr()), n, c, hw,
                                            C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return y


class SynchronizedBatchNorm2d(_BatchNorm):
    """MI355X-native stand-in for reference sync_batchnorm/batchnorm.py:SynchronizedBatchNorm2d (forward only).

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

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        self._check_input_dim(input)
        if input.shape[1] != self.num_features:
            raise RuntimeError(f"expected {self.num_features} channels, got {input.shape[1]}")
        self._check_device(input)
        x = input.contiguous()
        if not self.training:                                            # batchnorm.py:48-53, eval branch
            mean, scale = self._ops.finalize(None, self, BN_EVAL)
            return self._ops.apply(x, mean, scale, self.bias)
        world = self._replicas()
        parallel = world > 1 if self.sync is None else bool(self.sync)
        sums = self._ops.local_sums(x)                                   # batchnorm.py:61-64
        if world > 1 and parallel:                                       # batchnorm.py:66-70, 102-105
            self._all_reduce(sums[:2 * self.num_features + 2])
        mean, scale = self._ops.finalize(sums, self, BN_SYNC if parallel else BN_SINGLE)   # batchnorm.py:110-125
        return self._ops.apply(x, mean, scale, self.bias)                # batchnorm.py:72-79

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
