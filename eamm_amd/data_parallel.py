"""Data-parallel fine-tuning of the generator across GPUs: one process per GPU, every rank runs
``generator.train()`` on its shard of the batch.

The reference replicates the generator with ``nn.DataParallel`` inside one process (train.py: ``DataParallelWithCallback``):
the replicas' BatchNorm statistics meet in ``SyncMaster`` and autograd's ``ReduceAddCoalesced`` adds their parameter gradients
on the master.  Here the statistics -- and the backward's (sum dy, sum dy * xhat) -- are all-reduced per BatchNorm site by the
BatchNorm operator itself (``sync_batchnorm.py``); what is left is the parameter gradients' sum, and that is this module:
``all_reduce_gradients`` adds the ranks' ``.grad`` tensors in a few large flat buckets -- xGMI is point to point, a ring
all-reduce is bound per link, so few large collectives (64 MB default: 181 MB of generator gradients in three) beat one per
parameter -- over RCCL (``nccl`` backend) or, in tests, gloo.  Like the reference's ReduceAdd it SUMS (the loss of a
DataParallel run is already a mean over the whole batch on the master; divide by the world size afterwards with
``average=True`` when every rank's loss is a mean over its own shard)."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _buckets(tensors: List[torch.Tensor], cap_bytes: int) -> List[List[torch.Tensor]]:
    out, cur, size = [], [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if cur and (size + nbytes > cap_bytes or t.dtype != cur[0].dtype or t.device != cur[0].device):
            out.append(cur)
            cur, size = [], 0
        cur.append(t)
        size += nbytes
    if cur:
        out.append(cur)
    return out


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                         bucket_mb: float = 64.0, average: bool = False) -> int:
    """Sum (or average) the ``.grad`` of ``parameters`` over the ranks of ``group``, in place, in flat buckets of about
    ``bucket_mb`` megabytes.  Every rank must hold gradients for the same parameters in the same order (a parameter whose
    ``.grad`` is None on one rank must be None on all).  Returns the number of collectives issued; a no-op outside
    ``torch.distributed`` or on a single rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
    world = dist.get_world_size(group)
    stage_on_host = dist.get_backend(group) != "nccl"     # gloo (tests): device tensors travel through the host
    n = 0
    for bucket in _buckets(grads, int(bucket_mb * (1 << 20))):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        buf = flat.cpu() if (stage_on_host and flat.device.type != "cpu") else flat
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if buf is not flat:
            flat.copy_(buf)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n += 1
    return n
