"""eamm_amd.all_reduce_gradients: the parameter-gradient sum of data-parallel fine-tuning (reference: nn.DataParallel's
ReduceAddCoalesced in train.py's DataParallelWithCallback), in flat buckets, on two gloo ranks (CPU)."""
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eamm_amd import all_reduce_gradients
from eamm_amd.data_parallel import _buckets


def test_buckets_respect_the_cap_and_the_order():
    ts = [torch.zeros(n) for n in (10, 300, 5, 5, 1000, 2)]
    b = _buckets(ts, cap_bytes=4 * 320)
    assert [[t.numel() for t in g] for g in b] == [[10, 300, 5, 5], [1000], [2]]   # a tensor above the cap gets its own bucket
    assert sum(len(g) for g in b) == len(ts)
    assert all_reduce_gradients([torch.nn.Parameter(torch.zeros(3))]) == 0         # outside torch.distributed: a no-op


def _worker(rank, world, port, out):
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 3), (1000,), (2, 2, 2), (64, 64), (5,))]
    frozen = torch.nn.Parameter(torch.zeros(4))                     # no gradient on any rank: skipped
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1)) + torch.arange(p.numel(), dtype=torch.float32).view_as(p)
    n = all_reduce_gradients(params + [frozen], bucket_mb=0.004, average=(rank >= 0 and out["average"]))
    want_scale = 1.0 / world if out["average"] else 1.0
    for i, p in enumerate(params):
        want = (sum(float(r + 1) * (i + 1) for r in range(world)) + world * torch.arange(p.numel(), dtype=torch.float32).view_as(p)) * want_scale
        assert torch.allclose(p.grad, want), (rank, i)
    assert frozen.grad is None and n >= 2                            # 4 KB buckets: several collectives
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("average", [False, True])
def test_two_ranks_sum_their_gradients_in_buckets(average):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, {"average": average}), nprocs=2, join=True)
