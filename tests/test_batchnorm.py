"""Training-mode BatchNorm forward (first slice) -- reference sync_batchnorm/batchnorm.py:46-125.

CPU: the oracle against the fixture produced by the reference's own SynchronizedBatchNorm2d (evaluation, one replica,
two replicas with unequal shards driven through the reference's SyncMaster protocol), and the module's replica protocol
under a two-rank gloo group with a stand-in for the three device steps.
GPU (-m gpu): eamm_amd.SynchronizedBatchNorm2d (HIP kernels through the C ABI) against the same fixture and, at the
generator's sizes and awkward shapes, against float64 statistics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN
from oracle import eamm_oracle as orc

TOL_OUT, TOL_STAT = 2e-5, 2e-6   # max abs: normalised outputs are O(1); reference fp32-vs-fp64 floor ~1e-6


def fixture():
    z = np.load(os.path.join(GOLDEN, "batchnorm_train.npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else int(z[k]) for k in z.files}


def test_oracle_matches_reference_fixture():
    g = fixture()
    p = (g["weight"], g["bias"], g["running_mean"], g["running_var"])
    o, rm, rv = orc.sync_batchnorm_forward([g["x"]], *p, training=False)
    assert float((o[0] - g["eval_out"]).abs().max()) <= 1e-6 and torch.equal(rm, g["running_mean"])
    o, rm, rv = orc.sync_batchnorm_forward([g["x"]], *p)
    assert float((o[0] - g["single_out"]).abs().max()) <= 1e-6
    assert float((rm - g["single_running_mean"]).abs().max()) <= 1e-7 and float((rv - g["single_running_var"]).abs().max()) <= 1e-6
    k = g["split"]
    o, rm, rv = orc.sync_batchnorm_forward([g["x"][:k], g["x"][k:]], *p)
    assert torch.equal(torch.cat(o, 0), g["sync_out"])            # the same float32 operations in the same order
    assert torch.equal(rm, g["sync_running_mean"]) and torch.equal(rv, g["sync_running_var"])


class OracleOps:
    """CPU stand-in for HipBatchNormOps (tests only): same three steps, same `sums` layout."""

    def check(self, input, mod):
        pass

    def local_sums(self, x):
        c = x.shape[1]
        v = x.reshape(x.shape[0], c, -1)
        n = v.shape[0] * v.shape[2]
        return torch.cat([v.sum(0).sum(-1), (v ** 2).sum(0).sum(-1), torch.tensor([float(n % 4096), float(n // 4096)]),
                          torch.zeros(4 * c)])

    def finalize(self, sums, mod, mode):
        c = mod.num_features
        assert mode == 0
        size = float(sums[2 * c] + 4096.0 * sums[2 * c + 1])
        mean = sums[:c] / size
        sumvar = sums[c:2 * c] - sums[:c] * mean
        mod.running_mean.copy_((1 - mod.momentum) * mod.running_mean + mod.momentum * mean)
        mod.running_var.copy_((1 - mod.momentum) * mod.running_var + mod.momentum * sumvar / (size - 1))
        return mean, (sumvar / size).clamp(mod.eps) ** -0.5 * mod.weight

    def apply(self, x, mean, scale, bias):
        return (x - mean[None, :, None, None]) * scale[None, :, None, None] + bias[None, :, None, None]


def _load(mod, g):
    with torch.no_grad():
        mod.weight.copy_(g["weight"]); mod.bias.copy_(g["bias"])
        mod.running_mean.copy_(g["running_mean"]); mod.running_var.copy_(g["running_var"])
    return mod


def _cpu_worker(rank, world, port, tmp):
    from eamm_amd.sync_batchnorm import SynchronizedBatchNorm2d
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = fixture()
    k = g["split"]
    mod = _load(SynchronizedBatchNorm2d(g["x"].shape[1]), g).train()
    mod._ops = OracleOps()
    out = mod(g["x"][:k] if rank == 0 else g["x"][k:])        # unequal shards: 4 and 2 images
    np.save(os.path.join(tmp, f"out{rank}.npy"), out.numpy())
    np.save(os.path.join(tmp, f"rm{rank}.npy"), mod.running_mean.numpy())
    np.save(os.path.join(tmp, f"rv{rank}.npy"), mod.running_var.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol_on_cpu(tmp_path):
    """The N > 1 path on CPU: two gloo ranks, unequal shards, ONE all-reduce of 2C+2 floats (sums + element count);
    outputs and running statistics must equal the reference's two-replica run, on BOTH ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_cpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = fixture()
    out = np.concatenate([np.load(tmp_path / "out0.npy"), np.load(tmp_path / "out1.npy")])
    assert np.abs(out - g["sync_out"].numpy()).max() <= 1e-6
    for r in (0, 1):
        assert np.abs(np.load(tmp_path / f"rm{r}.npy") - g["sync_running_mean"].numpy()).max() <= 1e-7
        assert np.abs(np.load(tmp_path / f"rv{r}.npy") - g["sync_running_var"].numpy()).max() <= 1e-6


def test_module_interface_without_gpu():
    from eamm_amd import SynchronizedBatchNorm2d
    m = SynchronizedBatchNorm2d(8)
    assert sorted(m.state_dict()) == ["bias", "num_batches_tracked", "running_mean", "running_var", "weight"]
    assert m.eps == 1e-5 and m.momentum == 0.1 and m.training
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 8, 4, 4))              # CPU tensor: no fallback
    with pytest.raises(ValueError):
        m(torch.zeros(2, 8, 4))


# ---------------------------------------------------------------------------------------------------------------------
DEV = "cuda:0"


@pytest.mark.gpu
def test_gpu_module_matches_reference_fixture():
    from eamm_amd import SynchronizedBatchNorm2d
    g = fixture()
    x = g["x"].to(DEV)
    m = _load(SynchronizedBatchNorm2d(x.shape[1]), g).to(DEV).eval()
    assert float((m(x).cpu() - g["eval_out"]).abs().max()) <= TOL_OUT
    assert torch.equal(m.running_mean.cpu(), g["running_mean"])                 # evaluation updates nothing
    m = _load(SynchronizedBatchNorm2d(x.shape[1]), g).to(DEV).train()
    out = m(x)
    assert float((out.cpu() - g["single_out"]).abs().max()) <= TOL_OUT
    assert float((m.running_mean.cpu() - g["single_running_mean"]).abs().max()) <= TOL_STAT
    assert float((m.running_var.cpu() - g["single_running_var"]).abs().max()) <= TOL_STAT
    assert int(m.num_batches_tracked) == 0      # the reference's forward never touches it (batchnorm.py:46-82)
    # the replicas' formula on one rank: whole batch as one shard == the two-replica statistics of the fixture
    m = _load(SynchronizedBatchNorm2d(x.shape[1], sync=True), g).to(DEV).train()
    out = m(x)
    assert float((out.cpu() - g["sync_out"]).abs().max()) <= TOL_OUT
    assert float((m.running_var.cpu() - g["sync_running_var"]).abs().max()) <= TOL_STAT


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 256, 64, 64), (1, 64, 256, 256), (3, 5, 7, 9), (2, 33, 1, 1), (5, 128, 32, 32)])
def test_gpu_statistics_against_float64(shape):
    """Generator-sized and awkward shapes (odd plane sizes take the scalar path, 1x1 planes, one image): batch statistics
    and outputs against float64 arithmetic, running-statistics update with the unbiased variance."""
    from eamm_amd import SynchronizedBatchNorm2d
    rs = np.random.RandomState(sum(shape))
    c = shape[1]
    x = torch.from_numpy((rs.standard_normal(shape) * rs.uniform(0.5, 2, (1, c, 1, 1)) + rs.standard_normal((1, c, 1, 1))).astype(np.float32))
    m = SynchronizedBatchNorm2d(c).to(DEV).train()
    out = m(x.to(DEV)).cpu()
    xd = x.double()
    mean, var = xd.mean(dim=(0, 2, 3)), xd.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() // c
    ref = (xd - mean[None, :, None, None]) / torch.sqrt(var + 1e-5)[None, :, None, None]
    assert float((out.double() - ref).abs().max()) <= 5e-5
    assert float((m.running_mean.cpu().double() - 0.1 * mean).abs().max()) <= 1e-6
    assert float((m.running_var.cpu().double() - (0.9 + 0.1 * var * n / (n - 1))).abs().max()) <= 1e-5
    sums = m._ops.local_sums(x.to(DEV)).cpu().double()
    assert float(sums[2 * c] + 4096 * sums[2 * c + 1]) == n
    assert float(((sums[:c] - xd.sum(dim=(0, 2, 3))) / (xd.abs().sum(dim=(0, 2, 3)) + 1)).abs().max()) <= 2e-6


def _gpu_worker(rank, world, port, tmp):
    from eamm_amd import SynchronizedBatchNorm2d
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    g = fixture()
    k = g["split"]
    mod = _load(SynchronizedBatchNorm2d(g["x"].shape[1]), g).to(DEV).train()
    out = mod((g["x"][:k] if rank == 0 else g["x"][k:]).to(DEV))
    np.save(os.path.join(tmp, f"out{rank}.npy"), out.cpu().numpy())
    np.save(os.path.join(tmp, f"rv{rank}.npy"), mod.running_var.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_two_ranks_unequal_shards(tmp_path):
    """Two processes sharing GPU 0 under a gloo group, the real HIP kernels, shards of 4 and 2 images: the all-reduced
    statistics must reproduce the reference's two-replica outputs and running statistics."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = fixture()
    out = np.concatenate([np.load(tmp_path / "out0.npy"), np.load(tmp_path / "out1.npy")])
    assert np.abs(out - g["sync_out"].numpy()).max() <= TOL_OUT
    for r in (0, 1):
        assert np.abs(np.load(tmp_path / f"rv{r}.npy") - g["sync_running_var"].numpy()).max() <= TOL_STAT
