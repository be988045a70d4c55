"""Training-mode BatchNorm forward and backward -- reference sync_batchnorm/batchnorm.py:46-125.

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


def test_oracle_gradients_match_reference_fixture():
    """Round 3: the oracle under autograd against the gradients of the reference modules (evaluation, one replica, and two
    replicas whose gradient crosses the reference's SyncMaster exchange); parameter gradients are per replica."""
    g = fixture()
    k, dy = g["split"], g["dy"]
    for key, shards, training in (("eval", [g["x"]], False), ("single", [g["x"]], True), ("sync", [g["x"][:k], g["x"][k:]], True)):
        xin = [t.clone().requires_grad_(True) for t in shards]
        ws = [g["weight"].clone().requires_grad_(True) for _ in shards]
        bs = [g["bias"].clone().requires_grad_(True) for _ in shards]
        w_arg, b_arg = (ws[0], bs[0]) if len(shards) == 1 else (ws, bs)
        o, _, _ = orc.sync_batchnorm_forward(xin, w_arg, b_arg, g["running_mean"], g["running_var"], training=training)
        (torch.cat(o, 0) * dy).sum().backward()
        assert float((torch.cat([t.grad for t in xin], 0) - g[key + "_dx"]).abs().max()) <= 2e-6
        for r in range(len(shards)):
            sfx = "" if len(shards) == 1 else str(r)
            assert float((ws[r].grad - g[f"{key}_dw{sfx}"]).abs().max()) <= 2e-5 and float((bs[r].grad - g[f"{key}_db{sfx}"]).abs().max()) <= 2e-5


class OracleOps:
    """CPU stand-in for HipBatchNormOps (tests only): same steps, same `sums` / `coef` layouts."""

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
        inv_std = (sumvar / size).clamp(mod.eps) ** -0.5
        return mean, inv_std * mod.weight.detach(), inv_std

    def backward_sums(self, x, dy, mean):
        c = x.shape[1]
        n = x.numel() // c
        d = x - mean[None, :, None, None]
        sa, sb = dy.sum(dim=(0, 2, 3)), (dy * d).sum(dim=(0, 2, 3))
        exact = torch.cat([sa.double(), sb.double()]).view(torch.float32)       # the 2C doubles behind the exchanged floats
        return torch.cat([sa, sb, torch.tensor([float(n % 4096), float(n // 4096)]), exact])

    def backward_finalize(self, local, reduced, inv_std, weight, eps, mode, want_wb):
        c = inv_std.numel()
        assert mode == 0
        size = float(reduced[2 * c] + 4096.0 * reduced[2 * c + 1])
        clamped = inv_std >= eps ** -0.5
        coef = torch.cat([reduced[:c] / size, torch.where(clamped, torch.zeros(c), reduced[c:2 * c] * inv_std * inv_std / size),
                          inv_std * weight.detach()])
        return coef, local[c:2 * c] * inv_std, local[:c].clone()

    def backward_apply(self, x, dy, mean, coef):
        c = x.shape[1]
        e = lambda v: v[None, :, None, None]
        return e(coef[2 * c:]) * (dy - e(coef[:c]) - (x - e(mean)) * e(coef[c:2 * c]))

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
    sl = slice(0, k) if rank == 0 else slice(k, None)          # unequal shards: 4 and 2 images
    x = g["x"][sl].clone().requires_grad_(True)
    out = mod(x)
    out.backward(g["dy"][sl])                                  # second all-reduce of 2C+2 floats: sum dy, sum dy * (x - mean)
    np.save(os.path.join(tmp, f"out{rank}.npy"), out.detach().numpy())
    np.save(os.path.join(tmp, f"rm{rank}.npy"), mod.running_mean.numpy())
    np.save(os.path.join(tmp, f"rv{rank}.npy"), mod.running_var.numpy())
    np.save(os.path.join(tmp, f"dx{rank}.npy"), x.grad.numpy())
    np.save(os.path.join(tmp, f"dw{rank}.npy"), mod.weight.grad.numpy())
    np.save(os.path.join(tmp, f"db{rank}.npy"), mod.bias.grad.numpy())
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
        # backward: each rank's parameter gradients are its replica's, the input gradient needs the totals of both
        assert np.abs(np.load(tmp_path / f"dw{r}.npy") - g[f"sync_dw{r}"].numpy()).max() <= 2e-5
        assert np.abs(np.load(tmp_path / f"db{r}.npy") - g[f"sync_db{r}"].numpy()).max() <= 2e-5
    dx = np.concatenate([np.load(tmp_path / "dx0.npy"), np.load(tmp_path / "dx1.npy")])
    assert np.abs(dx - g["sync_dx"].numpy()).max() <= 2e-6


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
    assert float((m(x).detach().cpu() - g["eval_out"]).abs().max()) <= TOL_OUT
    assert torch.equal(m.running_mean.cpu(), g["running_mean"])                 # evaluation updates nothing
    m = _load(SynchronizedBatchNorm2d(x.shape[1]), g).to(DEV).train()
    out = m(x).detach()
    assert float((out.cpu() - g["single_out"]).abs().max()) <= TOL_OUT
    assert float((m.running_mean.cpu() - g["single_running_mean"]).abs().max()) <= TOL_STAT
    assert float((m.running_var.cpu() - g["single_running_var"]).abs().max()) <= TOL_STAT
    assert int(m.num_batches_tracked) == 0      # the reference's forward never touches it (batchnorm.py:46-82)
    # the replicas' formula on one rank: whole batch as one shard == the two-replica statistics of the fixture
    m = _load(SynchronizedBatchNorm2d(x.shape[1], sync=True), g).to(DEV).train()
    out = m(x).detach()
    assert float((out.cpu() - g["sync_out"]).abs().max()) <= TOL_OUT
    assert float((m.running_var.cpu() - g["sync_running_var"]).abs().max()) <= TOL_STAT


TOL_DX, TOL_DW = 1e-5, 1e-4    # max abs: dx is O(1); dweight / dbias are sums of ~1150 O(1) terms


@pytest.mark.gpu
def test_gpu_module_gradients_match_reference_fixture():
    """Round 3: backward through the HIP kernels (autograd.Function) against the reference modules' gradients."""
    from eamm_amd import SynchronizedBatchNorm2d
    g = fixture()
    dy = g["dy"].to(DEV)
    for key, kw, train in (("eval", {}, False), ("single", {}, True), ("sync", {"sync": True}, True)):
        m = _load(SynchronizedBatchNorm2d(g["x"].shape[1], **kw), g).to(DEV).train(train)
        x = g["x"].to(DEV).requires_grad_(True)
        out = m(x)
        assert out.requires_grad and out.grad_fn is not None           # the output used to be graph-less
        out.backward(dy)
        assert float((x.grad.cpu() - g[key + "_dx"]).abs().max()) <= TOL_DX, key
        # the replicas' formula on one rank sees the whole batch: its parameter gradients are the two replicas' added
        dw = g["sync_dw0"] + g["sync_dw1"] if key == "sync" else g[key + "_dw"]
        db = g["sync_db0"] + g["sync_db1"] if key == "sync" else g[key + "_db"]
        assert float((m.weight.grad.cpu() - dw).abs().max()) <= TOL_DW and float((m.bias.grad.cpu() - db).abs().max()) <= TOL_DW, key
    # gradient only with respect to the input (frozen parameters), and a non-contiguous upstream gradient
    m = _load(SynchronizedBatchNorm2d(g["x"].shape[1]), g).to(DEV).train()
    m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    x = g["x"].to(DEV).requires_grad_(True)
    m(x).backward(dy.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2))
    assert float((x.grad.cpu() - g["single_dx"]).abs().max()) <= TOL_DX and m.weight.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 256, 64, 64), (2, 64, 128, 128), (3, 5, 7, 9), (4, 33, 1, 1)])
def test_gpu_gradients_against_float64_autograd(shape):
    """Generator-sized and awkward shapes: dx / dweight / dbias against float64 autograd through F.batch_norm."""
    import torch.nn.functional as F
    from eamm_amd import SynchronizedBatchNorm2d
    rs = np.random.RandomState(sum(shape) + 1)
    c = shape[1]
    x = torch.from_numpy((rs.standard_normal(shape) * rs.uniform(0.5, 2, (1, c, 1, 1)) + rs.standard_normal((1, c, 1, 1))).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
    w = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
    m = SynchronizedBatchNorm2d(c).to(DEV).train()
    with torch.no_grad():
        m.weight.copy_(w)
    xg = x.to(DEV).requires_grad_(True)
    m(xg).backward(dy.to(DEV))
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), torch.zeros(c, dtype=torch.float64, requires_grad=True)
    F.batch_norm(xd, None, None, wd, bd, True, 0.1, 1e-5).backward(dy.double())
    n = x.numel() // c
    assert float((xg.grad.cpu().double() - xd.grad).abs().max()) <= 2e-5 * max(1.0, float(xd.grad.abs().max()))
    assert float((m.weight.grad.cpu().double() - wd.grad).abs().max()) <= 2e-6 * n ** 0.5 + 1e-5
    assert float((m.bias.grad.cpu().double() - bd.grad).abs().max()) <= 2e-6 * n ** 0.5 + 1e-5


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
    out = m(x.to(DEV)).detach().cpu()
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
    sl = slice(0, k) if rank == 0 else slice(k, None)
    x = g["x"][sl].to(DEV).requires_grad_(True)
    out = mod(x)
    out.backward(g["dy"][sl].to(DEV))
    np.save(os.path.join(tmp, f"out{rank}.npy"), out.detach().cpu().numpy())
    np.save(os.path.join(tmp, f"rv{rank}.npy"), mod.running_var.cpu().numpy())
    np.save(os.path.join(tmp, f"dx{rank}.npy"), x.grad.cpu().numpy())
    np.save(os.path.join(tmp, f"dw{rank}.npy"), mod.weight.grad.cpu().numpy())
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
        assert np.abs(np.load(tmp_path / f"dw{r}.npy") - g[f"sync_dw{r}"].numpy()).max() <= TOL_DW
    dx = np.concatenate([np.load(tmp_path / "dx0.npy"), np.load(tmp_path / "dx1.npy")])
    assert np.abs(dx - g["sync_dx"].numpy()).max() <= TOL_DX
