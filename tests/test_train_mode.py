"""The generator's forward in .train() mode (second slice): batch statistics in every BatchNorm,
running statistics updated, replicas' statistics all-reduced -- reference sync_batchnorm/batchnorm.py:55-125 inside the blocks
of modules/util.py:858-938 (the fine-tuning loop, train.py:133).

Fixture tests/golden/tiny64_train.npz (oracle/make_golden.py::train_mode_case): the REFERENCE generator in .train() on one
replica and on two replicas (3 + 1 pairs) driven through its own replication callbacks and SyncMaster protocol.
CPU: the oracle's training branch against it.  GPU (-m gpu): eamm_amd.OcclusionAwareGenerator.train() through the C ABI
(eamm_set_training / eamm_train_begin / eamm_train_next) against it -- outputs and the running statistics left in the
module's buffers -- on one rank, with the replicas' formula, and as two gloo ranks sharing the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, TOL
from eamm_amd import tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc

KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")
TOL_STAT = 1e-5          # running statistics (values are O(0.1..1)); the reference's own fp32 floor is ~1e-7
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _called_as_demo_py_calls_it():
    """Inference tests run under torch.no_grad(), as the reference's caller does (demo.py:195): with gradients enabled the
    module -- like the reference's -- would build an autograd graph through the differentiable operators instead."""
    with torch.no_grad():
        yield



def fixture():
    z = np.load(os.path.join(GOLDEN, "tiny64_train.npz"))
    return z, [str(s) for s in z["norm_names"]], int(z["n"]), int(z["split"])


def inputs(n):
    return synthetic_source(64, seed=1, batch=n), synthetic_keypoints(n, 10, seed=0), synthetic_keypoints(n, 10, seed=2)


def test_oracle_training_branch_matches_reference_fixture():
    z, names, n, _ = fixture()
    cfg, sd = tiny_config(), synthetic_state_dict(tiny_config(), seed=int(z["weight_seed"]))
    src, kp_s, kp_d = inputs(n)
    with torch.no_grad():
        for tag, par in (("single", False), ("sync", True)):
            out, stats = orc.generator_forward_train(sd, cfg, src, kp_d, kp_s, parallel=par)
            assert sorted(stats) == sorted(names) and len(names) == 15
            for k in KEYS:
                assert float((out[k] - torch.from_numpy(z[f"{tag}_{k}"])).abs().max()) <= TOL[k] / 2, (tag, k)
            for p in names:
                assert float((stats[p][0] - torch.from_numpy(z[f"{tag}_rm/{p}"])).abs().max()) <= 1e-6
                assert float((stats[p][1] - torch.from_numpy(z[f"{tag}_rv/{p}"])).abs().max()) <= 1e-6
            # the statistics really are the batch's: the running statistics moved
            assert max(float((stats[p][1] - sd[p + ".running_var"]).abs().max()) for p in names) > 1e-3


def test_site_names_follow_execution_order():
    """The library's BatchNorm sites are named by state_dict prefix; the module looks its BatchNorm2d holders up by them."""
    from eamm_amd import OcclusionAwareGenerator
    gen = OcclusionAwareGenerator(**tiny_config())
    names = sorted(n for n, m in gen.named_modules() if isinstance(m, torch.nn.BatchNorm2d))
    _, fix_names, _, _ = fixture()
    assert names == sorted(fix_names)
    gen.train()
    with pytest.raises(RuntimeError):
        gen.encode_source(torch.zeros(1, 3, 64, 64))       # the clip interface is the inference path
    with pytest.raises(RuntimeError):
        gen(torch.zeros(2, 3, 64, 64), {}, {})              # CPU module: no fallback in training mode either


# ---------------------------------------------------------------------------------------------------------------------
def _make(cfg, seed):
    from eamm_amd import OcclusionAwareGenerator
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(synthetic_state_dict(cfg, seed=seed), strict=True)
    return gen.to(DEV)


def _cuda(d):
    return {k: v.to(DEV) for k, v in d.items()}


def _check(out, gen, z, names, tag, sl=slice(None)):
    for k in KEYS:
        err = float((out[k].cpu() - torch.from_numpy(z[f"{tag}_{k}"])[sl]).abs().max())
        print(f"train {tag:6s} {k:16s} max|hip - reference| = {err:.2e} (tol {TOL[k]:g})")
        assert err <= TOL[k], (tag, k, err)
    sd = gen.state_dict()
    for p in names:
        assert float((sd[p + ".running_mean"].cpu() - torch.from_numpy(z[f"{tag}_rm/{p}"])).abs().max()) <= TOL_STAT, p
        assert float((sd[p + ".running_var"].cpu() - torch.from_numpy(z[f"{tag}_rv/{p}"])).abs().max()) <= TOL_STAT, p


@pytest.mark.gpu
def test_gpu_train_forward_matches_reference_fixture():
    z, names, n, _ = fixture()
    cfg = tiny_config()
    src, kp_s, kp_d = inputs(n)
    gen = _make(cfg, int(z["weight_seed"])).train()
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    assert set(out) == set(KEYS) and not out["prediction"].requires_grad
    _check(out, gen, z, names, "single")
    # .eval() afterwards folds the UPDATED running statistics (the version counters of the buffers were bumped)
    sd_new = {k: v.cpu() for k, v in gen.state_dict().items()}
    gen.eval()
    ev = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    with torch.no_grad():
        ref = orc.generator_forward(sd_new, cfg, src, kp_d, kp_s)
    assert float((ev["prediction"].cpu() - ref["prediction"]).abs().max()) <= TOL["prediction"]
    # the replicas' formula on one rank sees the whole batch = what the reference's two replicas compute together
    gen = _make(cfg, int(z["weight_seed"])).train()
    gen.sync_batchnorm = True
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    _check(out, gen, z, names, "sync")
    # a second call moves the running statistics again (momentum 0.1) and reuses the packed weights
    rv = gen.first.norm.running_var.clone()
    gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    assert float((gen.first.norm.running_var - rv).abs().max()) > 1e-4


@pytest.mark.gpu
def test_gpu_train_forward_full_size_against_oracle():
    """The shipped configuration at 256x256, two pairs: every output key and three sites' running statistics vs the oracle."""
    from eamm_amd import hot_path_config
    cfg = hot_path_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    src, kp_s, kp_d = synthetic_source(256, seed=1, batch=2), synthetic_keypoints(2, 10, seed=0), synthetic_keypoints(2, 10, seed=2)
    gen = _make(cfg, 1234).train()
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    with torch.no_grad():
        ref, stats = orc.generator_forward_train(sd, cfg, src, kp_d, kp_s)
    # Batch statistics over two pairs (eight values per channel at the deepest hourglass level) amplify fp32 rounding: the
    # REFERENCE's own fp32-vs-fp64 difference in this mode at this size is 7.9e-5 on 'prediction' and 1.6e-4 on 'deformed'
    # (measured with the oracle in float64; evaluation mode: 3e-6 / 4e-5), so two fp32 implementations may differ by twice that.
    tol = dict(TOL, prediction=4e-4, deformed=1e-3)
    for k in KEYS:
        err = float((out[k].cpu() - ref[k]).abs().max())
        print(f"train 256 {k:16s} max|hip - oracle| = {err:.2e} (tol {tol[k]:g})")
        assert err <= tol[k], (k, err)
    for p in ("first.norm", "bottleneck.r5.norm2", "dense_motion_network.hourglass.encoder.down_blocks.4.norm"):
        assert float((gen.state_dict()[p + ".running_var"].cpu() - stats[p][1]).abs().max()) <= TOL_STAT, p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)     # the graph-free resumable pass (the differentiable one: test_train_backward.py)
    z, names, n, split = fixture()
    sl = slice(0, split) if rank == 0 else slice(split, n)
    src, kp_s, kp_d = inputs(n)
    gen = _make(tiny_config(), int(z["weight_seed"])).train()          # sync_batchnorm None: replicas because world > 1
    out = gen(src[sl].to(DEV), kp_source=_cuda({k: v[sl] for k, v in kp_s.items()}), kp_driving=_cuda({k: v[sl] for k, v in kp_d.items()}))
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **{k: out[k].cpu().numpy() for k in KEYS},
             **{"rv/" + p: gen.state_dict()[p + ".running_var"].cpu().numpy() for p in names})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_two_ranks_unequal_shards_match_the_reference_replicas(tmp_path):
    """Two processes sharing GPU 0 under gloo, shards of 3 and 1 pairs: 15 all-reduces of 2C+2 floats; outputs and running
    statistics must equal the reference's two-replica run on BOTH ranks (the reference updates its master only)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z, names, n, split = fixture()
    r = [np.load(tmp_path / f"rank{i}.npz") for i in (0, 1)]
    for k in KEYS:
        got = np.concatenate([r[0][k], r[1][k]])
        assert np.abs(got - z["sync_" + k]).max() <= TOL[k], k
    for i in (0, 1):
        for p in names:
            assert np.abs(r[i]["rv/" + p] - z["sync_rv/" + p]).max() <= TOL_STAT, (i, p)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["no_jacobian", "no_motion_network", "one_pair_more_than_max_frames", "kp5", "kp30"])
def test_gpu_train_forward_variants_against_oracle(variant):
    """Branches of the forward in training mode against the oracle's training branch: key points without jacobians
    (dense_motion.py:55), a generator built without a motion network (generator.py:22-23), and a batch larger than the
    module's max_frames (the training engine is re-created for it)."""
    cfg = tiny_config()
    n = 3
    if variant == "no_motion_network":
        cfg = dict(cfg, dense_motion_params=None, estimate_occlusion_map=False)
    if variant == "one_pair_more_than_max_frames":
        n = 5
    if variant.startswith("kp"):    # round 6: num_kp != 10 in training mode (the resumable batch-statistics pass)
        cfg = dict(cfg, num_kp=int(variant[2:]))
    sd = synthetic_state_dict(cfg, seed=77)
    src = synthetic_source(64, seed=3, batch=n)
    kp_s = synthetic_keypoints(n, cfg["num_kp"], seed=4, jacobian=variant != "no_jacobian")
    kp_d = synthetic_keypoints(n, cfg["num_kp"], seed=5, jacobian=variant != "no_jacobian")
    from eamm_amd import OcclusionAwareGenerator
    gen = OcclusionAwareGenerator(**cfg, max_frames=4)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV).train()
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    with torch.no_grad():
        ref, stats = orc.generator_forward_train(sd, cfg, src, kp_d, kp_s)
    keys = ("prediction",) if variant == "no_motion_network" else ("prediction", "mask", "sparse_deformed", "deformed")
    assert set(out) >= set(keys)
    for k in keys:
        assert float((out[k].cpu() - ref[k]).abs().max()) <= TOL[k], (variant, k)
    new = gen.state_dict()
    for p in stats:
        assert float((new[p + ".running_var"].cpu() - stats[p][1]).abs().max()) <= TOL_STAT, (variant, p)


TRAIN_VARIANTS = {
    # name: (config overrides, dense_motion overrides, H, W) -- the constructor variants of tests/test_gpu_generator.py
    "rect_64x96": ({}, {}, 64, 96),
    "scale_half": ({}, {"scale_factor": 0.5}, 64, 64),
    "no_occlusion": ({"estimate_occlusion_map": False}, {}, 64, 64),
    "three_down": ({"num_down_blocks": 3, "max_features": 256}, {}, 64, 64),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TRAIN_VARIANTS))
def test_gpu_train_forward_constructor_variants(name):
    over, dm_over, H, W = TRAIN_VARIANTS[name]
    cfg = tiny_config()
    cfg.update(over)
    cfg["dense_motion_params"] = {**cfg["dense_motion_params"], **dm_over}
    sd = synthetic_state_dict(cfg, seed=4321)
    from eamm_amd import OcclusionAwareGenerator
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV).train()
    n = 3
    rs = np.random.RandomState(11)
    src = torch.from_numpy(rs.uniform(0, 1, (n, 3, H, W)).astype(np.float32))
    kp_s, kp_d = synthetic_keypoints(n, 10, seed=0), synthetic_keypoints(n, 10, seed=2)
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    with torch.no_grad():
        ref, stats = orc.generator_forward_train(sd, cfg, src, kp_d, kp_s)
    for k in out:
        assert out[k].shape == ref[k].shape
        assert float((out[k].cpu() - ref[k]).abs().max()) <= 2 * TOL[k], (name, k)     # small batches: twice the evaluation bars
    new = gen.state_dict()
    for p in stats:
        assert float((new[p + ".running_mean"].cpu() - stats[p][0]).abs().max()) <= TOL_STAT, (name, p)


@pytest.mark.gpu
def test_gpu_train_forward_operator_route_matches_reference_fixture():
    """``train_route = "operators"``: the graph-free .train() forward on the differentiable operators under no_grad (the faster
    of the two routes) against the same reference fixture as the engine's resumable pass."""
    z, names, n, _ = fixture()
    src, kp_s, kp_d = inputs(n)
    gen = _make(tiny_config(), int(z["weight_seed"])).train()
    gen.train_route = "operators"
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    assert gen._train_engine is None and not out["prediction"].requires_grad
    _check(out, gen, z, names, "single")


@pytest.mark.gpu
def test_gpu_train_forward_while_the_weights_keep_changing():
    """graph-free .train() forwards between optimiser steps.  The first weight change rebuilds the training engine;
    from the second on the module takes the operator composition under no_grad (no engine rebuild per step) -- same outputs
    as the oracle's training branch on the CURRENT weights each time, running statistics moving."""
    cfg = tiny_config()
    src, kp_s, kp_d = inputs(3)
    gen = _make(cfg, 1234).train()
    routes = []
    for step in range(4):
        if step:
            with torch.no_grad():                      # an optimiser step: in-place updates bump the version counters
                gen.bottleneck.r0.conv1.weight.mul_(1.01)
                gen.up_blocks[0].conv.bias.add_(0.01)
        sd_now = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}
        out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
        assert not out["prediction"].requires_grad
        routes.append("engine" if gen._train_engine is not None else "graph")
        ref, _ = orc.generator_forward_train(sd_now, cfg, src, kp_d, kp_s, parallel=False)
        for k in KEYS:
            assert float((out[k].cpu() - ref[k]).abs().max()) <= (2e-4 if k == "deformed" else 4e-5), (step, k)
    assert routes == ["engine", "engine", "graph", "graph"], routes


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["odd_widths", "gray"])
def test_gpu_train_forward_with_widths_that_are_not_multiples_of_32(variant):
    """The training-mode ENGINE needs the kernels' 32-channel granule (and three image channels); a generator with other widths
    takes the operator composition for its graph-free .train() forward (train_route "operators") -- against the oracle's
    training branch."""
    if variant == "gray":
        cfg = dict(tiny_config(), num_channels=1)
    else:
        cfg = dict(tiny_config(), block_expansion=48, max_features=200)
        cfg["dense_motion_params"] = dict(cfg["dense_motion_params"], block_expansion=40, max_features=100)
    src, kp_s, kp_d = inputs(3)
    src = src[:, :cfg["num_channels"]].contiguous()
    gen = _make(cfg, 1234).train()
    assert gen.train_route == "operators"
    sd_now = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}
    out = gen(src.to(DEV), kp_source=_cuda(kp_s), kp_driving=_cuda(kp_d))
    ref, _ = orc.generator_forward_train(sd_now, cfg, src, kp_d, kp_s, parallel=False)
    for k in KEYS:
        assert float((out[k].cpu() - ref[k]).abs().max()) <= (2e-4 if k == "deformed" else 4e-5), k
