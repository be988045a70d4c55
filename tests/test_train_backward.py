"""loss.backward() through the generator in .train() mode (reference train.py:133).

Fixture tests/golden/tiny64_train_backward.npz (oracle/make_golden.py::train_backward_case): the REFERENCE generator in
.train(), a scalar loss weighing every output with fixed random tensors, the gradients of every parameter, of the key points
(value + jacobian, driving and source) and of the source image -- as the reference computes them in fp32, the same in double
(`grad64`), and their distance (`floor`): the algorithm's own fp32 noise floor, up to 9e-3 of a tensor's largest gradient on
this randomly-initialised network (batch-statistics BatchNorm backward cancels heavily), median 9e-4.

Round 4 adds ``full256_train_backward.npz`` (the shipped configuration, 2 pairs at 256x256: the size at which the Winograd weight
gradient, the thin 7x7 kernels on 64 channels and the split-group F(4x4) run inside the graph; strided samples) and
``tiny64_eval_backward.npz`` (the reference generator differentiated in .eval(): running statistics).

CPU: the oracle's training / evaluation branch differentiated by autograd against the fixtures.
GPU (-m gpu): eamm_amd.OcclusionAwareGenerator -- the composition of differentiable HIP operators in eamm_amd/train_graph.py --
against the double gradients, each tensor within a few of ITS OWN fp32 floors; ``Adam(gen.parameters())`` straight after
construction as train.py:136 builds it."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from eamm_amd import tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc

KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")
DEV = "cuda:0"
# Bar of an fp32 run against the double gradients, per tensor, relative to the tensor's largest gradient: 4 x the fixture's
# own fp32-vs-fp64 distance for that tensor, and at least this (one run's distance is a noisy estimate of the floor: another
# summation order lands elsewhere inside it -- the oracle in fp32 differs from the reference in fp32 by up to 1.5e-3).
FP32_MIN_REL = 5e-3


def fixture(name="tiny64_train_backward"):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def inputs(n, size=64):
    return synthetic_source(size, seed=1, batch=n), synthetic_keypoints(n, 10, seed=0), synthetic_keypoints(n, 10, seed=2)


def loss_weights(z, shapes=None):
    """The fixed random tensors that weigh every output in the fixture's scalar loss: stored in full for the 64x64 fixtures,
    regenerated from the recorded seed for the 256x256 one (torch.randn on the CPU generator, KEYS order, as make_golden.py)."""
    if "w/prediction" in z:
        return {k: torch.from_numpy(z["w/" + k]) for k in KEYS}
    g = torch.Generator().manual_seed(int(z["weights_seed"]))
    return {k: torch.randn(shapes[k], generator=g) for k in KEYS}


# Tensors in front of which (towards the loss) no ReLU sits for the final layer, and the inputs whose gradient is a sum over
# the whole frame (one flipped ReLU pixel is diluted in them): bar = 4 x the tensor's OWN fp32-vs-fp64 distance, no 5e-3 floor.
def _tight(name):
    return name.startswith(("kp_source/", "kp_driving/", "param/final.")) or name == "source_image"


TIGHT_MIN_REL = 2e-5      # (the operator-level bar of tests/test_gpu_backward.py: two fp32 summation orders differ by this much)


def sampled(t, step):
    return t.reshape(-1)[::step] if step > 1 else t


def compare(z, grads, floors_allowed, min_rel, report=False):
    """grads: {fixture name: tensor}.  Returns the worst ratio error / bar over all tensors (and asserts every one)."""
    names = [str(s) for s in z["names"]]
    assert sorted(grads) == sorted(names)
    gmax = max(float(z["scale/" + k]) for k in names)
    worst = (0.0, None)
    rows = []
    for k in names:
        want = torch.from_numpy(z["grad64/" + k]).double().reshape(-1)
        got = sampled(grads[k].detach().cpu().double(), int(z["step/" + k])).reshape(-1)
        scale = gmax if bool(z["zero/" + k]) else float(z["scale/" + k])
        floor_k = float(z["floor/" + k])
        rel_bar = max(floors_allowed * floor_k, min(min_rel, TIGHT_MIN_REL) if _tight(k) else min_rel)
        bar = rel_bar * scale
        diff = (got - want).abs()
        # A few elements may sit far outside the rounding floor: an activation within an ulp of zero takes the other branch of
        # its ReLU in a run that rounds differently, and that one pixel's whole upstream gradient enters (or leaves) every sum
        # it feeds -- a weight gradient's Cin x taps entries of one output channel at once.  (Seen: replacing a batched 2x2
        # matmul by its element-wise form moved 8 sampled entries of up_blocks.1.conv.weight by up to 0.8 % of the tensor's
        # largest gradient, everything else by ~1e-6.)  So (not "95 % within the bar" -- a systematic halo error of
        # a weight-gradient kernel could hide in the other 5 %): at most 0.5 % of a tensor's sampled elements (2 of a tensor of a few
        # hundred, ONE of a smaller one: a flip moves a bias / BatchNorm-parameter gradient too) beyond the bar, none beyond 20 bars.
        allowed = max(2, diff.numel() // 200) if diff.numel() >= 400 else 1
        beyond = int((diff > bar).sum())
        err = float(torch.quantile(diff, 1.0 - allowed / diff.numel())) if allowed else float(diff.max())
        rows.append((err / bar, k, err / scale, floor_k, beyond, diff.numel()))
        assert beyond <= allowed, (k, beyond, allowed, err, bar, floor_k)
        assert float(diff.max()) <= 20 * bar, (k, float(diff.max()), bar, floor_k)
        if err / bar > worst[0]:
            worst = (err / bar, k)
    if report:   # per-tensor error against its own floor
        rows.sort(reverse=True)
        for ratio, k, rel, fl, beyond, numel in rows[:12]:
            print(f"  {k:58s} err/scale {rel:.2e}  own floor {fl:.2e}  err/bar {ratio:.2f}  beyond bar {beyond}/{numel}")
    return worst


def test_oracle_autograd_matches_reference_gradients():
    z = fixture()
    cfg, n = tiny_config(), int(z["n"])
    sd = synthetic_state_dict(cfg, seed=int(z["weight_seed"]))
    src, kp_s, kp_d = inputs(n)
    for dtype, floors, min_rel in ((torch.float64, 0.0, 1e-6), (torch.float32, 4.0, FP32_MIN_REL)):
        leaf = lambda k, v: v.is_floating_point() and "running" not in k and "down.weight" not in k
        sdd = {k: (v.to(dtype).requires_grad_() if leaf(k, v) else v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        s = src.to(dtype).requires_grad_()
        ks = {k: v.to(dtype).requires_grad_() for k, v in kp_s.items()}
        kd = {k: v.to(dtype).requires_grad_() for k, v in kp_d.items()}
        out, _ = orc.generator_forward_train(sdd, cfg, s, kd, ks, parallel=False)
        loss = sum((out[k] * torch.from_numpy(z["w/" + k]).to(dtype)).sum() for k in KEYS)
        loss.backward()
        assert abs(float(loss.detach()) - float(z["loss64"])) <= (1e-9 if dtype == torch.float64 else 2e-4) * abs(float(z["loss64"]))
        grads = {"source_image": s.grad}
        grads.update({"kp_source/" + k: v.grad for k, v in ks.items()})
        grads.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
        grads.update({"param/" + k: v.grad for k, v in sdd.items() if v.requires_grad})
        compare(z, grads, floors, min_rel)


def test_oracle_autograd_matches_reference_gradients_in_eval_mode():
    """The reference module is differentiable in .eval() (running statistics): the oracle's evaluation forward, differentiated
    by autograd in double, against the reference's gradients (fixture tiny64_eval_backward)."""
    z = fixture("tiny64_eval_backward")
    cfg, n = tiny_config(), int(z["n"])
    sd = synthetic_state_dict(cfg, seed=int(z["weight_seed"]))
    src, kp_s, kp_d = inputs(n)
    leaf = lambda k, v: v.is_floating_point() and "running" not in k and "down.weight" not in k
    sdd = {k: (v.double().requires_grad_() if leaf(k, v) else v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    s = src.double().requires_grad_()
    ks = {k: v.double().requires_grad_() for k, v in kp_s.items()}
    kd = {k: v.double().requires_grad_() for k, v in kp_d.items()}
    out = orc.generator_forward(sdd, cfg, s, kd, ks)
    loss = sum((out[k] * torch.from_numpy(z["w/" + k]).double()).sum() for k in KEYS)
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss64"])) <= 1e-9 * abs(float(z["loss64"]))
    grads = {"source_image": s.grad}
    grads.update({"kp_source/" + k: v.grad for k, v in ks.items()})
    grads.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
    grads.update({"param/" + k: v.grad for k, v in sdd.items() if v.requires_grad})
    compare(z, grads, 0.0, 1e-6)


# ---------------------------------------------------------------------------------------------------------------------
def _make(cfg, seed):
    from eamm_amd import OcclusionAwareGenerator
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(synthetic_state_dict(cfg, seed=seed), strict=True)
    return gen.to(DEV)


def _run_graph(gen, z, n, size=64):
    src, kp_s, kp_d = inputs(n, size)
    s = src.to(DEV).requires_grad_()
    ks = {k: v.to(DEV).requires_grad_() for k, v in kp_s.items()}
    kd = {k: v.to(DEV).requires_grad_() for k, v in kp_d.items()}
    out = gen(s, kp_driving=kd, kp_source=ks)
    w = loss_weights(z, {k: out[k].shape for k in KEYS})
    loss = sum((out[k] * w[k].to(DEV)).sum() for k in KEYS)
    loss.backward()
    torch.cuda.synchronize()
    grads = {"source_image": s.grad}
    grads.update({"kp_source/" + k: v.grad for k, v in ks.items()})
    grads.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
    grads.update({"param/" + k: p.grad for k, p in gen.named_parameters()})
    return out, loss, grads


@pytest.mark.gpu
def test_generator_backward_matches_reference_gradients():
    z = fixture()
    n = int(z["n"])
    gen = _make(tiny_config(), int(z["weight_seed"])).train()  # parameters require grad by default, as the reference's
    before = {k: v.clone() for k, v in gen.state_dict().items() if "running_var" in k}
    out, loss, grads = _run_graph(gen, z, n)
    for k in KEYS:                                             # the forward of the graph path is the reference's
        err = float((out[k].detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max())
        assert err <= (1e-4 if k == "deformed" else 2e-5), (k, err)
    assert abs(float(loss.detach()) - float(z["loss64"])) <= 5e-4 * abs(float(z["loss64"]))
    assert all(g is not None for g in grads.values())
    worst = compare(z, grads, floors_allowed=4.0, min_rel=FP32_MIN_REL, report=True)
    print(f"generator backward: worst error / bar = {worst[0]:.2f} at {worst[1]}")
    # the running statistics moved, as in every training-mode forward
    after = gen.state_dict()
    assert max(float((after[k] - v).abs().max()) for k, v in before.items()) > 1e-3


@pytest.mark.gpu
def test_generator_backward_matches_reference_gradients_at_256():
    """the gradients at the size the path runs at -- the shipped configuration, 2 pairs at 256x256 (Winograd
    weight gradient on >= 512 tiles, thin 7x7 kernels on 64 channels, split-group F(4x4) inside the graph) -- against the
    REFERENCE's autograd in double (fixture: strided samples of all 71 gradient tensors + each tensor's own fp32 floor)."""
    from eamm_amd import hot_path_config
    z = fixture("full256_train_backward")
    n, size = int(z["n"]), int(z["size"])
    gen = _make(hot_path_config(), int(z["weight_seed"])).train()
    out, loss, grads = _run_graph(gen, z, n, size)
    for k in KEYS:   # forward of the graph path against the reference's (strided samples; bars of tests/test_train_mode.py at 256)
        got = out[k].detach().cpu().reshape(-1)[::int(z["ostep/" + k])]
        err = float((got - torch.from_numpy(z["out64/" + k])).abs().max())
        ref_floor = float((torch.from_numpy(z["out/" + k]) - torch.from_numpy(z["out64/" + k])).abs().max())
        print(f"256 train graph forward {k:16s} max|hip - ref64| = {err:.2e} (reference's own fp32 run: {ref_floor:.2e})")
        assert err <= {"prediction": 4e-4, "deformed": 1e-3, "sparse_deformed": 5e-4}.get(k, 1e-4), (k, err)
    assert abs(float(loss.detach()) - float(z["loss64"])) <= 5e-4 * abs(float(z["loss64"]))
    assert all(g is not None for g in grads.values())
    worst = compare(z, grads, floors_allowed=4.0, min_rel=FP32_MIN_REL, report=True)
    print(f"generator backward at 256x256: worst error / bar = {worst[0]:.2f} at {worst[1]}")


@pytest.mark.gpu
def test_eval_mode_backward_matches_reference_gradients_at_256():
    """The SHARP gradient check at the size the path runs at: the shipped configuration, 2 pairs at 256x256, in .eval() -- running
    statistics, so the batch statistics' cancellation (which puts the reference's own fp32 run 1-4 % away from its double run
    in .train() at two pairs) is absent and the decoder / bottleneck tensors' own floors are 1e-6 ... 1e-3: the Winograd
    weight-gradient form, the thin 7x7 kernels on 64 channels and the F(4x4) data gradient inside the real graph, against
    the REFERENCE's autograd in double."""
    from eamm_amd import hot_path_config
    z = fixture("full256_eval_backward")
    n, size = int(z["n"]), int(z["size"])
    gen = _make(hot_path_config(), int(z["weight_seed"])).eval()
    with pytest.warns(UserWarning, match="autograd graph in .eval"):
        out, loss, grads = _run_graph(gen, z, n, size)
    for k in KEYS:
        got = out[k].detach().cpu().reshape(-1)[::int(z["ostep/" + k])]
        err = float((got - torch.from_numpy(z["out64/" + k])).abs().max())
        print(f"256 eval graph forward {k:16s} max|hip - ref64| = {err:.2e}")
        assert err <= {"prediction": 1e-4, "deformed": 5e-4, "sparse_deformed": 5e-4}.get(k, 1e-5), (k, err)
    assert abs(float(loss.detach()) - float(z["loss64"])) <= 2e-4 * abs(float(z["loss64"]))
    worst = compare(z, grads, floors_allowed=4.0, min_rel=FP32_MIN_REL, report=True)
    print(f"eval-mode backward at 256x256: worst error / bar = {worst[0]:.2f} at {worst[1]}")


@pytest.mark.gpu
def test_adam_straight_after_construction_trains_every_parameter():
    """VERDICT r03 item 9(i) / reference train.py:136: ``torch.optim.Adam(generator.parameters())`` built straight after
    construction + load_state_dict, one ``loss.backward()`` + ``step()``: every parameter has a gradient and moves -- except
    the convolution biases in front of a batch-statistics BatchNorm, whose gradient is mathematically zero (the mean is
    subtracted again; the reference holds rounding noise there, this library returns exact zeros)."""
    z = fixture()
    n = int(z["n"])
    gen = _make(tiny_config(), int(z["weight_seed"])).train()
    opt = torch.optim.Adam(gen.parameters(), lr=1e-3)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(list(gen.parameters()))
    before = {k: p.detach().clone() for k, p in gen.named_parameters()}
    src, kp_s, kp_d = inputs(n)
    cu = lambda d: {k: v.to(DEV) for k, v in d.items()}
    out = gen(src.to(DEV), kp_driving=cu(kp_d), kp_source=cu(kp_s))
    assert out["prediction"].requires_grad and out["prediction"].grad_fn is not None
    (out["prediction"] - 0.5).abs().mean().backward()
    opt.step()
    zero_by_construction = {str(k)[len("param/"):] for k in z["names"] if bool(z["zero/" + str(k)])}
    moved, still = [], []
    for k, p in gen.named_parameters():
        assert p.grad is not None, k
        (moved if float((p.detach() - before[k]).abs().max()) > 0 else still).append(k)
    # outputs other than 'prediction' are not in this loss: the occlusion / mask heads still get gradients through the warp
    assert set(still) <= zero_by_construction, sorted(set(still) - zero_by_construction)
    assert len(moved) >= len(before) - len(zero_by_construction)


@pytest.mark.gpu
def test_eval_mode_forward_is_differentiable_like_the_reference():
    """.eval() with gradients enabled is differentiable (running statistics in every BatchNorm), never
    a silently detached tensor.  Values equal the folded inference engine's; gradients equal the REFERENCE's autograd in
    evaluation mode (fixture tiny64_eval_backward: every parameter, the key points, the source)."""
    z = fixture("tiny64_eval_backward")
    n = int(z["n"])
    gen = _make(tiny_config(), int(z["weight_seed"])).eval()
    stats = {k: v.clone() for k, v in gen.state_dict().items() if "running" in k}
    with pytest.warns(UserWarning, match="autograd graph in .eval"):
        out, loss, grads = _run_graph(gen, z, n)
    assert all(out[k].requires_grad for k in KEYS)
    assert all(torch.equal(v, gen.state_dict()[k]) for k, v in stats.items())       # evaluation mode: statistics untouched
    src, kp_s, kp_d = inputs(n)
    cu = lambda d: {k: v.to(DEV) for k, v in d.items()}
    with torch.no_grad():
        fast = gen(src.to(DEV), kp_driving=cu(kp_d), kp_source=cu(kp_s))             # the folded engine
    assert not fast["prediction"].requires_grad
    for k in KEYS:
        ref = float((out[k].detach().cpu() - torch.from_numpy(z["out/" + k])).abs().max())
        eng = float((out[k].detach() - fast[k]).abs().max())
        print(f"eval graph {k:16s} vs reference {ref:.2e}   vs engine {eng:.2e}")
        assert ref <= (1e-4 if k == "deformed" else 2e-5) and eng <= (2e-4 if k == "deformed" else 4e-5), (k, ref, eng)
    assert abs(float(loss.detach()) - float(z["loss64"])) <= 5e-4 * abs(float(z["loss64"]))
    worst = compare(z, grads, floors_allowed=4.0, min_rel=FP32_MIN_REL, report=True)
    print(f"eval-mode backward: worst error / bar = {worst[0]:.2f} at {worst[1]}")
    # a frozen generator in .eval() with an input that requires grad: the flow's backward alone, still a graph
    gen.requires_grad_(False)
    kd = {k: v.to(DEV).requires_grad_() for k, v in kp_d.items()}
    o2 = gen(src.to(DEV), kp_driving=kd, kp_source=cu(kp_s))
    assert o2["prediction"].requires_grad
    o2["prediction"].sum().backward()
    assert kd["value"].grad is not None and float(kd["value"].grad.abs().max()) > 0


@pytest.mark.gpu
def test_graph_and_engine_training_forwards_agree():
    # the same module, the same batch: the differentiable composition and the resumable engine are two routes through the
    # same kernels' arithmetic (different fusions): outputs and running statistics within the parity bars
    z = fixture()
    n = int(z["n"])
    src, kp_s, kp_d = inputs(n)
    cu = lambda d: {k: v.to(DEV) for k, v in d.items()}
    g1 = _make(tiny_config(), 1234).train()
    with torch.no_grad():
        o1 = g1(src.to(DEV), kp_driving=cu(kp_d), kp_source=cu(kp_s))
    g2 = _make(tiny_config(), 1234).train()
    o2 = g2(src.to(DEV), kp_driving=cu(kp_d), kp_source=cu(kp_s))
    assert o2["prediction"].requires_grad and not o1["prediction"].requires_grad
    for k in KEYS:
        assert float((o1[k] - o2[k].detach()).abs().max()) <= (1e-4 if k == "deformed" else 2e-5), k
    s1, s2 = g1.state_dict(), g2.state_dict()
    for k in s1:
        if "running" in k:
            assert float((s1[k] - s2[k]).abs().max()) <= 1e-5, k


@pytest.mark.gpu
def test_only_the_key_points_need_a_gradient():
    # the audio-to-key-point stage trains THROUGH a frozen generator (train.py: optimizer_audio_feature): parameters without
    # gradients, driving key points with -- the flow's backward alone
    z = fixture()
    n = int(z["n"])
    gen = _make(tiny_config(), 1234).train().requires_grad_(False)
    src, kp_s, kp_d = inputs(n)
    kd = {k: v.to(DEV).requires_grad_() for k, v in kp_d.items()}
    out = gen(src.to(DEV), kp_driving=kd, kp_source={k: v.to(DEV) for k, v in kp_s.items()})
    loss = sum((out[k] * torch.from_numpy(z["w/" + k]).to(DEV)).sum() for k in KEYS)
    loss.backward()
    assert all(p.grad is None for p in gen.parameters())
    for k in ("value", "jacobian"):
        name = "kp_driving/" + k
        want = torch.from_numpy(z["grad64/" + name]).double()
        diff = (kd[k].grad.cpu().double() - want).abs().reshape(-1)
        bar = max(4 * float(z["floor/" + name]), FP32_MIN_REL) * float(z["scale/" + name])
        assert float(diff.median()) <= bar and float(diff.max()) <= 20 * bar, (name, float(diff.max()), bar)


# ---------------------------------------------------------------------------------------------------------------------
# replicas and constructor variants: the oracle's training branch, differentiated in double, is the reference here
def _oracle_gradients(cfg, sd, src, kp_s, kp_d, weights, parallel, dtype=torch.float64):
    leaf = lambda k, v: v.is_floating_point() and "running" not in k and "down.weight" not in k
    sdd = {k: (v.to(dtype).requires_grad_() if leaf(k, v) else v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    ks = {k: v.to(dtype).requires_grad_() for k, v in kp_s.items()}
    kd = {k: v.to(dtype).requires_grad_() for k, v in kp_d.items()}
    out, _ = orc.generator_forward_train(sdd, cfg, src.to(dtype), kd, ks, parallel=parallel)
    sum((out[k] * weights[k].to(dtype)).sum() for k in weights).backward()
    grads = {"param/" + k: v.grad for k, v in sdd.items() if v.requires_grad}
    grads.update({"kp_source/" + k: v.grad for k, v in ks.items()})
    grads.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
    return grads


def _close(got, want, what, want32=None):
    """fp32 run against the double gradients: the reference's own fp32 run sits up to 9.3e-3 of a tensor's largest entry away
    from its double run on this network (fixture: up_blocks.0.conv.weight; 7.9e-3 down_blocks.1.norm.weight, ...), so the bar
    is 2e-2 -- 95 % of a tensor's entries within it, none beyond 10 x (a wrong kernel is off by the gradient's own size).
    With ``want32`` (the ORACLE's fp32 run of the same computation) a tensor's bar is at least 4 x its own fp32-vs-fp64
    distance: one- and few-element gradients that are heavily cancelling sums (the occlusion head's bias: one float summed over
    every pixel of the batch) sit several per cent away from the double value in any fp32 run."""
    gmax = max(float(v.abs().max()) for v in want.values())
    for k, w in want.items():
        diff = (got[k].detach().cpu().double() - w).abs().reshape(-1)
        scale = float(w.abs().max())
        scale = scale if scale >= 1e-9 * gmax else gmax
        rel = 2e-2
        if want32 is not None:
            rel = max(rel, 4.0 * float((want32[k].double() - w).abs().max()) / scale)
        if diff.numel() < 8:
            # a one-float gradient that is a cancelling sum over every pixel (dense_motion_network.occlusion.bias: -0.65 left of
            # terms two orders larger) moves by 1.3 % when the source image is scaled by (1 + 1e-7) and by 3.7 % between two
            # routes that agree to 1e-5 in evaluation mode (tools/micro/noise_probe.py, batch-statistics BatchNorm): 6e-2
            rel = max(rel, 6e-2)
        bar = rel * scale
        q = float(torch.quantile(diff, 0.95)) if diff.numel() >= 40 else float(diff.max())
        assert q <= bar and float(diff.max()) <= 10 * bar, (what, k, q, float(diff.max()), bar)


def _ddp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    z = fixture()
    n, split = int(z["n"]), 3
    sl = slice(0, split) if rank == 0 else slice(split, n)
    src, kp_s, kp_d = inputs(n)
    gen = _make(tiny_config(), int(z["weight_seed"])).train()      # replicas because world > 1 (sync_batchnorm None)
    ks = {k: v[sl].to(DEV).requires_grad_() for k, v in kp_s.items()}
    kd = {k: v[sl].to(DEV).requires_grad_() for k, v in kp_d.items()}
    out = gen(src[sl].to(DEV), kp_driving=kd, kp_source=ks)
    loss = sum((out[k] * torch.from_numpy(z["w/" + k])[sl].to(DEV)).sum() for k in KEYS)
    loss.backward()
    torch.cuda.synchronize()
    blob = {"param/" + k: p.grad.cpu().numpy() for k, p in gen.named_parameters()}
    blob.update({"kp_source/" + k: v.grad.cpu().numpy() for k, v in ks.items()})
    blob.update({"kp_driving/" + k: v.grad.cpu().numpy() for k, v in kd.items()})
    # ... and eamm_amd.all_reduce_gradients leaves the SUM of the replicas' parameter gradients on every rank (the reference's
    # ReduceAddCoalesced on the DataParallel master), in flat buckets
    from eamm_amd import all_reduce_gradients
    assert all_reduce_gradients(gen.parameters(), bucket_mb=1.0) >= 2
    torch.cuda.synchronize()
    blob.update({"sum/" + k: p.grad.cpu().numpy() for k, p in gen.named_parameters()})
    np.savez(os.path.join(tmp, f"grad{rank}.npz"), **blob)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_replicas_gradients_add_up_to_the_whole_batch(tmp_path):
    """Two processes sharing GPU 0 under gloo, shards of 3 and 1 pairs, loss.backward() on each: the BatchNorm statistics AND
    the backward's (sum dy, sum dy * xhat) are all-reduced per site, so the replicas' parameter gradients must ADD UP to --
    and their key-point gradients BE the slices of -- the gradient of the whole batch under the replicas' formula
    (what the reference's DataParallel master accumulates, sync_batchnorm/batchnorm.py:55-125)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = fixture()
    n, split = int(z["n"]), 3
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=int(z["weight_seed"]))
    src, kp_s, kp_d = inputs(n)
    want = _oracle_gradients(cfg, sd, src, kp_s, kp_d, {k: torch.from_numpy(z["w/" + k]) for k in KEYS}, parallel=True)
    r = [np.load(tmp_path / f"grad{i}.npz") for i in (0, 1)]
    got = {}
    for k in want:
        if k.startswith("param/"):
            got[k] = torch.from_numpy(r[0][k]) + torch.from_numpy(r[1][k])
        else:
            got[k] = torch.cat([torch.from_numpy(r[0][k]), torch.from_numpy(r[1][k])])
    _close(got, want, "two replicas")
    for k in want:                                                    # the bucketed all-reduce: exactly the sum, on both ranks
        if k.startswith("param/"):
            name = "sum/" + k[len("param/"):]
            for i in (0, 1):
                assert np.abs(r[i][name] - got[k].numpy()).max() <= 1e-6 * max(1.0, float(np.abs(got[k].numpy()).max())), (i, k)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["no_jacobian", "no_motion_network", "odd_widths", "gray", "kp5", "kp15"])
def test_generator_backward_variants_against_oracle_autograd(variant):
    """Branches of the differentiable forward: key points without jacobians (dense_motion.py:55) and a generator built
    without a motion network (generator.py:22-23)."""
    cfg = tiny_config()
    if variant == "no_motion_network":
        cfg = dict(cfg, dense_motion_params=None, estimate_occlusion_map=False)
    if variant == "odd_widths":     # widths that are not multiples of the kernels' 32-channel granule (48 / 96 / 192; 80 / 100)
        cfg = dict(cfg, block_expansion=48, max_features=200,
                   dense_motion_params=dict(cfg["dense_motion_params"], block_expansion=40, max_features=100))
    if variant == "gray":           # one image channel: the generic thin-layer / motion-stage routes of the operator composition
        cfg = dict(cfg, num_channels=1)
    if variant.startswith("kp"):    # round 6: num_kp != 10 through the differentiable motion operators (24 / 64 hourglass input channels)
        cfg = dict(cfg, num_kp=int(variant[2:]))
    n = 3
    sd = synthetic_state_dict(cfg, seed=77)
    src = synthetic_source(64, seed=3, batch=n, channels=cfg["num_channels"])
    kp_s = synthetic_keypoints(n, cfg["num_kp"], seed=4, jacobian=variant != "no_jacobian")
    kp_d = synthetic_keypoints(n, cfg["num_kp"], seed=5, jacobian=variant != "no_jacobian")
    gen = _make(cfg, 77).train()
    ks = {k: v.to(DEV).requires_grad_() for k, v in kp_s.items()}
    kd = {k: v.to(DEV).requires_grad_() for k, v in kp_d.items()}
    out = gen(src.to(DEV), kp_driving=kd, kp_source=ks)
    g = torch.Generator().manual_seed(9)
    weights = {k: torch.randn(out[k].shape, generator=g) for k in out}
    sum((out[k] * weights[k].to(DEV)).sum() for k in out).backward()
    want = _oracle_gradients(cfg, sd, src, kp_s, kp_d, weights, parallel=False)
    want32 = _oracle_gradients(cfg, sd, src, kp_s, kp_d, weights, parallel=False, dtype=torch.float32)
    got = {"param/" + k: p.grad for k, p in gen.named_parameters()}
    if variant == "no_motion_network":     # the key points are never looked at (generator.py:64)
        assert all(v.grad is None for v in list(ks.values()) + list(kd.values()))
        want = {k: v for k, v in want.items() if k.startswith("param/") and v is not None}
        want32 = {k: want32[k] for k in want}
    else:
        got.update({"kp_source/" + k: v.grad for k, v in ks.items()})
        got.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
    _close(got, want, variant, want32)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["jacobian", "no_jacobian"])
def test_hip_motion_operators_and_torch_composition_give_the_same_gradients(variant, monkeypatch):
    """Round 4: the dense-motion stage of the differentiable forward runs on HIP operators with HIP backward (motion_ops); the
    torch-ROCm composition it replaced (EAMM_MOTION_TORCH=1) is the same function -- outputs and every gradient of the two
    routes agree far inside the fixtures' bars (they differ by fp32 rounding and the odd flipped ReLU only)."""
    cfg = tiny_config()
    n = 3
    src = synthetic_source(64, seed=3, batch=n)
    jac = variant == "jacobian"
    kp_s, kp_d = synthetic_keypoints(n, 10, seed=4, jacobian=jac), synthetic_keypoints(n, 10, seed=5, jacobian=jac)
    runs = {}
    for route in ("hip", "torch"):
        monkeypatch.setenv("EAMM_MOTION_TORCH", "1" if route == "torch" else "0")
        gen = _make(cfg, 77).eval()          # running statistics: no batch-statistics cancellation in the comparison
        s = src.to(DEV).requires_grad_()
        ks = {k: v.to(DEV).requires_grad_() for k, v in kp_s.items()}
        kd = {k: v.to(DEV).requires_grad_() for k, v in kp_d.items()}
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = gen(s, kp_driving=kd, kp_source=ks)
        g = torch.Generator().manual_seed(9)
        weights = {k: torch.randn(out[k].shape, generator=g) for k in KEYS}
        sum((out[k] * weights[k].to(DEV)).sum() for k in KEYS).backward()
        grads = {"source_image": s.grad}
        grads.update({"kp_source/" + k: v.grad for k, v in ks.items()})
        grads.update({"kp_driving/" + k: v.grad for k, v in kd.items()})
        grads.update({"param/" + k: p.grad for k, p in gen.named_parameters()})
        runs[route] = ({k: out[k].detach() for k in KEYS}, grads)
    for k in KEYS:
        assert float((runs["hip"][0][k] - runs["torch"][0][k]).abs().max()) <= 2e-5, k
    worst = (0.0, None)
    for k, gt in runs["torch"][1].items():
        gh = runs["hip"][1][k]
        scale = float(gt.abs().max())
        diff = (gh - gt).abs().reshape(-1)
        rel = float(diff.max()) / max(scale, 1e-12)
        if rel > worst[0]:
            worst = (rel, k)
        # everything within 2e-3 of the tensor's largest entry; 99.5 % within 2e-4
        assert rel <= 2e-3, (k, rel)
        if diff.numel() >= 400:
            assert float(torch.quantile(diff[:1 << 20], 0.995)) <= 2e-4 * scale, k
    print(f"hip vs torch motion stage ({variant}): worst relative gradient difference {worst[0]:.2e} at {worst[1]}")
