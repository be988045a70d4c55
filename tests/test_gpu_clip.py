"""Clip pipeline on the GPU (BASELINE.json configs[3] and the reference's loop 2, demo.py:251-281): the real
EngineBackend through eamm_amd.animate_clip -- a 2048-frame clip on one GPU, two ranks sharing the GPU under a gloo
group (broadcast of the source cache, contiguous shards, gather), the normalize_kp / emotion-offset steps in front of
the generator -- against the CPU oracle and the reference-derived fixtures."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, TOL
from eamm_amd import EngineBackend, OcclusionAwareGenerator, animate_clip, hot_path_config, shard_bounds, tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _called_as_demo_py_calls_it():
    """Inference tests run under torch.no_grad(), as the reference's caller does (demo.py:195): with gradients enabled the
    module -- like the reference's -- would build an autograd graph through the differentiable operators instead."""
    with torch.no_grad():
        yield

DEV = "cuda:0"


def make_generator(cfg, seed=1234, **kw):
    gen = OcclusionAwareGenerator(**cfg, **kw)
    gen.load_state_dict(synthetic_state_dict(cfg, seed=seed), strict=True)
    return gen.to(DEV).eval()


def test_clip_2048_frames_single_gpu():
    """BASELINE configs[3] workload on one GPU: 2048 frames at 256x256, 16 per launch, spot frames against the oracle,
    uint8 packing of the same clip, and the shard a rank of an 8-GPU job would compute == the same frames of the whole."""
    cfg = hot_path_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = make_generator(cfg)
    T = 2048
    src, kp_s, kp_d = synthetic_source(256, seed=1), synthetic_keypoints(1, 10, seed=0), synthetic_keypoints(T, 10, seed=2)
    timings = {}
    frames, span = animate_clip(EngineBackend(gen, batch=16), src.to(DEV), kp_s, kp_d, 256, 256, timings=timings)
    assert span == (0, T) and frames.shape == (T, 3, 256, 256) and frames.device.type == "cuda"
    assert torch.isfinite(frames).all() and float(frames.min()) > 0 and float(frames.max()) < 1
    assert {"encode_ms", "compute_ms"} <= set(timings) and timings["compute_ms"] > 0
    print(f"\n2048-frame clip: {T / ((timings['encode_ms'] + timings['compute_ms']) * 1e-3):.0f} frames/s "
          f"(encode {timings['encode_ms']:.2f} ms, compute {timings['compute_ms']:.1f} ms)")
    for t in (0, 1000, 2047):
        with torch.no_grad():
            ref = orc.generator_forward(sd, cfg, src, {k: v[t:t + 1] for k, v in kp_d.items()}, kp_s)["prediction"]
        assert float((frames[t].cpu() - ref[0]).abs().max()) <= TOL["prediction"], t
    # what rank 5 of 8 would compute from its own shard of the key points
    a, b = shard_bounds(T, 8, 5)
    assert (a, b) == (1280, 1536)
    shard, _ = animate_clip(EngineBackend(gen, batch=16), src.to(DEV), kp_s, {k: v[a:b] for k, v in kp_d.items()}, 256, 256)
    assert torch.equal(shard, frames[a:b])
    u8, _ = animate_clip(EngineBackend(gen, batch=16), src.to(DEV), kp_s, {k: v[:64] for k, v in kp_d.items()}, 256, 256,
                         uint8=True)
    want = torch.clamp(torch.round(frames[:64] * 255), 0, 255).permute(0, 2, 3, 1)
    assert u8.dtype == torch.uint8 and float((u8.float() - want).abs().max()) <= 1


def test_animate_clip_relative_adapt_matches_oracle():
    """animate_clip(relative=True, adapt_movement_scale=True, kp_driving_initial=...) -- the reference loop's
    normalize_kp call (demo.py:276) in front of the generator -- against the oracle fed the key points the REFERENCE's
    normalize_kp produced (fixture normalize_kp.npz), inputs deliberately spread over host and device."""
    z = np.load(os.path.join(GOLDEN, "normalize_kp.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = make_generator(cfg)
    src = synthetic_source(64, seed=1)
    kp_s = {"value": g["kp_source_value"], "jacobian": g["kp_source_jacobian"]}
    kp_i = {"value": g["kp_initial_value"].to(DEV), "jacobian": g["kp_initial_jacobian"].to(DEV)}   # on the device
    kp_d = {"value": g["kp_driving_value"], "jacobian": g["kp_driving_jacobian"]}                   # on the host
    frames, span = animate_clip(EngineBackend(gen, batch=2), src, kp_s, kp_d, 64, 64, kp_driving_initial=kp_i,
                                relative=True, adapt_movement_scale=True)
    assert span == (0, 5)
    kp_ref = {"value": g["value_a1r1j1"], "jacobian": g["jacobian_a1r1j1"]}
    ref = np.stack(orc.animate_clip(sd, cfg, src, kp_s, kp_ref))
    err = np.abs(frames.cpu().numpy().transpose(0, 2, 3, 1) - ref).max()
    assert err <= TOL["prediction"], err
    plain, _ = animate_clip(EngineBackend(gen, batch=2), src, kp_s, kp_d, 64, 64)
    assert float((plain - frames).abs().max()) > 1e-3          # the normalisation really changed the key points


@pytest.mark.parametrize("k", [5, 15, 30])
def test_clip_interface_with_other_key_point_counts(k):
    """Round 6: num_kp != 10 through the clip interface -- encode once, batches of ragged size (two whole-pass chains from 10 frames),
    relative + adapt_movement_scale normalisation (normalize_kp is K-agnostic: the convex hull of K points), uint8 packing -- against the
    oracle generator fed the oracle's own normalize_kp."""
    cfg = {**tiny_config(), "num_kp": k}
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = make_generator(cfg)
    src = synthetic_source(64, seed=1)
    kp_s, kp_d = synthetic_keypoints(1, k, seed=0), synthetic_keypoints(23, k, seed=2)
    frames, span = animate_clip(EngineBackend(gen, batch=12), src, kp_s, kp_d, 64, 64)
    assert span == (0, 23)
    ref = np.stack(orc.animate_clip(sd, cfg, src, kp_s, kp_d))
    assert np.abs(frames.cpu().numpy().transpose(0, 2, 3, 1) - ref).max() <= TOL["prediction"]
    if k >= 3:   # adapt_movement_scale takes the convex hull's area: needs three points
        kp_i = {kk: v[:1] for kk, v in kp_d.items()}
        rel, _ = animate_clip(EngineBackend(gen, batch=12), src, kp_s, kp_d, 64, 64, kp_driving_initial=kp_i, relative=True,
                              adapt_movement_scale=True)
        per_frame = [orc.normalize_kp(kp_s, {kk: v[t:t + 1] for kk, v in kp_d.items()}, kp_i, adapt_movement_scale=True,
                                      use_relative_movement=True, use_relative_jacobian=True) for t in range(23)]      # demo.py:276, frame by frame
        norm = {kk: torch.cat([f[kk] for f in per_frame]) for kk in ("value", "jacobian")}
        ref_rel = np.stack(orc.animate_clip(sd, cfg, src, kp_s, norm))
        assert np.abs(rel.cpu().numpy().transpose(0, 2, 3, 1) - ref_rel).max() <= TOL["prediction"]


def test_animate_clip_emotion_offsets_match_reference_loop():
    """`--add_emo` path of loop 2 (demo.py:263-276): emotion offsets, then normalize_kp, then the generator; expected
    frames = oracle on the key points the reference's own statements produced (fixture emotion_offsets.npz)."""
    z = np.load(os.path.join(GOLDEN, "emotion_offsets.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = make_generator(cfg)
    src = synthetic_source(64, seed=1)
    kp_s = {"value": g["kp_source_value"], "jacobian": g["kp_source_jacobian"]}
    kp_i = {"value": g["kp_initial_value"], "jacobian": g["kp_initial_jacobian"]}
    kp_d = {"value": g["kp_driving_value"], "jacobian": g["kp_driving_jacobian"]}
    emo = {"value": g["emo_value"].to(DEV), "jacobian": g["emo_jacobian"].to(DEV)}
    frames, _ = animate_clip(EngineBackend(gen, batch=4), src, kp_s, kp_d, 64, 64, kp_driving_initial=kp_i, relative=True,
                             adapt_movement_scale=True, emo_driving=emo)
    ref = np.stack(orc.animate_clip(sd, cfg, src, kp_s, {"value": g["normalized_value"], "jacobian": g["normalized_jacobian"]}))
    err = np.abs(frames.cpu().numpy().transpose(0, 2, 3, 1) - ref).max()
    assert err <= TOL["prediction"], err
    only_offsets, _ = animate_clip(EngineBackend(gen, batch=4), src, kp_s, kp_d, 64, 64, emo_driving=emo)
    ref2 = np.stack(orc.animate_clip(sd, cfg, src, kp_s, {"value": g["offset_value"], "jacobian": g["offset_jacobian"]}))
    assert np.abs(only_offsets.cpu().numpy().transpose(0, 2, 3, 1) - ref2).max() <= TOL["prediction"]


def test_module_forward_survives_clip_use_of_the_same_engine():
    """gen(src_A) -> animate_clip(backend over the same generator, src_B) -> gen(src_A): the clip pipeline replaces the
    engine's source cache behind forward()'s back; forward() must notice (cache generation) and re-encode src_A."""
    cfg = tiny_config()
    gen = make_generator(cfg)
    kp_s, kp_d = synthetic_keypoints(1, 10, seed=0), synthetic_keypoints(1, 10, seed=2)
    ks, kd = {k: v.to(DEV) for k, v in kp_s.items()}, {k: v.to(DEV) for k, v in kp_d.items()}
    src_a, src_b = synthetic_source(64, seed=1).to(DEV), synthetic_source(64, seed=7).to(DEV)
    first = gen(src_a, kp_source=ks, kp_driving=kd)["prediction"].clone()
    clip_b, _ = animate_clip(EngineBackend(gen, batch=4), src_b, kp_s, kp_d, 64, 64)
    again = gen(src_a, kp_source=ks, kp_driving=kd)["prediction"]
    assert torch.equal(first, again)
    assert float((clip_b[0] - first[0]).abs().max()) > 1e-3
    eng = gen.engine
    eng.import_source_cache(eng.export_source_cache(1), 1)      # a direct import invalidates as well
    gen_b = gen(src_b, kp_source=ks, kp_driving=kd)["prediction"]
    assert float((gen_b[0] - clip_b[0]).abs().max()) <= 1e-6


def test_host_delivery_of_one_backend_never_overwrites_a_clip_the_caller_still_holds():
    """`to_host=True` hands out pinned buffers from the backend's pool.  A delivered clip must stay intact for as long as
    the caller keeps it (two clips in a row on ONE backend used to share storage, the second silently overwriting the first), and a
    buffer the caller has dropped must be reused instead of pinned again."""
    cfg = tiny_config()
    gen = make_generator(cfg)
    be = EngineBackend(gen, batch=4)
    src = synthetic_source(64, seed=1).to(DEV)
    kp_s = synthetic_keypoints(1, 10, seed=0)
    kp_a, kp_b = synthetic_keypoints(6, 10, seed=2), synthetic_keypoints(6, 10, seed=30)
    first, _ = animate_clip(be, src, kp_s, kp_a, 64, 64, to_host=True)
    assert first.is_pinned()
    keep = first.clone()
    as_numpy = first.numpy()                       # a numpy view keeps the buffer out of circulation as well
    second, _ = animate_clip(be, src, kp_s, kp_b, 64, 64, to_host=True)
    assert second.data_ptr() != first.data_ptr()
    assert torch.equal(first, keep) and not torch.equal(first, second)
    ptr_first = first.data_ptr()
    del first
    third, _ = animate_clip(be, src, kp_s, kp_b, 64, 64, to_host=True)       # the numpy view still holds the first buffer
    assert third.data_ptr() not in (ptr_first, second.data_ptr()) and np.array_equal(as_numpy, keep.numpy())
    assert torch.equal(third, second)
    del as_numpy, third
    fourth, _ = animate_clip(be, src, kp_s, kp_a, 64, 64, to_host=True)      # dropped buffers come back: no new pinning
    assert fourth.data_ptr() in (ptr_first, ) or len(be._host[(torch.float32, (3, 64, 64))]) <= 3
    assert torch.equal(fourth, keep)


def _rank_worker(rank, world, port, total, batch, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)                                   # both ranks share the one GPU of the box
    cfg = tiny_config()
    gen = make_generator(cfg)
    be = EngineBackend(gen, batch=batch)
    if rank == 0:
        src, kp_s, kp_d = synthetic_source(64, seed=1), synthetic_keypoints(1, 10, seed=0), synthetic_keypoints(total, 10, seed=2)
    else:
        src = kp_s = kp_d = None                                # only rank 0 holds the clip's inputs
    local, (a, b) = animate_clip(be, src, kp_s, kp_d, 64, 64)
    assert (a, b) == shard_bounds(total, world, rank) and local.shape == (b - a, 3, 64, 64) and local.is_cuda
    np.save(os.path.join(tmp, f"shard{rank}.npy"), local.cpu().numpy())
    full, span = animate_clip(be, src, kp_s, kp_d, 64, 64, gather=True)
    u8, _ = animate_clip(be, src, kp_s, kp_d, 64, 64, uint8=True, gather=True)
    if rank == 0:
        assert span == (0, total) and full.shape == (total, 3, 64, 64) and u8.shape == (total, 64, 64, 3)
        np.save(os.path.join(tmp, "gathered.npy"), full.cpu().numpy())
        np.save(os.path.join(tmp, "gathered_u8.npy"), u8.cpu().numpy())
    else:
        assert full.shape[0] == 0 and u8.shape[0] == 0
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_engine_backend_gather(tmp_path):
    """Two processes, one gloo group, both on GPU 0, the REAL EngineBackend: rank 0 encodes, the exported source cache
    is broadcast and imported on rank 1, each rank computes its contiguous shard in the HIP library, frames are
    gathered on rank 0 -- must equal the single-process clip bit for bit (same launch geometry per batch)."""
    total, batch, world = 11, 3, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_worker, args=(world, port, total, batch, str(tmp_path)), nprocs=world, join=True)
    cfg = tiny_config()
    gen = make_generator(cfg)
    src, kp_s, kp_d = synthetic_source(64, seed=1), synthetic_keypoints(1, 10, seed=0), synthetic_keypoints(total, 10, seed=2)
    ref, span = animate_clip(EngineBackend(gen, batch=batch), src, kp_s, kp_d, 64, 64)
    assert span == (0, total)
    ref = ref.cpu().numpy()
    got = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)], axis=0)
    # a rank's batches start at its shard's first frame, so batch membership (and with it the split-K plan of a ragged
    # last batch) differs from the single-process run: equal to rounding, not bit-exact
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), got)
    want = np.clip(np.rint(ref * 255), 0, 255).transpose(0, 2, 3, 1)
    assert np.abs(np.load(tmp_path / "gathered_u8.npy").astype(np.float32) - want).max() <= 1
    sd = synthetic_state_dict(cfg, seed=1234)
    orc_frames = np.stack(orc.animate_clip(sd, cfg, src, kp_s, kp_d)).transpose(0, 3, 1, 2)
    assert np.abs(got - orc_frames).max() <= TOL["prediction"]
