"""N>1 path on CPU: two gloo processes run eamm_amd.clip.animate_clip with a stand-in backend that
computes frames with the oracle.  Checks the protocol of the multi-GPU clip pipeline: one broadcast
of the source cache from rank 0, key points broadcast, contiguous shards, optional gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eamm_amd.clip import animate_clip, shard_bounds
from eamm_amd.config import tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc


class OracleBackend:
    """CPU stand-in for EngineBackend (tests only): the 'source cache' blob is the encoder feature
    map + the source itself, frames come from the oracle's decode path."""

    def __init__(self, cfg, sd, batch, size):
        self.cfg, self.sd, self.batch, self.size = cfg, sd, batch, size
        self.device = torch.device("cpu")
        self.blob = None
        self.calls = []

    def prepare(self, h, w):
        assert (h, w) == (self.size, self.size)

    def encode(self, src):
        with torch.no_grad():
            feat = orc.encode_source(self.sd, self.cfg, src)
        self.feat_shape = feat.shape
        self.blob = torch.cat([feat.flatten(), src.flatten()])
        return self.blob

    def blob_like(self):
        c = self.cfg
        hf = self.size >> c["num_down_blocks"]
        cb = min(c["max_features"], c["block_expansion"] << c["num_down_blocks"])
        self.feat_shape = (1, cb, hf, hf)
        return torch.empty(cb * hf * hf + 3 * self.size * self.size)

    def install(self, blob):
        self.blob = blob

    def run(self, kp_d, kp_s, uint8):
        n = kp_d["value"].shape[0]
        self.calls.append(n)
        nf = int(np.prod(self.feat_shape))
        src = self.blob[nf:].view(1, 3, self.size, self.size)
        srcb = src.expand(n, -1, -1, -1)
        ksb = {k: v.expand(n, *v.shape[1:]) for k, v in kp_s.items()}
        with torch.no_grad():
            pred = orc.generator_forward(self.sd, self.cfg, srcb, kp_d, ksb)["prediction"]
        if uint8:
            return torch.clamp(torch.round(pred * 255), 0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        return pred

    def finish(self):
        pass


COLLECTIVES = ("broadcast", "broadcast_object_list", "all_reduce", "all_gather", "all_gather_object", "gather", "gather_object",
               "scatter", "scatter_object_list", "reduce", "all_to_all", "barrier", "send", "recv")


class CountedCollectives:
    """Counts every torch.distributed collective issued while active (the clip pipeline calls them as ``dist.<name>``)."""

    def __init__(self):
        self.counts = {}
        self._saved = {}

    def __enter__(self):
        for name in COLLECTIVES:
            fn = getattr(dist, name)
            self._saved[name] = fn

            def counted(*a, __fn=fn, __name=name, **kw):
                self.counts[__name] = self.counts.get(__name, 0) + 1
                return __fn(*a, **kw)
            setattr(dist, name, counted)
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(dist, name, fn)


def _worker(rank, world, port, total, batch, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    be = OracleBackend(cfg, sd, batch, 64)
    if rank == 0:
        src, kp_s, kp_d = synthetic_source(64, seed=1), synthetic_keypoints(1, 10, seed=0), synthetic_keypoints(total, 10, seed=2)
    else:
        src = kp_s = kp_d = None  # only rank 0 holds the clip's inputs
    with CountedCollectives() as cc:
        local, (a, b) = animate_clip(be, src, kp_s, kp_d, 64, 64, uint8=False)
    # the data path of one clip: a fixed-size header + ONE packed payload (source cache | kp_source | kp_driving), nothing pickled
    assert cc.counts == {"broadcast": 2}, cc.counts
    assert (a, b) == shard_bounds(total, world, rank) and local.shape[0] == b - a
    assert all(n <= batch for n in be.calls) and sum(be.calls) == b - a
    np.save(os.path.join(tmp, f"shard{rank}.npy"), local.numpy())
    with CountedCollectives() as cc:
        full, span = animate_clip(be, src, kp_s, kp_d, 64, 64, uint8=True, gather=True)
    assert cc.counts == {"broadcast": 2, "gather": 1}, cc.counts
    if rank == 0:
        assert span == (0, total) and full.shape == (total, 64, 64, 3) and full.dtype == torch.uint8
        np.save(os.path.join(tmp, "gathered.npy"), full.numpy())
    else:
        assert full.shape[0] == 0
    dist.barrier()
    dist.destroy_process_group()


# (1, 2): the second rank's shard is EMPTY (more ranks than frames); (7, 4): four ranks, shards of 2 / 2 / 2 / 1 frames; (3, 8): the widest
# job the driver launches, five of its eight shards empty
@pytest.mark.parametrize("total,world", [(5, 2), (1, 2), (7, 4), (3, 8)])
def test_n_rank_clip_equals_single_process(tmp_path, total, world):
    batch = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, total, batch, str(tmp_path)), nprocs=world, join=True)
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    be = OracleBackend(cfg, sd, batch, 64)
    ref, span = animate_clip(be, synthetic_source(64, seed=1), synthetic_keypoints(1, 10, seed=0),
                             synthetic_keypoints(total, 10, seed=2), 64, 64)
    assert span == (0, total)
    got = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)], axis=0)
    assert got.shape == tuple(ref.shape)
    assert np.abs(got - ref.numpy()).max() <= 1e-6  # frames are independent: sharding must not change them
    u8 = np.load(tmp_path / "gathered.npy")
    want = np.clip(np.rint(ref.numpy() * 255), 0, 255).astype(np.uint8).transpose(0, 2, 3, 1)
    assert np.abs(u8.astype(int) - want.astype(int)).max() <= 1


def _chain_parts():
    """Oracle-backed stand-ins for the three front-end modules of make_animation_smooth (tiny shapes) + the clip backend."""
    from eamm_amd.config import tiny_kp_config
    from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, trained_like_kp_state_dict
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    ck, ca = tiny_kp_config(), tiny_kp_config(audio=True)
    sd_k, sd_a = trained_like_kp_state_dict(ck, 78), trained_like_kp_state_dict(ca, 77)
    ch = (64, 32, 32, 35)                                   # 1x1 -> 4x4 -> 8x8 -> 16x16 feature maps of 32 + 3 channels
    sd_d = synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec(ch))
    with torch.no_grad():
        kp = lambda x: orc.kp_detector_forward(sd_k, ck, x)
        tail = lambda x: orc.deconv_tail(sd_d, x)
        kpa = lambda f: orc.kp_detector_a_forward(sd_a, ca, f)
    return cfg, sd, kp, tail, kpa, synthetic_source(64, seed=1), synthetic_lstm_features(7, channels=64, seed=5)


def _chain_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from eamm_amd import animate_from_features
    cfg, sd, kp, tail, kpa, src, feats = _chain_parts()
    be = OracleBackend(cfg, sd, 2, 64)
    emo = {"value": 0.02 * torch.ones(7, 3, 2), "jacobian": 0.01 * torch.ones(7, 3, 2, 2)}
    args = (src, feats, emo) if rank == 0 else (None, None, None)     # only rank 0 holds the clip's inputs
    with CountedCollectives() as cc:
        frames, (a, b), kps = animate_from_features(None, kp, tail, kpa, args[0], args[1], emo_driving=args[2], backend=be, uint8=False,
                                                    to_host=True, front_batch=2, return_keypoints=True)
    # inputs: header + payload; key points: one all-gather; the generator's frames: no collective at all
    assert cc.counts == {"broadcast": 2, "all_gather": 1}, cc.counts
    assert (a, b) == shard_bounds(7, world, rank) and frames.shape[0] == b - a
    np.save(os.path.join(tmp, f"chain{rank}.npy"), frames.numpy())
    np.save(os.path.join(tmp, f"kpnorm{rank}.npy"), kps["kp_norm"]["value"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])     # 8 ranks over 7 frames: one frame each, the last rank's shard empty
def test_n_rank_end_to_end_chain_shards_the_front_end_too(tmp_path, world):
    """animate_from_features under a process group: both of the reference's loops shard by frames (demo.py:212-228 and :251-281);
    three collectives per clip; the result equals the single-process run."""
    from eamm_amd import animate_from_features
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_chain_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cfg, sd, kp, tail, kpa, src, feats = _chain_parts()
    emo = {"value": 0.02 * torch.ones(7, 3, 2), "jacobian": 0.01 * torch.ones(7, 3, 2, 2)}
    ref, span, kps = animate_from_features(None, kp, tail, kpa, src, feats, emo_driving=emo, backend=OracleBackend(cfg, sd, 2, 64),
                                           uint8=False, to_host=True, front_batch=3, return_keypoints=True)
    assert span == (0, 7) and ref.shape == (7, 3, 64, 64)
    got = np.concatenate([np.load(tmp_path / f"chain{r}.npy") for r in range(world)])
    for r in range(world):     # every rank ended up with the whole clip's normalised key points
        assert np.abs(np.load(tmp_path / f"kpnorm{r}.npy") - kps["kp_norm"]["value"].numpy()).max() <= 1e-6
    assert np.abs(got - ref.numpy()).max() <= 1e-5
    # and the chain is the oracle's own (reference statements frame by frame), here on the host filter
    from eamm_amd.config import tiny_kp_config
    from eamm_amd.weights import deconv_state_dict_spec, trained_like_kp_state_dict
    ck, ca = tiny_kp_config(), tiny_kp_config(audio=True)
    _, norm, _, _ = orc.animation_keypoints(trained_like_kp_state_dict(ck, 78), ck,
                                            synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec((64, 32, 32, 35))),
                                            trained_like_kp_state_dict(ca, 77), ca, src, feats, emo_driving=emo)
    want = torch.cat([n["value"] for n in norm])
    assert float((kps["kp_norm"]["value"] - want).abs().max()) <= 2e-5


def test_bare_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment (the command shape the driver uses)
    must become the launcher: two ranks of bench.py under torch.distributed.run that find each other on 127.0.0.1.
    EAMM_BENCH_RENDEZVOUS_ONLY=1 stops each rank after the rendezvous, before any GPU work (this container has none)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EAMM_BENCH_RENDEZVOUS_ONLY"] = "1"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=root,
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d == {"rendezvous": True, "world": 2, "gpus_arg": 2, "rank_sum": 1.0}
