#!/usr/bin/env python
"""bench.py -- 256x256 frames/sec of the dense-motion + generator forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Timing: W warm-up steps, then EXACTLY K steps between barrier + synchronize (the contract's region; nothing but the steps is
enqueued, the library's stage events are off), then the same K steps once more with the events on for the roofline / stage keys.

A "step" is one pass of the hot path (eamm_forward_frames: key points -> dense motion -> warp ->
decoder) over one batch of 16 synthetic driving frames of a 256x256 clip on each GPU, with the source
already encoded (the frame-invariant encoder runs once per clip; with N > 1 rank 0 encodes and the
cached source tensors are broadcast once over RCCL/xGMI before the timed region).  Inputs (key
points) are resident in HBM before the timed region; outputs stay on the device.  Frames of a clip
are independent, so the path shards by frames with no data-path collective: weak scaling, value =
all ranks' frames / max-over-ranks time.   `--size 512 --batch 8` is BASELINE.json configs[4].

Prints ONE JSON line with the contract keys plus
  roofline     -- the dominant kernel (the 3x3 256->256 bottleneck convolution: 12 launches per step, 67 % of
                  the reference FLOPs), FLOPs / its average launch duration measured with HIP events on the
                  launch stream inside the timed steps (executed-MFMA and reference-algorithmic figures);
                  `traffic` comes from profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes written
                  by tools/pmc_traffic.py) and only when that record was taken at this (size, batch) on the kernel
                  source that is being run now -- otherwise null;
  cpu_baseline -- the CPU oracle (PyTorch-CPU restatement of the reference, reference loop structure:
                  one frame per call, source encoder re-run every frame) timed on this box's host
                  cores on a bounded sample of the same workload, median of >= 5 passes;
  clip         -- BASELINE.json configs[3]: a `--clip-frames` (2048) frame clip through eamm_amd.animate_clip,
                  frames sharded contiguously over the ranks, timed from the un-encoded source on rank 0 to the
                  last frame on every rank: encode + RCCL broadcast of the source cache and key points + compute
                  (+ gather of uint8 frames to rank 0 with --clip-gather); frames/s = frames / max-over-ranks time;
  train_step   -- SURVEY.md 8f row N4, N = 1 only, never part of `value`: one fine-tuning step (`--train-pairs` = 8 pairs;
                  .train() forward with an autograd graph + loss.backward() on the HIP operators of eamm_amd/train_graph.py).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from eamm_amd import EngineBackend, OcclusionAwareGenerator, animate_clip, hot_path_config, shard_bounds  # noqa: E402
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
KERNEL_SOURCES = {4: "conv_winograd4.hip", 2: "conv_winograd.hip", 0: "conv_mfma_dma.hip"}   # by bottleneck form


def kernel_source_digest(form):
    path = os.path.join(ROOT, "eamm_amd", "csrc", KERNEL_SOURCES[form])
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(form, size, batch, chains=1):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 per the
    gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE) -- (bytes, source) or (None, reason).  A record counts only for
    the (size, batch) it was taken at and while the kernel's source file is byte-identical to the profiled one."""
    try:
        with open(TRAFFIC_FILE) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, f"{os.path.relpath(TRAFFIC_FILE, ROOT)} missing"
    rec = table.get(f"form{form}_{size}x{size}_b{batch}_c{chains}")
    if rec is None:
        return None, f"no PMC record for form {form} at {size}x{size} batch {batch} with {chains} chain(s)"
    if rec.get("source_sha256_16") != kernel_source_digest(form):
        return None, f"PMC record of {rec.get('files')} predates the current {KERNEL_SOURCES[form]}"
    return int((rec["fetch_size_kb"] * 2 + rec["write_size_kb"]) * 1024), rec.get("files")


FIXTURES = {256: "full256_clip2", 512: "full512_clip1"}   # reference outputs (oracle/make_golden.py) at the two BASELINE sizes
PARITY_TOL = 1e-4          # max |prediction - reference|, tests/conftest.py TOL["prediction"]
PLAN_TOL = 2e-5            # the same frames under two launch plans (summation order only), tests/test_gpu_plan64.py


def fixture_parity(pred, size):
    """max |pred[t] - reference| over the frames of the committed reference fixture of this size: frames t = 0.. of a clip whose
    driving key points are synthetic_keypoints(seed=2), source seed 1, weights seed 1234 -- exactly what the timed launches
    and the clip leg compute -- at the fixture's sampling stride.  Data only: nothing of the oracle or the reference runs here."""
    import numpy as np
    name = FIXTURES.get(size)
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz") if name else None
    if not path or not os.path.exists(path):
        return {"fixture": None, "ok": None, "note": f"no committed reference fixture at {size}x{size}"}
    z = np.load(path)
    want = torch.from_numpy(z["prediction"])
    st = int(z["prediction_stride"])
    n = want.shape[0]
    got = pred[:n, :, ::st, ::st].float().cpu()
    err = float((got - want).abs().max())
    return {"fixture": f"tests/golden/{name}.npz", "frames": n, "stride": st, "max_abs_err": err, "tolerance": PARITY_TOL,
            "ok": bool(err <= PARITY_TOL)}


def knob_record(eng, frames, extra_plans=None):
    """What the measured library was configured with: every EAMM_* variable in the environment, every knob the library read
    (value in effect, set or default) and the launch plan it chose for the timed call."""
    from eamm_amd import _lib
    from eamm_amd.engine import library_knobs
    rec = {"env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("EAMM_") and not k.startswith("EAMM_BENCH_")},
           "library": {k: v["value"] for k, v in library_knobs().items() if v["set"] == 1},
           "ignored": sorted(k for k, v in library_knobs().items() if v["set"] == 2),   # tuning aids present without EAMM_TUNING=1
           "library_defaults_read": len(library_knobs()),
           "experiments_build": int(_lib.lib().eamm_build_experiments()),
           "plan": eng.describe_plan(frames)}
    for name, (e, n) in (extra_plans or {}).items():
        rec[name] = e.describe_plan(n)
    return rec


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo; psutil as a fallback; None if unknown)."""
    try:
        pairs, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    try:
        import psutil
        return psutil.cpu_count(logical=False)
    except Exception:
        return None


def cpu_baseline(cfg, sd, size, frames, passes=5):
    """Reference-equivalent CPU loop (demo.py:251-281) on the oracle: B=1, encoder per frame; median of `passes`."""
    from oracle import eamm_oracle as orc  # checker / baseline only; never on the product path
    src = synthetic_source(size, seed=1)
    kp_s = synthetic_keypoints(1, cfg["num_kp"], seed=0)
    kp_d = synthetic_keypoints(frames + 1, cfg["num_kp"], seed=2)
    # pick the thread count that serves the CPU path best on this box (all logical CPUs oversubscribe
    # oneDNN on a B=1 256x256 frame); one timed frame per candidate, after a warm-up frame each
    ncpu = os.cpu_count() or 1
    best, best_t = 1, float("inf")
    with torch.no_grad():
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(nt)
            one = {k: v[:1] for k, v in kp_d.items()}
            orc.generator_forward(sd, cfg, src, one, kp_s)
            t = float("inf")
            for _ in range(2):          # best of two timed frames (one frame alone mis-ranked 8 vs 16 threads between boxes)
                t0 = time.perf_counter()
                orc.generator_forward(sd, cfg, src, one, kp_s)
                t = min(t, time.perf_counter() - t0)
            if t < best_t:
                best, best_t = nt, t
            if t > 5.0:   # already pathological; larger counts only get worse
                break
    torch.set_num_threads(best)
    # `passes` passes over the same frames, the whole sample bounded to ~20 s of CPU work
    per_pass = max(2, min(frames, int(20.0 / passes / max(best_t, 1e-3))))
    rates = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for _ in range(passes):
            t0 = time.perf_counter()
            for t in range(1, per_pass + 1):
                out = orc.generator_forward(sd, cfg, src, {k: v[t:t + 1] for k, v in kp_d.items()}, kp_s)
                out["prediction"].numpy()
            rates.append(per_pass / (time.perf_counter() - t0))
    dt = time.perf_counter() - t_all
    phys = physical_cores()
    return {"value": round(statistics.median(rates), 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            # `cores` (the contract's key) = the threads the timed passes used; the box itself (VERDICT r05 "weak" 8):
            "threads_used": torch.get_num_threads(), "physical_cores": phys, "logical_cpus": ncpu,
            "passes": [round(r, 3) for r in rates],
            "sample": f"median of {passes} passes x {per_pass} frames {size}x{size}, one generator call per frame incl. source "
                      f"encoder (reference loop demo.py:251-281), PyTorch-CPU fp32 oracle, best of 8..128 threads on "
                      f"{ncpu} logical CPUs, {dt:.1f} s"}


def train_step_target(pairs, size):
    """The training-step leg's L1 target: seeded, so that the printed loss / gradient checksums can be re-derived."""
    return torch.rand(pairs, 3, size, size, generator=torch.Generator().manual_seed(11))


def train_step_leg(cfg, sd, size, pairs, steps=4):
    """SURVEY.md 8f row N4 (not part of `value`): one fine-tuning step of the generator -- .train() forward with an autograd
    graph (eamm_amd/train_graph.py: HIP convolution / BatchNorm / warp operators) + loss.backward() -- `pairs` (source, driving)
    pairs per step, wall clock with the stream drained, best of `steps` after one warm-up step."""
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).train()       # parameters require grad by default, as the reference module's (train.py:136)
    src = synthetic_source(size, seed=1, batch=pairs).to(dev)
    kp_s = {k: v.to(dev) for k, v in synthetic_keypoints(pairs, cfg["num_kp"], seed=0).items()}
    kp_d = {k: v.to(dev).requires_grad_() for k, v in synthetic_keypoints(pairs, cfg["num_kp"], seed=2).items()}
    target = train_step_target(pairs, size).to(dev)
    from eamm_amd import _lib as _L
    fwd, bwd, gflop = [], [], []
    for it in range(steps + 1):
        for p in list(gen.parameters()) + list(kp_d.values()):
            p.grad = None
        torch.cuda.synchronize()
        f0 = _L.lib().eamm_total_mfma_flops()
        t0 = time.perf_counter()
        loss = (gen(src, kp_driving=kp_d, kp_source=kp_s)["prediction"] - target).abs().mean()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            fwd.append((t1 - t0) * 1e3)
            bwd.append((t2 - t1) * 1e3)
            gflop.append((_L.lib().eamm_total_mfma_flops() - f0) * 1e-9)
    best = min(f + b for f, b in zip(fwd, bwd))
    finite = all(bool(torch.isfinite(p.grad).all()) for p in gen.parameters())
    # deterministic inputs (seeds above): the loss and three gradient checksums of the last step, checked against the oracle's
    # autograd in double by tests/test_gpu_bench.py -- `gradients_finite` alone would pass a wrong kernel
    checks = {"loss": float(loss.detach()),
              "grad_l1/kp_driving.value": float(kp_d["value"].grad.abs().sum()),
              "grad_l1/final.weight": float(gen.final.weight.grad.abs().sum()),
              "grad_l1/bottleneck.r0.conv1.weight": float(gen.bottleneck.r0.conv1.weight.grad.abs().sum())}
    del gen
    torch.cuda.empty_cache()
    return {"pairs": pairs, "size": size, "step_ms": round(best, 3), "forward_ms": round(min(fwd), 3), "backward_ms": round(min(bwd), 3),
            "pairs_per_s": round(pairs / best * 1e3, 1), "gradients_finite": finite, "checks": checks,
            "roofline": {"bound": "mfma", "executed_gflop_per_step": round(gflop[-1], 1), "achieved": round(gflop[-1] / best, 2),
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gflop[-1] / best / FP32_MFMA_PEAK_TFLOPS, 4),
                         "note": "executed matrix-core GFLOP of one step (forward + backward, library-side count of what every launched "
                                 "grid issues, all host threads) / step time / chip fp32 matrix peak"},
            "note": "generator .train() forward with autograd graph + loss.backward() (L1 to a random target), HIP operators of "
                    "eamm_amd/train_graph.py; batch-statistics BatchNorm; not part of `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per step per GPU (default 16; 8 at --size 512)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--cpu-frames", type=int, default=64, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--clip-frames", type=int, default=2048, help="frames of the configs[3] clip leg (0 = skip)")
    ap.add_argument("--clip-batch", type=int, default=None,
                    help="frames per launch sequence of the clip leg (the clip harness is free to batch: 128 frames per call at 256x256 "
                         "-- four chains of 32 -- run 4 %% faster than 16, 2.7 %% faster than 64: profiles/r05_sweeps.txt; the contract "
                         "line above stays at --batch).  Default: 128 at 256x256, the same number of pixels per call at other sizes "
                         "(32 at 512x512: activations stay under the 4 GiB per tensor)")
    ap.add_argument("--clip-gather", action="store_true", help="clip leg: gather uint8 frames on rank 0 inside the timed region")
    ap.add_argument("--e2e-frames", type=int, default=2048,
                    help="frames of the end-to-end leg (make_animation_smooth: LSTM features -> uint8 frames in host memory; 0 = skip)")
    ap.add_argument("--e2e-front-batch", type=int, default=None, help="frames per front-end call of the end-to-end leg (default: the harness's own default)")
    ap.add_argument("--train-pairs", type=int, default=8, help="pairs per step of the training-step leg (N = 1 only; 0 = skip)")
    ap.add_argument("--latency-frames", type=int, default=64,
                    help="frames per pass of the one-frame-per-call leg `latency_b1` (BASELINE configs[1]; N = 1, 256x256 only; 0 = skip)")
    ap.add_argument("--no-all-outputs", action="store_true", help="skip the five-key `all_outputs` leg (profiling runs: the contract line's kernels only)")
    ap.add_argument("--graph", action="store_true",
                    help="after the contract's timed region: capture one step into a HIP graph and time `--steps` replays (extra key "
                         "`graph`).  Used by tools/gpu_profile.sh: under rocprofv3 the host's per-launch cost delays the second "
                         "chain's launches by ~1.8 ms, a replayed graph keeps the two chains' kernels together as an unprofiled run does")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16 if args.size <= 256 else 8

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- N ranks of this script under torch.distributed.run, one per
        # GPU, rendezvous on 127.0.0.1 (rank 0 prints the JSON line on the inherited stdout)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EAMM_BENCH_RENDEZVOUS_ONLY") == "1":
        # launcher check for the CPU test suite (tests/test_clip_sharding.py): the ranks this command started meet over gloo,
        # add up their ranks, and rank 0 reports -- no GPU is touched
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous": True, "world": world, "gpus_arg": args.gpus, "rank_sum": float(t.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    # EAMM_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a box with fewer GPUs than ranks (ranks then
    # share devices and collectives are staged through the host) -- used by tests/test_gpu_bench.py only.
    backend = os.environ.get("EAMM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev) if backend != "nccl" else local
    # EAMM_BENCH_FORCE_DIST=1 runs the collective code path even with one rank (a 1-GPU box can then check that
    # RCCL initialises and that broadcast / barrier / all-reduce work in this environment)
    use_dist = world > 1 or os.environ.get("EAMM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":   # = RCCL on ROCm, one rank per GPU over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def bcast(t):       # device tensor broadcast from rank 0
        if backend == "nccl":
            dist.broadcast(t, src=0)
        else:
            h = t.cpu()
            dist.broadcast(h, src=0)
            t.copy_(h)

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    cfg = hot_path_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = OcclusionAwareGenerator(**cfg, max_frames=args.batch)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).eval()
    B, S = args.batch, args.size
    eng = gen._ensure_engine(S, S, B, 1)
    from eamm_amd import _lib as _eamm_lib
    if _eamm_lib.lib().eamm_build_experiments():
        raise SystemExit("bench.py: this libeamm_hip.so was built with -DEAMM_EXPERIMENTS (timing experiments that compute wrong "
                         "results are compiled in): refusing to benchmark it -- rebuild with `make -C eamm_amd/csrc`")
    rccl_warmup_ms = None
    if use_dist:
        # communicator set-up (RCCL ring construction, first-call kernel loads) happens HERE, outside every timed region:
        # one broadcast + one all-reduce + one gather of the kinds the legs use, then a barrier
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wdev = dev if backend == "nccl" else "cpu"
        w = torch.ones(1 << 20, device=wdev)
        dist.broadcast(w, src=0)
        dist.all_reduce(w)
        bufs = [torch.empty(16, device=wdev) for _ in range(world)] if rank == 0 else None
        dist.gather(torch.zeros(16, device=wdev), bufs, dst=0)
        dist.barrier()
        torch.cuda.synchronize()
        rccl_warmup_ms = (time.perf_counter() - t0) * 1e3

    # once per clip: rank 0 encodes the source, the cached tensors are broadcast (the only collective)
    t_bcast_ms = None
    if rank == 0:
        eng.encode_source(synthetic_source(S, seed=1).to(dev))
    if use_dist:
        blob = eng.export_source_cache(1) if rank == 0 else torch.empty(eng.source_cache_numel(1), device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bcast(blob)
        torch.cuda.synchronize()
        t_bcast_ms = (time.perf_counter() - t0) * 1e3
        if rank != 0:
            eng.import_source_cache(blob, 1)
    kp_s = {k: v.to(dev) for k, v in synthetic_keypoints(1, cfg["num_kp"], seed=0).items()}
    # this rank's frames of the clip: contiguous shard, seeds 2 + global frame index
    kp_d = {k: v.to(dev) for k, v in synthetic_keypoints(B, cfg["num_kp"], seed=2 + rank * B).items()}

    def step():
        return eng.forward_frames(kp_d, kp_s, outputs=("prediction",))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # the contract's timed region: exactly K steps, nothing else enqueued (the library's stage events stay OFF here)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # the same K steps again with the library's HIP events on (stage boundaries, every bottleneck launch, every chain's bottleneck
    # window): the roofline / stage keys come from this second region (the events cost ~0.3 % of a step)
    eng.profile(True)
    eng.profile_read(reset=True)
    fence()
    t0p = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_prof = max_over_ranks(time.perf_counter() - t0p)
    prof = eng.profile_read(reset=True)
    eng.profile(False)
    eng.check_numeric()
    # the timed launches must have produced frames (parity itself is tests/' job: this only refuses to print a rate for
    # a build whose kernels write garbage): sigmoid-ranged, finite, and not constant
    graph_leg = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            g_out = step()["prediction"]
        g.replay()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        fence()
        dt_g = max_over_ranks(time.perf_counter() - t0)
        assert bool(torch.isfinite(g_out).all()) and float(g_out.std()) > 0.02
        graph_leg = {"steps": args.steps, "ms_per_step": round(dt_g / args.steps * 1e3, 4),
                     "value": round(args.steps * B * world / dt_g, 2), "unit": "frames/s",
                     "note": "the same step captured once into a HIP graph (both chains' streams) and replayed; not the contract's `value`"}
        del g, g_out
    chk = step()["prediction"]
    assert bool(torch.isfinite(chk).all()) and 0.0 < float(chk.min()) and float(chk.max()) < 1.0 and float(chk.std()) > 0.02, \
        "bench.py: the forward pass produced non-finite or degenerate frames"
    # parity spot check AT THE TIMED GEOMETRY: rank 0's first frames are the committed reference fixture's frames (same seeds);
    # a build or knob that computes something else does not get a `value` printed
    parity = fixture_parity(chk, S) if rank == 0 else None
    if parity is not None and parity["ok"] is False:
        raise SystemExit(f"bench.py: the timed launches do not reproduce the reference fixture: {json.dumps(parity)}")
    del chk
    dt = max_over_ranks(dt)

    # ---- the FULL forward: the reference's forward always produces five keys (generator.py:70-75,86,95); the
    # contract line asks for `prediction` only (SURVEY H9b sanctions the flag).  The same K steps with every key requested: 'deformed'
    # (flow up-sampling + image warp) and the NCHW exports of mask / sparse_deformed / occlusion_map are then inside the timed region.
    ALL_KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")

    def step_all():
        return eng.forward_frames(kp_d, kp_s, outputs=ALL_KEYS)

    all_outputs = None
    full = None
    if not args.no_all_outputs:
        for _ in range(max(2, args.warmup)):
            full = step_all()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full = step_all()
        fence()
        dt_all = max_over_ranks(time.perf_counter() - t0)
    if rank == 0 and full is not None:
        all_ok = all(k in full and bool(torch.isfinite(full[k]).all()) for k in ALL_KEYS)
        all_par = fixture_parity(full["prediction"], S)
        if not all_ok or (all_par["ok"] is False):
            raise SystemExit(f"bench.py: the five-key forward produced non-finite outputs or misses the fixture: {json.dumps(all_par)}")
        all_outputs = {"outputs": list(ALL_KEYS), "steps": args.steps, "ms_per_step": round(dt_all / args.steps * 1e3, 4),
                       "frames_per_s": round(args.steps * B * world / dt_all, 2),
                       "delta_ms_per_step_vs_prediction_only": round((dt_all - dt) / args.steps * 1e3, 4),
                       "ratio_to_value": round(dt / dt_all, 4), "prediction_max_abs_err_vs_fixture": all_par.get("max_abs_err"),
                       "note": "the same step with every key of the reference's forward requested (generator.py:70-75,86,95): + flow "
                               "up-sampling and the image warp of 'deformed', + NCHW exports of mask / sparse_deformed / occlusion_map"}
    del full

    # ---- ONE frame per call (BASELINE configs[1], the reference's own calling pattern demo.py:279): latency of the engine at
    # batch 1, source cached, key points resident, prediction left on the device; its own handle (max_frames = 1)
    latency_b1 = None
    if world == 1 and S == 256 and args.latency_frames > 0:
        LF = args.latency_frames
        gen1 = OcclusionAwareGenerator(**cfg, max_frames=1)
        gen1.load_state_dict(sd, strict=True)
        gen1 = gen1.to(dev).eval()
        eng1 = gen1.encode_source(synthetic_source(S, seed=1).to(dev), max_frames=1)
        kps1 = [{k: v.to(dev) for k, v in synthetic_keypoints(1, cfg["num_kp"], seed=2 + t).items()} for t in range(LF)]
        for t in range(min(16, LF)):
            eng1.forward_frames(kps1[t], kp_s, outputs=("prediction",))
        passes1 = []
        f0 = _eamm_lib.lib().eamm_total_mfma_flops()
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(LF):
                o1 = eng1.forward_frames(kps1[t], kp_s, outputs=("prediction",))
            torch.cuda.synchronize()
            passes1.append((time.perf_counter() - t0) / LF * 1e3)
        gf1 = (_eamm_lib.lib().eamm_total_mfma_flops() - f0) * 1e-9 / (5 * LF)
        eng1.check_numeric()
        # parity THROUGH the one-frame plan: the fixture's two frames, one call each
        kp2 = synthetic_keypoints(2, cfg["num_kp"], seed=2)
        par1 = fixture_parity(torch.cat([eng1.forward_frames({k: v[i:i + 1].to(dev) for k, v in kp2.items()}, kp_s,
                                                             outputs=("prediction",))["prediction"].clone() for i in range(2)]), S)
        if par1["ok"] is False:
            raise SystemExit(f"bench.py: the one-frame plan does not reproduce the reference fixture: {json.dumps(par1)}")
        ms1 = statistics.median(passes1)
        latency_b1 = {"ms_per_frame": round(ms1, 4), "frames_per_s": round(1e3 / ms1, 1), "best_ms": round(min(passes1), 4),
                      "executed_gflop_per_frame": round(gf1, 3), "achieved_tflops": round(gf1 / ms1, 2),
                      "frac_chip_executed": round(gf1 / ms1 / FP32_MFMA_PEAK_TFLOPS, 4),
                      "parity_max_abs_err_vs_fixture": par1.get("max_abs_err"), "plan": eng1.describe_plan(1),
                      "sample": f"median of 5 passes x {LF} calls of one frame each, back to back on one stream, no D2H (tools/module_latency.py "
                                "times the module wrapper incl. D2H)",
                      "workload": "256x256, 10 keypoints, batch=1 (BASELINE.json configs[1]; the reference's loop demo.py:251-281)"}
        del gen1, eng1, o1
        torch.cuda.empty_cache()

    # ---- BASELINE configs[3]: one whole clip, frame-sharded, from the un-encoded source to the last frame ---------
    clip = None
    if args.clip_frames > 0:
        T = args.clip_frames
        CB = args.clip_batch if args.clip_batch else max(B, 128 * 256 * 256 // (S * S))
        CB = max(1, min(CB, -(-T // world)))
        if CB == B:
            gen_clip = gen
        else:   # its own module + engine handle (workspace for CB frames per call); the contract line's handle stays as it is
            gen_clip = OcclusionAwareGenerator(**cfg, max_frames=CB)
            gen_clip.load_state_dict(sd, strict=True)
            gen_clip = gen_clip.to(dev).eval()
        be = EngineBackend(gen_clip, batch=CB)
        if rank == 0:
            c_src = synthetic_source(S, seed=1).to(dev)
            c_kps = {k: v.to(dev) for k, v in synthetic_keypoints(1, cfg["num_kp"], seed=0).items()}
            c_kpd = {k: v.to(dev) for k, v in synthetic_keypoints(T, cfg["num_kp"], seed=2).items()}
        else:
            c_src = c_kps = c_kpd = None

        def run_clip(timings=None, keep=False):
            out, span = animate_clip(be, c_src, c_kps, c_kpd, S, S, uint8=args.clip_gather, gather=args.clip_gather,
                                     timings=timings)
            n = out.shape[0]
            if keep:
                return out, span
            del out
            return n, span

        run_clip()                        # warm-up pass (allocator, RCCL channels)
        fence()
        t0 = time.perf_counter()
        clip_out, span = run_clip(keep=True)
        fence()
        dt_clip = max_over_ranks(time.perf_counter() - t0)
        # what the TIMED pass produced, checked (it used to be deleted unseen): (i) rank 0's first frames against the committed
        # reference fixture, (ii) on every rank three frames of its shard -- first, middle, last -- against the same frames
        # recomputed by the contract handle (its launch plan: `--batch` frames per call) from this rank's own copy of the
        # key points, (iii) range.  The clip's plan (64 frames per call = four chains) and the shards are thereby verified
        # in the run that is reported; a mismatch refuses the line.
        a_sh, b_sh = (span if not args.clip_gather else shard_bounds(T, world, rank))
        verify = {"ok": True}
        if clip_out.shape[0] and not args.clip_gather:
            assert bool(torch.isfinite(clip_out).all()) and 0.0 < float(clip_out.min()) and float(clip_out.max()) < 1.0
            if rank == 0:
                verify["fixture"] = fixture_parity(clip_out, S)
                verify["ok"] = verify["fixture"]["ok"] is not False
            spots = sorted({a_sh, (a_sh + b_sh) // 2, b_sh - 1})[:B]   # (the contract handle holds the same source: encoded /
            # imported before the timed region above)
            kp_sp = {k: torch.cat([v.to(dev) for v in [synthetic_keypoints(1, cfg["num_kp"], seed=2 + t)[k] for t in spots]])
                     for k in ("value", "jacobian")}
            again = eng.forward_frames(kp_sp, kp_s, outputs=("prediction",))["prediction"]
            worst = float((again - clip_out[[t - a_sh for t in spots]]).abs().max())
            verify.update({"spot_frames": spots, "vs_contract_plan_max_abs": worst, "plan_tolerance": PLAN_TOL})
            verify["ok"] = bool(verify["ok"] and worst <= PLAN_TOL)
            del again
        ok_all = max_over_ranks(0.0 if verify["ok"] else 1.0) == 0.0
        if not ok_all:
            raise SystemExit(f"bench.py: the clip leg's frames failed verification on some rank (rank {rank}: {json.dumps(verify)})")
        n_local = clip_out.shape[0]
        del clip_out
        phases = {}
        run_clip(phases)                  # third pass with a device sync at each phase boundary: where the time goes
        fence()
        # every rank's phases, gathered (one fixed-order float tensor; outside every timed region)
        PH = ("keypoints_ms", "encode_ms", "broadcast_ms", "compute_ms", "gather_ms")
        mine = torch.tensor([phases.get(k, 0.0) for k in PH], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if use_dist:
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
        else:
            allp = [mine]
        phases_all = [{k: round(float(v), 3) for k, v in zip(PH, t.tolist())} for t in allp]
        clip_plan = be.engine.describe_plan(min(CB, max(1, b_sh - a_sh))) if rank == 0 else None
        if gen_clip is not gen:
            del be, gen_clip
        torch.cuda.empty_cache()
        if rank == 0:
            clip = {"frames": T, "frames_per_s": round(T / dt_clip, 2), "seconds": round(dt_clip, 4), "n_gpus": world,
                    "frames_per_s_per_gpu": round(T / dt_clip / world, 2), "batch": CB, "shard_rank0": list(shard_bounds(T, world, 0)),
                    "timed": "source encode (rank 0) + broadcast of source cache and key points + compute of every "
                             "shard" + (" + uint8 gather on rank 0" if args.clip_gather else "") + ", max over ranks",
                    "phases_ms_rank0": {k: round(v, 3) for k, v in phases.items()},
                    "phases_ms_per_rank": phases_all, "verify": verify, "plan": clip_plan,
                    "collectives_per_clip": 0 if world == 1 else 2 + int(bool(args.clip_gather)),
                    "workload": f"{S}x{S}, {T}-frame clip, contiguous shards of {T}/{world} frames, batch {CB} "
                                f"(BASELINE.json configs[3])"}

    # ---- end-to-end leg: the reference's make_animation_smooth (demo.py:194-282) whole -- KPDetector on the
    # source, DeconvTail + KPDetector_a per frame, One-Euro smoothing, normalize_kp, generator, uint8 frames in pinned HOST memory
    e2e = None
    if args.e2e_frames > 0 and S == 256:
        from eamm_amd import DeconvTail, KPDetector, KPDetector_a, animate_from_features, kp_detector_a_config, kp_detector_config
        from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, trained_like_kp_state_dict
        T2 = args.e2e_frames
        CB2 = max(1, min(128, -(-T2 // world)))
        gen_e = OcclusionAwareGenerator(**cfg, max_frames=CB2)
        gen_e.load_state_dict(sd, strict=True)
        gen_e = gen_e.to(dev).eval()
        ck, ca = kp_detector_config(), kp_detector_a_config()
        m_kp, m_kpa, m_tail = KPDetector(**ck), KPDetector_a(**ca), DeconvTail()
        m_kp.load_state_dict(trained_like_kp_state_dict(ck, 78), strict=True)
        m_kpa.load_state_dict(trained_like_kp_state_dict(ca, 77), strict=True)
        m_tail.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()), strict=True)
        m_kp, m_kpa, m_tail = m_kp.to(dev).eval(), m_kpa.to(dev).eval(), m_tail.to(dev).eval()
        be2 = EngineBackend(gen_e, batch=CB2)
        e_src = synthetic_source(S, seed=1).to(dev) if rank == 0 else None
        e_feat = synthetic_lstm_features(T2, seed=5).to(dev) if rank == 0 else None

        def run_e2e(timings=None, keys=False):
            extra = {} if args.e2e_front_batch is None else {"front_batch": args.e2e_front_batch}
            return animate_from_features(gen_e, m_kp, m_tail, m_kpa, e_src, e_feat, batch=CB2, uint8=True, to_host=True, backend=be2,
                                         timings=timings, return_keypoints=keys, size=(S, S), **extra)

        run_e2e()                          # warm-up: engines, pinned buffer, allocator
        fence()
        t0 = time.perf_counter()
        e_frames, e_span = run_e2e()
        fence()
        dt_e2e = max_over_ranks(time.perf_counter() - t0)
        e_ph = {}
        e_frames2, _, e_kps = run_e2e(e_ph, keys=True)
        fence()
        # verification of what was delivered: the host frames of the timed pass == the instrumented pass (deterministic), and on
        # rank 0 three frames recomputed by the contract handle from the harness's own normalised key points, to one uint8 level
        # (the timed pass STREAMS -- front end beside the generator --, the instrumented one runs the phases one after the other)
        # (the two passes must not share host storage or the comparison is vacuous -- the backend's pinned pool never
        #  hands out a buffer the caller still holds, and the line says so)
        e_distinct = e_frames.numel() == 0 or e_frames.data_ptr() != e_frames2.data_ptr()
        e_same = int((e_frames.to(torch.int16) - e_frames2.to(torch.int16)).abs().max()) if e_frames.numel() else 0
        e_ok = e_distinct and e_same <= 1 and e_frames.dtype == torch.uint8 and e_frames.is_pinned()
        e_worst = None
        if rank == 0 and e_frames.shape[0]:
            a2, b2 = e_span
            spots = sorted({a2, (a2 + b2) // 2, b2 - 1})[:B]
            again = eng.forward_frames({k: v[spots].contiguous() for k, v in e_kps["kp_norm"].items()},
                                       {k: v.contiguous() for k, v in e_kps["kp_source"].items()}, outputs=("prediction",))["prediction"]
            want = torch.clamp(torch.round(again * 255), 0, 255).permute(0, 2, 3, 1).cpu()
            e_worst = float((e_frames[[t - a2 for t in spots]].float() - want).abs().max())
            e_ok = e_ok and e_worst <= 1
        if max_over_ranks(0.0 if e_ok else 1.0) != 0.0:
            raise SystemExit(f"bench.py: the end-to-end leg's frames failed verification (rank {rank}, uint8 levels {e_worst})")
        if rank == 0:
            e2e = {"frames": T2, "frames_per_s": round(T2 / dt_e2e, 2), "seconds": round(dt_e2e, 4), "n_gpus": world, "batch": CB2,
                   "phases_ms_rank0": {k: round(v, 3) for k, v in e_ph.items()},
                   "delivered": "uint8 [T,H,W,3] frames in pinned host memory (non_blocking copies on a copy stream, overlapped with the next batch)",
                   "host_bytes": int(e_frames.numel()),
                   "verify": {"ok": True, "uint8_levels_vs_contract_plan": e_worst, "uint8_levels_streamed_vs_phased": e_same,
                              "passes_in_distinct_host_buffers": bool(e_distinct)},
                   "streamed": "timed pass: the detectors / smoothing / normalisation of later frames run on their own stream beside the "
                               "generator of earlier ones (animate_from_features stream=True); phases_ms come from a second, un-streamed pass",
                   "timed": "LSTM features + source on the device -> KPDetector, DeconvTail + KPDetector_a, One-Euro smoothing, "
                            "normalize_kp (relative, adapt_movement_scale), source encode, generator, D2H of every frame; max over ranks",
                   "workload": f"make_animation_smooth (reference demo.py:194-282) on a {T2}-frame clip at {S}x{S}, synthetic LSTM features and weights"}
        del e_frames, e_frames2, be2, gen_e, m_kp, m_kpa, m_tail
        torch.cuda.empty_cache()

    # isolated launch of the HBM-bound warp kernel at the step's full batch (in the pipeline a launch covers one chain's frames
    # and runs beside the other chain's kernels): C-ABI op entry, HIP events on the launch stream
    warp_iso_ms = None
    if rank == 0 and gen.dense_motion_network is not None:
        import ctypes as C
        from eamm_amd import _lib
        hf_, cb_ = S >> cfg["num_down_blocks"], min(cfg["max_features"], cfg["block_expansion"] << cfg["num_down_blocks"])
        g0 = torch.Generator(device="cpu").manual_seed(7)
        feat = torch.rand(1, hf_, hf_, cb_, generator=g0).to(dev)
        ident = torch.stack(torch.meshgrid(torch.linspace(-1, 1, eng.h), torch.linspace(-1, 1, eng.w), indexing="ij")[::-1], -1)
        defo = (ident[None] + 0.05 * torch.randn(B, eng.h, eng.w, 2, generator=g0)).contiguous().to(dev)
        occ = torch.rand(B, eng.h, eng.w, generator=g0).to(dev)
        wout = torch.empty(B, hf_, hf_, cb_, device=dev)
        ms = C.c_float(0.0)
        _lib.check(_lib.lib().eamm_op_warp(dev.index, C.c_void_p(feat.data_ptr()), C.c_void_p(defo.data_ptr()), C.c_void_p(occ.data_ptr()),
                                           B, 1, hf_, hf_, cb_, eng.h, eng.w, C.c_void_p(wout.data_ptr()), 50, C.byref(ms),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), None)
        warp_iso_ms = float(ms.value)
        del feat, defo, occ, wout

    if rank == 0:
        frames = args.steps * B * world
        fps = frames / dt
        # dominant kernel: the bottleneck 3x3 256->256 convolution, 2*num_bottleneck_blocks launches per step (per chain).
        # From 5 frames per call it runs as wino4_gemm_kernel (Winograd F(4x4,3x3): 36 GEMMs over the transformed
        # input, 4x fewer MACs than the reference's direct convolution; F(2x2,3x3) / wino_gemm_kernel when a map side
        # is not a multiple of 4), else as the direct LDS-DMA conv_mfma_dma_kernel (eamm_bottleneck_form).
        # Everything under "roofline" is priced against the CHIP's fp32 matrix peak:
        #   frac             = executed GEMM flops of ALL chains in a step / wall time during which any chain is inside its
        #                      bottleneck stage (union of the chains' windows, HIP events on every chain's stream) / 157.3
        #   per_launch       = one launch of the main stream: executed flops / its own duration, against the chip peak and
        #                      against the share of the chip's CUs its grid can occupy (the round-2 headline, now labelled)
        #   whole_path       = executed matrix-core flops of the whole step (every kernel, library-side count) / step time
        #   achieved_algorithmic prices the stage at the reference's direct-convolution FLOPs (can exceed the peak)
        hf = S >> cfg["num_down_blocks"]
        cb = min(cfg["max_features"], cfg["block_expansion"] << cfg["num_down_blocks"])
        calls = max(1, prof["calls"])
        nres = 2 * cfg["num_bottleneck_blocks"]
        launches = nres * calls
        form = eng.bottleneck_form(B)          # 0 direct, 2 Winograd F(2x2,3x3), 4 Winograd F(4x4,3x3)
        algo_flop_step = 2.0 * (B * hf * hf) * cb * (9 * cb) * nres          # reference FLOPs of the stage, all frames
        chains = eng.bottleneck_chains(B)
        pchains = eng.pass_chains(B)
        pm = prof["ms"]
        kernel_ms = pm.pop("bneck_gemm_kernel")          # GEMM kernels' own durations on the main stream
        union_ms = pm.pop("bneck_union") / calls          # any chain inside its bottleneck stage
        windows_ms = pm.pop("bneck_windows") / calls      # sum of the chains' windows
        exec_gf_step = pm.pop("exec_gflop") / calls       # executed matrix-core GFLOP per step, all kernels, all chains
        bneck_exec_gf_step = pm.pop("bneck_exec_gflop") / calls
        ms_conv = (kernel_ms if form != 0 else pm["bneck_conv"]) / launches if prof["calls"] else float("nan")
        exec_gf_launch = bneck_exec_gf_step / nres / chains       # one main-stream launch covers B / chains frames
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if form == 4:
            blocks = -(-(B // chains) * (hf // 4) * (hf // 4) // 64) * -(-cb // 64)
        else:
            blocks = cus
        shared = pchains > 1 and blocks * chains > cus
        cu_share = 1.0 / chains if shared else min(1.0, blocks / cus)
        ms_tr = pm["bneck_transform"] / launches if prof["calls"] else 0.0
        per_launch = exec_gf_launch / ms_conv                      # GFLOP / ms = TFLOP/s
        stage_tflops = bneck_exec_gf_step / union_ms if union_ms > 0 else float("nan")
        algo = algo_flop_step / 1e9 / union_ms if union_ms > 0 else float("nan")
        ms_step = dt / args.steps * 1e3
        ms_step_prof = dt_prof / args.steps * 1e3          # the events' region (stage times, executed-flop rates)
        traffic, traffic_src = measured_traffic(form, S, B, chains)
        which = "configs[2]" if (S, B) == (256, 16) else ("configs[4]" if (S, B) == (512, 8) else "a non-BASELINE size")
        warp_bytes_frame = (2 * hf * hf * cb + 3 * (S // 4) * (S // 4)) * 4.0    # 8.438 MB at 256^2
        warp_joint = os.environ.get("EAMM_WARP_JOINT") == "1" and pchains > 1      # one launch for all chains' frames (knob, off by default)
        warp_frames = B if warp_joint else B // pchains                         # frames of ONE in-pipeline launch
        warp_ms = pm["warp"] / calls

        # per-stage fraction of the chip's fp32 matrix peak: executed matrix-core GFLOP of the stage (ALL chains, library-side
        # count of what the grids issue) / the main stream's time in the stage -- the other chain runs the same stage beside it
        # and takes the same time within a few per cent (compare bneck_union with bneck_windows / chains)
        stage_gf = {k[3:]: pm.pop(k) / calls for k in list(pm) if k.startswith("gf_")}
        stage_time = {"front": pm["front"], "hg_enc": pm["hg_enc"], "hg_dec": pm["hg_dec"], "head": pm["head"], "warp": pm["warp"],
                      "bneck": pm["bneck_transform"] + pm["bneck_conv"], "up": pm["up"], "final": pm["final"]}
        stage_roofline = {}
        for k, gf in stage_gf.items():
            ms = stage_time[k] / calls
            if gf > 0 and ms > 0:
                stage_roofline[k] = {"executed_gflop_per_step": round(gf, 2), "ms": round(ms, 4),
                                     "tflops": round(gf / ms, 2), "frac_chip": round(gf / ms / FP32_MFMA_PEAK_TFLOPS, 4)}
        stage_roofline["note"] = ("executed matrix-core GFLOP of all chains per step / the main stream's time in the stage / 157.3; "
                                  "stages without matrix-core work (front, warp) are HBM-bound: see roofline_warp")
        total_ms = sum(pm.values())

        def hbm(by, ms):
            return {"achieved": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_launch": int(by), "avg_launch_ms": round(ms, 4)}

        line = {
            "metric": "256x256 frames/sec (dense-motion + generator forward)" if S == 256 else f"{S}x{S} frames/sec",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S}x{S}, 10 keypoints, batch={B} synthetic frames per GPU per step, source "
                                   f"encoded once per clip (BASELINE.json {which})",
                       "frames_per_step_per_gpu": B, "parallelism": f"frame-sharded x{world}",
                       "flops_per_frame": round(eng.flops_per_frame / 1e9, 3)},
            "roofline": {"bound": "mfma", "flops_basis": "executed (what the kernels' grids issue: Winograd F(4x4) = 1/4 of the reference's "
                                                         "multiplies); `achieved` / `frac` / `per_launch` / `whole_path.executed_*` all use it",
                         "kernel": {4: f"wino4_gemm_kernel (bottleneck 3x3 {cb}->{cb} @{hf}x{hf} in Winograd F(4x4,3x3) form)",
                                    2: f"wino_gemm_kernel<1,2,4,2> (bottleneck 3x3 {cb}->{cb} @{hf}x{hf} in Winograd F(2x2,3x3) form)",
                                    0: f"conv_mfma_dma_kernel<3,3,...> (bottleneck 3x3 {cb}->{cb} @{hf}x{hf}, direct)"}[form],
                         "achieved": round(stage_tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(stage_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                         "definition": "executed GEMM flops of all chains per step / wall time with any chain inside its bottleneck "
                                       "stage (union of per-chain windows; includes the stage's input transforms and whatever the "
                                       "other chain runs beside it) / chip fp32 matrix peak",
                         "bneck_executed_gflop_per_step": round(bneck_exec_gf_step, 2),
                         "bneck_union_ms_per_step": round(union_ms, 4), "bneck_windows_sum_ms_per_step": round(windows_ms, 4),
                         "chains": chains, "pass_chains": pchains, "cus": cus,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "per_launch": {"executed_gflop": round(exec_gf_launch, 3), "avg_launch_ms": round(ms_conv, 4),
                                        "frames": B // chains, "launch_blocks": blocks,
                                        "achieved": round(per_launch, 2),
                                        "frac_chip": round(per_launch / FP32_MFMA_PEAK_TFLOPS, 4),
                                        "cu_share": round(cu_share, 4),
                                        "frac_of_occupied_cus": round(per_launch / (FP32_MFMA_PEAK_TFLOPS * cu_share), 4),
                                        "stream": "chain 0 (the main stream; the other chains' launches carry no events)",
                                        "note": "one main-stream launch (HIP events around the kernel); with chains it covers "
                                                "1/chains of the frames and the other chain's kernels run beside it: frac_chip is "
                                                "against the whole chip, frac_of_occupied_cus against the CUs its grid can occupy"},
                         "whole_path": {"executed_gflop_per_frame": round(exec_gf_step / B, 3),
                                        "executed_tflops": round(exec_gf_step / ms_step, 2),
                                        "frac_chip_executed": round(exec_gf_step / ms_step / FP32_MFMA_PEAK_TFLOPS, 4),
                                        "algorithmic_tflops": round(fps / world * eng.flops_per_frame / 1e12, 2)},
                         "achieved_algorithmic": round(algo, 2),
                         "frac_algorithmic": round(algo / FP32_MFMA_PEAK_TFLOPS, 4),   # whole stage vs the CHIP peak
                         "algorithmic_note": "reference-equivalent: the REFERENCE's direct-convolution FLOPs of the stage / the same time; "
                                             "exceeds 1 by construction (the kernels execute 4x fewer multiplies) -- not a hardware "
                                             "utilisation figure, see `frac`",
                         "algorithmic_gflop_per_step": round(algo_flop_step / 1e9, 2),
                         "avg_input_transform_ms": round(ms_tr, 4)},
            # the HBM-bound kernel of the path (north_star: ">= 60 % of HBM roofline on the warp"): feature warp x occlusion,
            # algorithmic bytes per frame = feature map read + written once + flow + occlusion
            "roofline_warp": dict({"bound": "hbm", "kernel": "warp_features_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frames_per_launch": warp_frames,
                                   "note": ("in the pipeline: ONE launch for all chains' frames behind a join of the chains (EAMM_WARP_JOINT=1; "
                                            "stage events incl. the boundary)") if warp_joint else
                                           ("in the pipeline: one launch of the main stream's chain (stage events incl. the boundary), "
                                            "beside the other chain's kernels; EAMM_WARP_JOINT=1 joins the chains for one launch over all "
                                            "frames: 0.58 of the HBM peak in the pipeline, -0.8 % frames/s (profiles/r04_experiments.txt)")},
                                  **hbm(warp_frames * warp_bytes_frame, warp_ms),
                                  **({"isolated": dict({"frames_per_launch": B, "note": "the same kernel alone on the chip "
                                                        "(eamm_op_warp, 50 back-to-back launches)"},
                                                       **hbm(B * warp_bytes_frame, warp_iso_ms))} if warp_iso_ms else {})),
            "stage_ms_per_step": {k: round(v / calls, 4) for k, v in pm.items()},
            "stage_sum_ms": round(total_ms / calls, 4),
            "stage_region_ms_per_step": round(ms_step_prof, 4),   # wall clock of the second (event-instrumented) K steps
            "stage_roofline": stage_roofline,
        }
        line["parity_check"] = parity
        line["knobs"] = knob_record(eng, B)
        if t_bcast_ms is not None:
            line["source_broadcast_ms"] = round(t_bcast_ms, 3)
        if rccl_warmup_ms is not None:
            line["rccl_warmup_ms"] = round(rccl_warmup_ms, 2)    # communicator set-up, before every timed region
        line["all_outputs"] = all_outputs
        line["latency_b1"] = latency_b1
        line["clip"] = clip
        line["e2e_clip"] = e2e
        if graph_leg is not None:
            line["graph"] = graph_leg
        if world == 1 and args.cpu_frames > 0:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, S, args.cpu_frames)
            line["gpu_over_cpu"] = round(fps / line["cpu_baseline"]["value"], 1)
        else:
            line["cpu_baseline"] = None
        if world == 1 and args.train_pairs > 0:
            try:
                line["train_step"] = train_step_leg(cfg, sd, S, args.train_pairs if S <= 256 else max(1, args.train_pairs // 4))
            except Exception as exc:   # never at the expense of the contract line
                line["train_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
