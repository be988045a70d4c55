"""Clip pipeline: the counterpart of the per-frame loop in reference demo.py:251-281.

Reference: for each of T driving frames call ``generator(source, kp_source, kp_driving[t])`` (which
re-runs the source encoder every time) and copy ``out['prediction']`` to the host (one sync per
frame).  Here: the source is encoded ONCE, frames are processed ``batch`` at a time, results stay on
the device (optionally as uint8 HWC frames, the format demo.py:507 writes), and -- because the
generator has no cross-frame dependence -- a clip shards across the GPUs of a node by contiguous
frame ranges.  The data path has TWO collectives per clip, both broadcasts from rank 0 over RCCL/xGMI: a six-integer
header and one float32 payload = cached source tensors (encoder feature map + down-sampled source +
full-resolution source, ~5 MB at 256x256) | kp_source | kp_driving (0.5 MB for 2048 frames); outputs are
gathered only if asked for (one more).

``torch.distributed`` is used as plumbing: backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the
CPU tests, which drive this file with a stand-in backend object (tests/test_clip_sharding.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced frame range [start, stop) of `rank` (first n%world ranks get one more)."""
    base, rem = divmod(n_frames, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class EngineBackend:
    """Runs the two halves of the path on one GPU through the HIP library."""

    def __init__(self, generator, batch: int = 16):
        self.generator = generator
        self.batch = int(batch)
        self.device = next(generator.parameters()).device
        self.engine = None
        self._host = {}             # pinned delivery buffers by (dtype, frame shape): host_buffer()
        self._copy_stream = None    # device -> host copies beside the kernels: copy_stream()
        self._front_stream = None   # the streamed front end of animate_from_features

    def prepare(self, height: int, width: int):
        self.engine = self.generator._ensure_engine(height, width, self.batch, 1)

    def encode(self, source_image: torch.Tensor) -> torch.Tensor:
        self.engine.encode_source(source_image.to(self.device))
        return self.engine.export_source_cache(1)

    def blob_like(self) -> torch.Tensor:
        return torch.empty(self.engine.source_cache_numel(1), dtype=torch.float32, device=self.device)

    def blob_numel(self) -> int:
        return self.engine.source_cache_numel(1)

    def install(self, blob: torch.Tensor):
        self.engine.import_source_cache(blob, 1)

    def run(self, kp_driving: Dict[str, torch.Tensor], kp_source: Dict[str, torch.Tensor], uint8: bool):
        out = self.engine.forward_frames(kp_driving, kp_source, outputs=("prediction",), uint8_frames=uint8)
        return out["frames_u8"] if uint8 else out["prediction"]

    def finish(self):
        self.engine.check_numeric()

    # -- device -> host delivery (the reference copies every frame to the host, demo.py:281) -----------------------------
    def host_buffer(self, frames: int, tail, dtype) -> torch.Tensor:
        """A pinned host tensor [frames, *tail] out of a grow-only pool (pinning 400 MB costs ~100 ms, so buffers are reused
        from clip to clip) -- but never one the caller still holds: a buffer goes back into circulation only when nothing outside
        the pool references its storage (the tensor `animate_clip(..., to_host=True)` returned, a view of it, a numpy array made
        from it), so a delivered clip stays valid for as long as the caller keeps it, as a fresh tensor would."""
        need = int(frames)
        for d in tail:
            need *= int(d)
        key = (dtype, tuple(tail))
        if getattr(torch._C, "_storage_Use_Count", None) is None:      # no reference counter in this torch build: a fresh buffer per clip
            return torch.empty(max(need, 1), dtype=dtype, pin_memory=True)[:need].view((frames,) + tuple(tail))
        pool = self._host.setdefault(key, [])
        buf = None
        for cand, idle_count in pool:
            if cand.numel() >= need and _storage_use_count(cand) <= idle_count:
                buf = cand
                break
        if buf is None:
            # drop idle buffers that are too small before pinning a larger one (grow-only, but not a leak)
            pool[:] = [(c, n) for c, n in pool if _storage_use_count(c) > n]
            buf = torch.empty(max(need, 1), dtype=dtype, pin_memory=True)
            pool.append((buf, _storage_use_count(buf)))
        return buf[:need].view((frames,) + tuple(tail))

    def copy_stream(self):
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        return self._copy_stream


@torch.no_grad()
def driving_keypoints(deconv_tail, kp_detector_a, lstm_features: torch.Tensor, batch: int = 64) -> Dict[str, torch.Tensor]:
    """Per-frame driving key points of a clip from the audio network's LSTM output, on the device.

    Reference: ``AT_net2.forward`` runs ``self.decon`` on ``lstm_out[:, t]`` for every step t (modules/util.py:600-607)
    and demo.py:219 feeds each ``deco_out[:, t]`` to ``kp_detector_a`` -- two batch-1 module calls per frame.  Here the
    T frames go through ``DeconvTail`` and ``KPDetector_a`` ``batch`` at a time.  ``lstm_features``: [T,256] (or the
    reference's [1,T,256]).  Returns {'value': [T,K,2], 'jacobian': [T,K,2,2]} (no heat-maps: demo.py never reads them).
    """
    if lstm_features.dim() == 3 and lstm_features.shape[0] == 1:
        lstm_features = lstm_features[0]
    if lstm_features.dim() != 2:
        raise RuntimeError(f"expected [T,C] LSTM features, got {tuple(lstm_features.shape)}")
    parts: Dict[str, List[torch.Tensor]] = {}
    # round 6: both ends of `deco_out[:, t]` live in the library -- when the tail's output is 32 m + 3 channels wide (the shipped 35) it
    # is handed to the heads in the layout they read (NHWC wide part + one float4 per pixel) instead of the reference's NCHW tensor
    split = getattr(deconv_tail, "forward_split", None) if getattr(deconv_tail, "split_channels", lambda: 0)() else None
    for t0 in range(0, lstm_features.shape[0], batch):
        x = lstm_features[t0:t0 + batch].contiguous()
        fm = split(x) if split is not None else deconv_tail(x)
        detect = getattr(kp_detector_a, "detect", None)         # (stand-in detectors of the CPU tests have no such method)
        kp = detect(fm, heatmap=False) if detect is not None else kp_detector_a(fm)
        for k in ("value", "jacobian"):
            if k in kp:
                parts.setdefault(k, []).append(kp[k])
    return {k: torch.cat(v, 0) for k, v in parts.items()}


def _storage_use_count(t: torch.Tensor) -> int:
    """References to the tensor's storage (views, numpy arrays and the tensor itself all count)."""
    return int(torch._C._storage_Use_Count(t.untyped_storage()._cdata))


def _staged(group, device) -> bool:
    """True when collectives on `device` tensors must go through the host: the gloo backend only moves CPU tensors
    for gather (and stages the rest itself); RCCL ("nccl") takes device tensors directly over xGMI."""
    return torch.device(device).type == "cuda" and dist.get_backend(group) != "nccl"


def _broadcast(t: torch.Tensor, src: int, group, staged: bool):
    if not staged:
        dist.broadcast(t, src=src, group=group)
        return
    h = t.cpu()
    dist.broadcast(h, src=src, group=group)
    if dist.get_rank(group) != src:
        t.copy_(h)


_HEADER_FIELDS = 6   # T, K, driving jacobian present, source jacobian present, source sets, blob elements


def _broadcast_clip(blob: Optional[torch.Tensor], kp_source, kp_driving, blob_numel: int, device, src: int, group,
                    staged: bool):
    """Everything the ranks need for one clip in TWO collectives: a fixed six-integer header (frame count, key-point count,
    which jacobians exist, source sets, cache size) and ONE float32 payload = source cache | kp_source | kp_driving.
    No pickled objects on the data path (a ``broadcast_object_list`` is a host-synchronising pickle round trip per call; the
    round-4 pipeline issued two of them plus five tensor broadcasts per clip).  Returns (blob, kp_source, kp_driving)."""
    rank = dist.get_rank(group)
    hdev = torch.device("cpu") if staged else device
    if rank == src:
        kv, ksv = kp_driving["value"], kp_source["value"]
        head = torch.tensor([kv.shape[0], kv.shape[1], int("jacobian" in kp_driving), int("jacobian" in kp_source),
                             ksv.shape[0], blob.numel()], dtype=torch.int64, device=hdev)
    else:
        head = torch.zeros(_HEADER_FIELDS, dtype=torch.int64, device=hdev)
    dist.broadcast(head, src=src, group=group)
    T, K, jd, js, S, nblob = (int(v) for v in head.tolist())
    if nblob != blob_numel:
        raise RuntimeError(f"rank {rank}: source cache of {blob_numel} floats here, {nblob} on rank {src} (different configurations?)")
    sizes = [nblob, S * K * 2, S * K * 4 * js, T * K * 2, T * K * 4 * jd]
    if rank == src:
        f32 = lambda t: t.to(device=device, dtype=torch.float32).reshape(-1)
        parts = [blob.reshape(-1), f32(kp_source["value"])]
        if js:
            parts.append(f32(kp_source["jacobian"]))
        parts.append(f32(kp_driving["value"]))
        if jd:
            parts.append(f32(kp_driving["jacobian"]))
        payload = torch.cat(parts)
    else:
        payload = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    _broadcast(payload, src, group, staged)
    blob_o, ksv, ksj, kdv, kdj = torch.split(payload, sizes)
    ks = {"value": ksv.view(S, K, 2)}
    kd = {"value": kdv.view(T, K, 2)}
    if js:
        ks["jacobian"] = ksj.view(S, K, 2, 2)
    if jd:
        kd["jacobian"] = kdj.view(T, K, 2, 2)
    return blob_o, ks, kd


def animate_clip(backend, source_image: Optional[torch.Tensor], kp_source: Optional[Dict[str, torch.Tensor]],
                 kp_driving: Optional[Dict[str, torch.Tensor]], height: int, width: int, uint8: bool = False,
                 group=None, gather: bool = False, kp_driving_initial: Optional[Dict[str, torch.Tensor]] = None,
                 relative: bool = False, adapt_movement_scale: bool = False,
                 emo_driving: Optional[Dict[str, torch.Tensor]] = None, emo_type: str = "linear_3",
                 timings: Optional[Dict[str, float]] = None, to_host: bool = False,
                 replicated: bool = False, before_batch=None) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Animate one clip; returns (frames of this rank's shard, (start, stop)).

    ``emo_driving`` ({'value': [T,E,2], 'jacobian': [T,E,2,2]}, the emotion network's per-frame displacements) adds the
    reference's emotion offsets to the driving key points first (demo.py:263-271, ``--add_emo``); then
    ``kp_driving_initial`` / ``relative`` / ``adapt_movement_scale`` reproduce the loop's ``normalize_kp`` call
    (demo.py:276) -- both for the whole clip at once, on the rank that holds the key points, before the frames are sharded.

    Single process: all T frames.  Under torch.distributed: rank 0 supplies ``source_image`` and the
    key points (other ranks may pass None), every rank returns its contiguous shard; with
    ``gather=True`` rank 0 instead returns all T frames (others an empty tensor).

    ``replicated=True``: every rank already holds the source image and all key points (eamm_amd.animate_from_features with its
    sharded front end): no broadcast at all -- each rank encodes the source itself (0.25 ms at 256x256, less than moving the 5 MB
    cache) and computes its contiguous shard.

    ``before_batch(start, stop)`` (optional) is called before the launch sequence of every batch of frames [start, stop): the
    streaming harness (``animate_from_features``) makes the stream wait there for the event behind which that batch's key points
    are complete, so the front end of later frames runs beside the generator of earlier ones.

    ``to_host=True`` delivers this rank's frames in PINNED HOST memory (what demo.py:281 does per frame with a blocking
    ``.cpu()``): every batch's frames are copied ``non_blocking`` on a copy stream behind an event, so the copy of batch i
    overlaps the kernels of batch i + 1 (the device buffers are the caching allocator's: a batch's buffer is reused only after
    its copy has finished -- ``record_stream``); one synchronisation at the end.  Not combined with ``gather``.

    ``timings`` (a dict, filled in place): wall-clock milliseconds of the phases -- ``keypoints_ms``, ``encode_ms``,
    ``broadcast_ms``, ``compute_ms``, ``gather_ms`` -- with a device synchronisation at each phase boundary (only when
    asked for: the un-instrumented pipeline has no host synchronisation before ``backend.finish()``).
    """
    import time as _time
    _t = [_time.perf_counter()]

    def mark(name):
        if timings is None:
            return
        if torch.device(backend.device).type == "cuda":
            torch.cuda.synchronize(backend.device)
        now = _time.perf_counter()
        timings[name] = timings.get(name, 0.0) + (now - _t[0]) * 1e3
        _t[0] = now

    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    if kp_driving is not None:
        # host-side key-point logic runs where the caller's tensors live; they may sit on different devices
        kdev = kp_driving["value"].device
        on = lambda d: None if d is None else {k: v.to(kdev) for k, v in d.items() if k in ("value", "jacobian")}
        kp_driving, kp_driving_initial, emo_driving = on(kp_driving), on(kp_driving_initial), on(emo_driving)
        if emo_driving is not None:
            from .keypoints import apply_emotion_offsets
            kp_driving = apply_emotion_offsets(kp_driving, emo_driving, emo_type)
    if kp_driving_initial is not None and kp_driving is not None and (relative or adapt_movement_scale):
        from .keypoints import normalize_kp
        kp_source = on(kp_source)
        kp_driving = normalize_kp(kp_source, kp_driving, kp_driving_initial, adapt_movement_scale=adapt_movement_scale,
                                  use_relative_movement=relative, use_relative_jacobian=relative)
    world = dist.get_world_size(group) if distributed else 1
    backend.prepare(height, width)
    mark("keypoints_ms")
    # 1. frame-invariant source tensors: encode once on rank 0, one broadcast
    if distributed and replicated:
        staged = _staged(group, backend.device)
        backend.encode(source_image)
        mark("encode_ms")
        kp_source = {k: v.to(backend.device) for k, v in kp_source.items() if k in ("value", "jacobian")}
        kp_driving = {k: v.to(backend.device) for k, v in kp_driving.items() if k in ("value", "jacobian")}
    elif distributed:
        staged = _staged(group, backend.device)
        blob = backend.encode(source_image) if rank == 0 else None
        mark("encode_ms")
        nblob = backend.blob_numel() if hasattr(backend, "blob_numel") else backend.blob_like().numel()
        blob, kp_source, kp_driving = _broadcast_clip(blob, kp_source, kp_driving, nblob, backend.device, 0, group, staged)
        if rank != 0:
            backend.install(blob.contiguous())
        mark("broadcast_ms")
    else:
        backend.encode(source_image)
        mark("encode_ms")
        kp_source = {k: v.to(backend.device) for k, v in kp_source.items() if k in ("value", "jacobian")}
        kp_driving = {k: v.to(backend.device) for k, v in kp_driving.items() if k in ("value", "jacobian")}
    # 2. this rank's contiguous frame range, `batch` frames per launch sequence
    total = kp_driving["value"].shape[0]
    start, stop = shard_bounds(total, world, rank)
    chunks: List[torch.Tensor] = []
    channels = int(getattr(getattr(backend, "generator", None), "num_channels", 3))   # (test backends without a generator: RGB)
    shape_tail = (height, width, channels) if uint8 else (channels, height, width)
    dtype = torch.uint8 if uint8 else torch.float32
    host = copier = None
    if to_host:
        if distributed and gather:
            raise ValueError("to_host delivers every rank's own shard; it cannot be combined with gather")
        if torch.device(backend.device).type == "cuda":
            host = backend.host_buffer(stop - start, shape_tail, dtype)
            copier = backend.copy_stream()
    for s in range(start, stop, backend.batch):
        e = min(stop, s + backend.batch)
        if before_batch is not None:
            before_batch(s, e)
        out = backend.run({k: v[s:e] for k, v in kp_driving.items()}, kp_source, uint8)
        if host is None:
            chunks.append(out)
            continue
        done = torch.cuda.Event()
        done.record()                                  # on the stream the batch's kernels were enqueued on
        copier.wait_event(done)
        with torch.cuda.stream(copier):
            host[s - start:e - start].copy_(out, non_blocking=True)
        out.record_stream(copier)                      # its memory is reusable only after the copy
    backend.finish()
    if host is not None:
        if timings is not None:
            torch.cuda.current_stream(backend.device).synchronize()
            mark_no_sync = _time.perf_counter()
            timings["compute_ms"] = timings.get("compute_ms", 0.0) + (mark_no_sync - _t[0]) * 1e3
            _t[0] = mark_no_sync
        copier.synchronize()
        if timings is not None:
            now = _time.perf_counter()
            timings["d2h_tail_ms"] = timings.get("d2h_tail_ms", 0.0) + (now - _t[0]) * 1e3   # what the copies add BEHIND the last kernel
            _t[0] = now
        return host, (start, stop)
    local = torch.cat(chunks, dim=0) if chunks else torch.empty((0,) + shape_tail, dtype=dtype, device=backend.device)
    if to_host:                                        # (a CPU stand-in backend: already host memory)
        return local, (start, stop)
    mark("compute_ms")
    if not (distributed and gather):
        return local, (start, stop)
    # 3. optional gather of the shards on rank 0 (padded to the largest shard, then trimmed)
    longest = shard_bounds(total, world, 0)[1]
    cdev = torch.device("cpu") if staged else backend.device
    padded = torch.zeros((longest,) + shape_tail, dtype=dtype, device=cdev)
    padded[: stop - start] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0, group=group)
    mark("gather_ms")
    if rank != 0:
        return local[:0], (start, stop)
    parts = []
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        parts.append(bufs[r][: b - a])
    return torch.cat(parts, dim=0).to(backend.device), (0, total)
