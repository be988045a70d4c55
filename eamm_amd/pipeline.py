"""End-to-end counterpart of the reference's ``make_animation_smooth`` (demo.py:194-282): source image + the audio network's
LSTM features in, the clip's frames in host memory out -- every step of the reference's two loops, batched on the GPU.

    reference (per frame, batch 1, two device round trips per frame)       here (whole clip, on the device)
    ------------------------------------------------------------------     ------------------------------------------------
    kp_source = kp_detector(source)                        demo.py:206     KPDetector, once
    kp_driving_initial = kp_detector_a(deco_out[:, 0])     demo.py:207     frame 0 of the batched front end (un-smoothed)
    for t: kp_detector_a(deco_out[:, t])                   demo.py:212-219 DeconvTail + KPDetector_a, `front_batch` frames per call
           deco_out[:, t] = decon(lstm_out[:, t])          util.py:603-607 (eamm_amd.driving_keypoints)
    OneEuroFilter over emo displacements (.cpu() per frame) demo.py:231-239 eamm_op_one_euro on the device (csrc/keypoints.hip)
    OneEuroFilter over key points (.cpu() per frame)       demo.py:241-250 eamm_op_one_euro
    for t: emotion offsets                                 demo.py:263-271 apply_emotion_offsets, whole clip
           normalize_kp                                    demo.py:276     normalize_kp, whole clip
           generator(source, kp_source, kp_norm)           demo.py:279     animate_clip: source encoded ONCE, 64 frames per call,
           prediction.cpu().numpy() transposed             demo.py:281        uint8 / float frames copied to pinned host memory
                                                                              overlapped with the next batch's kernels

The emotion network (``emo_detector``) and the audio LSTM are outside this path (out of scope / "stays in
PyTorch-ROCm"): their outputs are inputs here -- ``lstm_features`` [T,256] and, optionally, ``emo_driving``.
Under torch.distributed BOTH loops shard by frames: rank 0's inputs are broadcast (header + one payload), every rank runs the
detectors on its frames, one all-gather gives every rank the whole key-point sequence (the One-Euro recurrence needs it; 0.7 ms,
computed redundantly), and every rank animates its shard from its own encoding of the source -- three collectives per clip.
"""
from __future__ import annotations

import time
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .clip import EngineBackend, _broadcast, _staged, animate_clip, driving_keypoints, shard_bounds
from .keypoints import apply_emotion_offsets, normalize_kp, smooth_keypoints

# the reference's two filter parameter sets (demo.py:232-233 and :241-242)
KP_FILTER = dict(mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)
EMO_FILTER = dict(mincutoff=1.0, beta=0.2, dcutoff=1.0, freq=100.0, scale=100.0)


def _kp_only(d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in d.items() if k in ("value", "jacobian")}


@torch.no_grad()
def animate_from_features(generator, kp_detector, deconv_tail, kp_detector_a, source_image: Optional[torch.Tensor],
                          lstm_features: Optional[torch.Tensor], emo_driving: Optional[Dict[str, torch.Tensor]] = None,
                          relative: bool = True, adapt_movement_scale: bool = True, smooth: bool = True, batch: int = 64,
                          front_batch: int = 256, uint8: bool = True, to_host: bool = True, group=None,
                          backend: Optional[EngineBackend] = None, timings: Optional[Dict[str, float]] = None,
                          return_keypoints: bool = False, size: Optional[Tuple[int, int]] = None, stream: Optional[bool] = None):
    """make_animation_smooth (demo.py:194-282) for one clip.

    ``source_image``: [1,C,H,W] float in [0,1]; ``lstm_features``: [T,256] (or [1,T,256]), the audio network's LSTM output;
    ``emo_driving``: None or {'value': [T,E,2], 'jacobian': [T,E,2,2]} (``--add_emo``, type 'linear_3').  ``relative`` /
    ``adapt_movement_scale``: the reference's defaults (True, True).  Returns ``(frames, (start, stop))`` -- this rank's frames
    [n,H,W,3] uint8 (``uint8=True``; what demo.py:507 writes) or [n,3,H,W] float32, in pinned host memory when ``to_host`` --
    plus, with ``return_keypoints``, a dict of the intermediate key points.  Rank > 0 of a process group may pass None for
    the inputs (they are broadcast from rank 0).  ``timings`` is filled with the phases' wall-clock milliseconds (a device synchronisation at each
    boundary, only when asked for): front_ms (detectors), smooth_ms, normalize_ms, then animate_clip's own.  ``stream`` (default:
    on for one process on a GPU when no ``timings`` are asked for): the front end runs on its own stream, `front_batch` frames at
    a time, BESIDE the generator of the frames whose key points are already complete (``_animate_streaming``)."""
    dev = backend.device if backend is not None else next(generator.parameters()).device
    on_gpu = torch.device(dev).type == "cuda"
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    t_last = [time.perf_counter()]

    def mark(name):
        if timings is None:
            return
        if on_gpu:
            torch.cuda.synchronize(dev)
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
        t_last[0] = now

    if backend is None:
        backend = EngineBackend(generator, batch=batch)
    if distributed:
        # every rank gets the clip's INPUTS (source image, LSTM features, emotion displacements: 2.9 MB for 2048 frames at
        # 256x256) in two broadcasts -- fixed header + one payload -- and runs the front end on ITS frames; one all-gather of the
        # raw key points (T x 60 floats) later every rank holds the whole sequence, smooths and normalises it (under 2 ms, redundant)
        # and animates its shard.  Three collectives per clip; nothing of the front end's 23 us per frame stays serial.
        source_image, lstm_features, emo_driving = _broadcast_inputs(source_image, lstm_features, emo_driving, dev, group)
        mark("broadcast_ms")
    src = source_image.to(dev)
    if lstm_features.dim() == 3 and lstm_features.shape[0] == 1:
        lstm_features = lstm_features[0]
    T = lstm_features.shape[0]
    if T < 1:
        raise ValueError("animate_from_features: the clip has no frames (lstm_features is empty)")
    kp_source = _kp_only(kp_detector(src))                                                   # demo.py:206
    H, W = (int(src.shape[-2]), int(src.shape[-1])) if size is None else size
    if stream is None:
        stream = on_gpu and not distributed and timings is None
    if stream and on_gpu and not distributed:
        return _animate_streaming(backend, kp_detector_a, deconv_tail, src, lstm_features.to(dev), kp_source, emo_driving, relative,
                                  adapt_movement_scale, smooth, front_batch, uint8, to_host, return_keypoints, H, W, dev)
    a, b = shard_bounds(T, world, rank)
    raw = driving_keypoints(deconv_tail, kp_detector_a, lstm_features[a:b].to(dev), batch=front_batch) if b > a else None   # demo.py:212-219
    if distributed:
        raw = _all_gather_keypoints(raw, T, kp_source["value"].shape[1], dev, group)
    kp_initial = {k: v[:1].clone() for k, v in raw.items()}                                  # demo.py:207 (never smoothed)
    mark("front_ms")
    kp_d = smooth_keypoints(raw, **KP_FILTER) if smooth else raw                             # demo.py:241-250
    if emo_driving is not None:
        emo = {k: v.to(dev) for k, v in _kp_only(emo_driving).items()}
        emo = smooth_keypoints(emo, **EMO_FILTER) if smooth else emo                         # demo.py:231-239
        kp_d = apply_emotion_offsets(kp_d, emo)                                              # demo.py:263-271
    mark("smooth_ms")
    kp_norm = normalize_kp(kp_source, kp_d, kp_initial, adapt_movement_scale=adapt_movement_scale,
                           use_relative_movement=relative, use_relative_jacobian=relative)    # demo.py:276
    mark("normalize_ms")
    kps = {"kp_source": kp_source, "kp_driving_raw": raw, "kp_driving_smoothed": kp_d, "kp_norm": kp_norm} if return_keypoints else {}
    frames, span = animate_clip(backend, src, kp_source, kp_norm, H, W, uint8=uint8, group=group, to_host=to_host,
                                timings=timings, replicated=distributed)
    return (frames, span, kps) if return_keypoints else (frames, span)


def _animate_streaming(backend, kp_detector_a, deconv_tail, src, feats, kp_source, emo_driving, relative, adapt_movement_scale, smooth,
                       front_batch, uint8, to_host, return_keypoints, H, W, dev):
    """One GPU, one process: the reference's FIRST loop (key points of every frame, demo.py:212-250) runs on its own stream,
    `front_batch` frames at a time, and the generator starts on the first frames as soon as their key points are complete --
    every step between the detector and the generator is causal (the One-Euro filter carries its memory from chunk to chunk:
    ``eamm_op_one_euro`` with ``state`` / ``resume``; ``normalize_kp`` needs frame 0 and the source only), so the front end of
    frames t >= front_batch is hidden behind the generator of the frames before.  Bit-identical to the un-streamed path: the same
    kernels on the same batches, the filter resumed instead of run whole."""
    from .keypoints import movement_scale
    T, K = feats.shape[0], kp_source["value"].shape[1]
    main = torch.cuda.current_stream(dev)
    front = getattr(backend, "_front_stream", None)
    if front is None:      # (small launches: a high-priority stream lets them in ahead of the generator's)
        front = backend._front_stream = torch.cuda.Stream(device=dev, priority=-1)
    norm = {"value": torch.empty(T, K, 2, device=dev), "jacobian": torch.empty(T, K, 2, 2, device=dev)}
    keep = {"raw": [], "smoothed": []} if return_keypoints else None
    emo = None if emo_driving is None else {k: v.to(dev) for k, v in _kp_only(emo_driving).items()}
    front.wait_stream(main)                        # the source's key points and the features are complete
    events, kp_state, emo_state, kp_initial, scale = [], {}, {}, None, None
    with torch.cuda.stream(front):
        for t0 in range(0, T, front_batch):
            t1 = min(T, t0 + front_batch)
            raw = driving_keypoints(deconv_tail, kp_detector_a, feats[t0:t1], batch=front_batch)          # demo.py:212-219
            if t0 == 0:
                kp_initial = {k: v[:1].clone() for k, v in raw.items()}                                       # demo.py:207
                if adapt_movement_scale:
                    scale = movement_scale(kp_source, kp_initial)        # the clip's one host read (2 x K points), behind the first batch
            kp_d = smooth_keypoints(raw, **KP_FILTER, state=kp_state, resume=t0 > 0) if smooth else raw       # demo.py:241-250
            if emo is not None:
                e_c = {k: v[t0:t1] for k, v in emo.items()}
                e_c = smooth_keypoints(e_c, **EMO_FILTER, state=emo_state, resume=t0 > 0) if smooth else e_c  # demo.py:231-239
                kp_d = apply_emotion_offsets(kp_d, e_c)                                                       # demo.py:263-271
            n_c = normalize_kp(kp_source, kp_d, kp_initial, adapt_movement_scale=adapt_movement_scale, use_relative_movement=relative,
                               use_relative_jacobian=relative, scale=scale)                                   # demo.py:276
            for k in ("value", "jacobian"):
                norm[k][t0:t1].copy_(n_c[k])
            if keep is not None:
                keep["raw"].append(raw)
                keep["smoothed"].append(kp_d)
            ev = torch.cuda.Event()
            ev.record(front)
            events.append(ev)

    def before_batch(s, e):        # the generator's stream waits for the key points of frames < e
        main.wait_event(events[(e - 1) // front_batch])

    frames, span = animate_clip(backend, src, kp_source, norm, H, W, uint8=uint8, to_host=to_host, before_batch=before_batch)
    main.wait_stream(front)
    if not return_keypoints:
        return frames, span
    # (concatenated on the caller's stream: the results belong to its allocator pool, the front stream's chunks die here, after the join)
    cat = lambda parts: {k: torch.cat([p[k] for p in parts]) for k in ("value", "jacobian")}
    return frames, span, {"kp_source": kp_source, "kp_driving_raw": cat(keep["raw"]), "kp_driving_smoothed": cat(keep["smoothed"]), "kp_norm": norm}


_IN_HEADER = 8   # T, feature channels, emotion points E (0: none), image channels, H, W, features given as [1,T,C], reserved


def _broadcast_inputs(source_image, lstm_features, emo_driving, dev, group):
    """Rank 0's clip inputs to every rank: a fixed eight-integer header, then ONE float32 payload
    source | features | emo value | emo jacobian."""
    rank = dist.get_rank(group)
    staged = _staged(group, dev)
    hdev = torch.device("cpu") if staged else dev
    if rank == 0:
        feats = lstm_features[0] if (lstm_features.dim() == 3 and lstm_features.shape[0] == 1) else lstm_features
        E = 0 if emo_driving is None else int(emo_driving["value"].shape[1])
        head = torch.tensor([feats.shape[0], feats.shape[1], E, source_image.shape[1], source_image.shape[2], source_image.shape[3], 0, 0],
                            dtype=torch.int64, device=hdev)
    else:
        head = torch.zeros(_IN_HEADER, dtype=torch.int64, device=hdev)
    dist.broadcast(head, src=0, group=group)
    T, C, E, ch, H, W, _, _ = (int(v) for v in head.tolist())
    sizes = [ch * H * W, T * C, T * E * 2, T * E * 4]
    if rank == 0:
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).reshape(-1)
        parts = [f32(source_image), f32(feats)]
        if E:
            parts += [f32(emo_driving["value"]), f32(emo_driving["jacobian"])]
        payload = torch.cat(parts)
    else:
        payload = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    _broadcast(payload, 0, group, staged)
    s_img, s_feat, s_ev, s_ej = torch.split(payload, sizes)
    emo = {"value": s_ev.view(T, E, 2), "jacobian": s_ej.view(T, E, 2, 2)} if E else None
    return s_img.view(1, ch, H, W), s_feat.view(T, C), emo


def _all_gather_keypoints(local, T: int, K: int, dev, group):
    """Every rank's raw key points of its contiguous frame shard -> the whole clip's on every rank: ONE all-gather of
    [longest shard, K * 6] floats per rank (shards padded to the longest, trimmed after)."""
    world, staged = dist.get_world_size(group), _staged(group, dev)
    longest = shard_bounds(T, world, 0)[1]
    cdev = torch.device("cpu") if staged else dev
    mine = torch.zeros(longest, K * 6, dtype=torch.float32, device=cdev)
    if local is not None:
        n = local["value"].shape[0]
        mine[:n, :K * 2] = local["value"].reshape(n, -1).to(cdev)
        mine[:n, K * 2:] = local["jacobian"].reshape(n, -1).to(cdev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    rows = torch.cat([bufs[r][: shard_bounds(T, world, r)[1] - shard_bounds(T, world, r)[0]] for r in range(world)]).to(dev)
    return {"value": rows[:, :K * 2].reshape(T, K, 2).contiguous(), "jacobian": rows[:, K * 2:].reshape(T, K, 2, 2).contiguous()}
