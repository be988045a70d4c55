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
Under torch.distributed the front end runs on rank 0 and the generator's frames are sharded over the ranks (animate_clip).
"""
from __future__ import annotations

import time
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .clip import EngineBackend, animate_clip, driving_keypoints
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
                          front_batch: int = 64, uint8: bool = True, to_host: bool = True, group=None,
                          backend: Optional[EngineBackend] = None, timings: Optional[Dict[str, float]] = None,
                          return_keypoints: bool = False, size: Optional[Tuple[int, int]] = None):
    """make_animation_smooth (demo.py:194-282) for one clip.

    ``source_image``: [1,C,H,W] float in [0,1]; ``lstm_features``: [T,256] (or [1,T,256]), the audio network's LSTM output;
    ``emo_driving``: None or {'value': [T,E,2], 'jacobian': [T,E,2,2]} (``--add_emo``, type 'linear_3').  ``relative`` /
    ``adapt_movement_scale``: the reference's defaults (True, True).  Returns ``(frames, (start, stop))`` -- this rank's frames
    [n,H,W,3] uint8 (``uint8=True``; what demo.py:507 writes) or [n,3,H,W] float32, in pinned host memory when ``to_host`` --
    plus, with ``return_keypoints``, a dict of the intermediate key points (rank 0).  Rank > 0 of a process group may pass
    None for the inputs and then gives the frame ``size`` (H, W) instead (no collective is spent on it).  ``timings`` is filled with the phases' wall-clock milliseconds (a device synchronisation at each
    boundary, only when asked for): front_ms (detectors), smooth_ms, normalize_ms, then animate_clip's own."""
    dev = next(generator.parameters()).device
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    t_last = [time.perf_counter()]

    def mark(name):
        if timings is None:
            return
        torch.cuda.synchronize(dev)
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
        t_last[0] = now

    kps: Dict[str, Dict[str, torch.Tensor]] = {}
    kp_source = kp_norm = None
    if rank == 0:
        src = source_image.to(dev)
        kp_source = _kp_only(kp_detector(src))                                              # demo.py:206
        raw = driving_keypoints(deconv_tail, kp_detector_a, lstm_features.to(dev), batch=front_batch)   # demo.py:212-219
        kp_initial = {k: v[:1].clone() for k, v in raw.items()}                             # demo.py:207 (never smoothed)
        mark("front_ms")
        kp_d = smooth_keypoints(raw, **KP_FILTER) if smooth else raw                        # demo.py:241-250
        if emo_driving is not None:
            emo = {k: v.to(dev) for k, v in _kp_only(emo_driving).items()}
            emo = smooth_keypoints(emo, **EMO_FILTER) if smooth else emo                    # demo.py:231-239
            kp_d = apply_emotion_offsets(kp_d, emo)                                         # demo.py:263-271
        mark("smooth_ms")
        kp_norm = normalize_kp(kp_source, kp_d, kp_initial, adapt_movement_scale=adapt_movement_scale,
                               use_relative_movement=relative, use_relative_jacobian=relative)   # demo.py:276
        mark("normalize_ms")
        if return_keypoints:
            kps = {"kp_source": kp_source, "kp_driving_raw": raw, "kp_driving_smoothed": kp_d, "kp_norm": kp_norm}
    else:
        src = None
    if backend is None:
        backend = EngineBackend(generator, batch=batch)
    if size is None:
        if source_image is None:
            raise ValueError("a rank without the source image must be given the frame size (H, W)")
        size = (int(source_image.shape[-2]), int(source_image.shape[-1]))
    H, W = size
    frames, span = animate_clip(backend, src, kp_source, kp_norm, H, W, uint8=uint8, group=group, to_host=to_host,
                                timings=timings)
    return (frames, span, kps) if return_keypoints else (frames, span)
