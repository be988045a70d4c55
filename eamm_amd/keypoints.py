"""Key-point normalisation of the clip loop (reference demo.py:112-132, called at demo.py:276).

Host-side logic on tiny tensors (K x 2 values, K x 2 x 2 jacobians): plain torch ops, batched over the
driving frames so a whole clip is normalised at once instead of once per frame.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


def _hull_area(points: torch.Tensor) -> float:
    from scipy.spatial import ConvexHull   # 2-D hull: .volume is the enclosed area (as in the reference)
    return float(ConvexHull(points.detach().cpu().numpy()).volume)


def _inverse_2x2(m: torch.Tensor) -> torch.Tensor:
    """Inverse of [...,2,2] matrices in closed form (adjugate / determinant) -- the reference calls ``torch.inverse``
    (demo.py:128), which on a GPU tensor is a solver-library call (handle creation + LU) for ten 2x2 matrices; the closed form
    is a handful of element-wise operations where the tensors live and agrees with the LU result to rounding.  A singular
    matrix gives inf / nan here where torch.inverse raises; the generator's own check (eamm_check_numeric) still raises."""
    if m.shape[-2:] != (2, 2):
        return torch.inverse(m)
    a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
    det = a * d - b * c
    return torch.stack([torch.stack([d, -b], -1), torch.stack([-c, a], -1)], -2) / det[..., None, None]


def movement_scale(kp_source: Dict[str, torch.Tensor], kp_driving_initial: Dict[str, torch.Tensor]) -> float:
    """sqrt(area(hull(source))) / sqrt(area(hull(initial driving)))  (demo.py:114-117); one host read of 2 x K points."""
    return float(np.sqrt(_hull_area(kp_source["value"][0])) / np.sqrt(_hull_area(kp_driving_initial["value"][0])))


def normalize_kp(kp_source: Dict[str, torch.Tensor], kp_driving: Dict[str, torch.Tensor],
                 kp_driving_initial: Dict[str, torch.Tensor], adapt_movement_scale: bool = False,
                 use_relative_movement: bool = False, use_relative_jacobian: bool = False,
                 scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Same contract as reference demo.py:normalize_kp; ``kp_driving`` may hold T frames ([T,K,2] / [T,K,2,2])
    against one source / initial set ([1,K,2] / [1,K,2,2]).

    * adapt_movement_scale: sqrt(area(hull(source))) / sqrt(area(hull(initial driving)))  (demo.py:114-117)
    * use_relative_movement: value = (driving - initial) * scale + source                  (demo.py:123-126)
    * use_relative_jacobian: jacobian = driving @ inverse(initial) @ source                (demo.py:128-130)
    With both flags off the driving key points pass through unchanged (`relative=False`, demo.py:558).  ``scale``: the
    adapt_movement_scale factor computed earlier by ``movement_scale`` (then the hulls are not read again).
    """
    if scale is not None:                  # computed before (movement_scale): a clip normalised chunk by chunk reads the hulls once
        scale = float(scale)
    elif adapt_movement_scale:
        scale = np.sqrt(_hull_area(kp_source["value"][0])) / np.sqrt(_hull_area(kp_driving_initial["value"][0]))
    else:
        scale = 1.0
    out = dict(kp_driving)
    if use_relative_movement:
        diff = (kp_driving["value"] - kp_driving_initial["value"]) * scale
        out["value"] = diff + kp_source["value"]
        if use_relative_jacobian:
            jd = torch.matmul(kp_driving["jacobian"], _inverse_2x2(kp_driving_initial["jacobian"]))
            out["jacobian"] = torch.matmul(jd, kp_source["jacobian"])
    return out


# demo.py:263-271 (`--add_emo`, type 'linear_3'): key point k of the driving pose receives weight * offset e of the
# emotion network's displacement, for value and jacobian alike -- (k, e, weight)
EMOTION_OFFSETS_LINEAR_3 = ((1, 0, 0.2), (4, 1, 1.0), (6, 2, 1.0))


def apply_emotion_offsets(kp_driving: Dict[str, torch.Tensor], emo_driving: Dict[str, torch.Tensor],
                          kind: str = "linear_3") -> Dict[str, torch.Tensor]:
    """Emotion displacement of the driving key points, the step between ``kp_driving_all[t]`` and ``normalize_kp`` in
    the reference's second loop (demo.py:263-271), for a whole clip at once: ``kp_driving`` holds [T,K,2] / [T,K,2,2],
    ``emo_driving`` the emotion network's [T,E,2] / [T,E,2,2] (E >= 3).  Same float32 operations per element as the
    reference (``v[:,1] + e[:,0]*0.2``; a weight of 1 is the plain sum the reference writes).  Returns new tensors;
    the reference edits ``kp_driving`` in place, which no later step observes."""
    if kind != "linear_3":
        raise ValueError(f"unknown emotion offset type {kind!r} (the reference implements 'linear_3' only, demo.py:265)")
    out = dict(kp_driving)
    for key in ("value", "jacobian"):
        if key not in kp_driving:
            continue
        if key not in emo_driving:
            raise KeyError(key)
        v, e = kp_driving[key].clone(), emo_driving[key].to(kp_driving[key].device)
        if e.shape[0] != v.shape[0] or e.shape[1] < 3 or e.shape[2:] != v.shape[2:]:
            raise RuntimeError(f"emo_driving[{key!r}] has shape {tuple(e.shape)}, expected [{v.shape[0]},>=3,...]")
        for k, j, wgt in EMOTION_OFFSETS_LINEAR_3:
            v[:, k] = v[:, k] + (e[:, j] * wgt if wgt != 1.0 else e[:, j])
        out[key] = v
    return out


def one_euro_smooth(seq: torch.Tensor, mincutoff: float = 1.0, beta: float = 0.0, dcutoff: float = 1.0,
                    freq: float = 30.0, scale: float = 1.0, state: Optional[torch.Tensor] = None, resume: bool = False) -> torch.Tensor:
    """One-Euro low-pass filter along dim 0 of ``seq`` ([T, ...]), element-wise over the rest -- the reference's
    ``filter1.OneEuroFilter`` (filter1.py:13-47) applied as ``process(x * scale) / scale`` frame after frame
    (demo.py:237-250).  The recurrence is sequential in T and independent per element.

    ``state`` (GPU tensors only): a float32 [3, E] tensor on ``seq``'s device that receives the filter's memory after the last
    frame; with ``resume`` it is read first, so chunks of a clip filtered one after the other give exactly the clip filtered
    whole (the filter is causal: ``animate_from_features`` streams a clip through it).

    A tensor on the GPU is filtered there by ``eamm_op_one_euro`` (csrc/keypoints.hip: one thread per element walks the
    frames; a 2048-frame clip takes ~0.15 ms, no host round trip) -- the clip pipeline's path.  A CPU tensor is filtered on
    the host in float32 in the reference's operation order (numpy; the reference itself filters on the host)."""
    if seq.is_cuda:
        import ctypes as C
        from . import _lib
        x = seq.detach().to(torch.float32).contiguous()
        T = x.shape[0]
        out = torch.empty_like(x)
        if T:
            E = x.numel() // T
            if state is not None and (state.device != x.device or state.dtype != torch.float32 or state.numel() != 3 * E
                                      or not state.is_contiguous()):
                raise RuntimeError(f"one_euro_smooth: state must be a contiguous float32 [3,{E}] tensor on {x.device}")
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().eamm_op_one_euro(x.device.index, C.c_void_p(x.data_ptr()), T, E, float(mincutoff),
                                                       float(beta), float(dcutoff), float(freq), float(scale), C.c_void_p(out.data_ptr()),
                                                       None if state is None else C.c_void_p(state.data_ptr()), int(bool(resume)),
                                                       C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), None)
        return out
    if state is not None or resume:
        raise RuntimeError("one_euro_smooth: the resumable form is the device path's (a CPU tensor is filtered whole)")
    f32 = np.float32
    if seq.shape[0] == 0:                       # an empty sequence (an empty shard): nothing to filter
        return torch.empty(seq.shape, dtype=torch.float32)
    per_frame = seq[0].numel()
    x_all = seq.detach().to(torch.float32).contiguous().numpy().reshape(seq.shape[0], per_frame) * f32(scale)
    out = np.empty_like(x_all)
    # The reference filters CPU tensors (demo.py:245 `.cpu() * 10 ... / 10`): ATen's float32 CPU kernels DIVIDE (`tau / te`,
    # `/ scale` are true divisions by the scalar cast to float32; `1.0 / tensor` is reciprocal(tensor)), unlike the CUDA kernels,
    # which multiply by the scalar's reciprocal.  Same operations here -> bit-equal to the reference filter's outputs.
    te = 1.0 / freq
    a_dd = 1.0 / (1.0 + (1.0 / (2 * np.pi * dcutoff)) / te)
    a_d, one_m_ad = f32(a_dd), f32(1.0 - a_dd)
    two_pi, te32, one, fq, mc, bt, sc = f32(2 * np.pi), f32(te), f32(1.0), f32(freq), f32(mincutoff), f32(beta), f32(scale)
    prev_x = prev_s = prev_edx = None
    for t in range(x_all.shape[0]):
        x = x_all[t]
        if prev_x is None:                      # first sample: dx = 0, both low-pass filters pass their input through
            edx = np.zeros_like(x)
            s = x
        else:
            dx = (x - prev_x) * fq
            edx = a_d * dx + one_m_ad * prev_edx
            tau = one / ((mc + bt * np.abs(edx)) * two_pi)
            a = one / (one + tau / te32)
            s = a * x + (one - a) * prev_s
        prev_x, prev_s, prev_edx = x, s, edx
        out[t] = s
    return torch.from_numpy(out / sc).reshape(seq.shape)


def smooth_keypoints(kp_seq: Dict[str, torch.Tensor], mincutoff: float = 0.05, beta: float = 8.0, dcutoff: float = 1.0,
                     freq: float = 100.0, scale: float = 10.0, state: Optional[Dict[str, torch.Tensor]] = None,
                     resume: bool = False) -> Dict[str, torch.Tensor]:
    """Temporal smoothing of a clip's driving key points, defaults = the reference's (demo.py:241-250: one filter for
    the values, one for the jacobians, inputs scaled by 10).  ``kp_seq``: {'value': [T,K,2], 'jacobian': [T,K,2,2]}."""
    if state is not None:      # chunked clip: one memory tensor per key, created on first use
        for k, v in kp_seq.items():
            if k in ("value", "jacobian") and k not in state:
                state[k] = torch.zeros(3, v.numel() // max(1, v.shape[0]), dtype=torch.float32, device=v.device)
    # (keys other than 'value' / 'jacobian' -- the detectors' 'heatmap' -- are not filtered by the reference either, demo.py:244-248:
    #  they are passed through untouched rather than dropped)
    return {k: (one_euro_smooth(v, mincutoff, beta, dcutoff, freq, scale, None if state is None else state[k], resume)
                if k in ("value", "jacobian") else v) for k, v in kp_seq.items()}
