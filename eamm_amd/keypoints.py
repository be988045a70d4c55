"""Key-point normalisation of the clip loop (reference demo.py:112-132, called at demo.py:276).

Host-side logic on tiny tensors (K x 2 values, K x 2 x 2 jacobians): plain torch ops, batched over the
driving frames so a whole clip is normalised at once instead of once per frame.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


def _hull_area(points: torch.Tensor) -> float:
    from scipy.spatial import ConvexHull   # 2-D hull: .volume is the enclosed area (as in the reference)
    return float(ConvexHull(points.detach().cpu().numpy()).volume)


def normalize_kp(kp_source: Dict[str, torch.Tensor], kp_driving: Dict[str, torch.Tensor],
                 kp_driving_initial: Dict[str, torch.Tensor], adapt_movement_scale: bool = False,
                 use_relative_movement: bool = False, use_relative_jacobian: bool = False) -> Dict[str, torch.Tensor]:
    """Same contract as reference demo.py:normalize_kp; ``kp_driving`` may hold T frames ([T,K,2] / [T,K,2,2])
    against one source / initial set ([1,K,2] / [1,K,2,2]).

    * adapt_movement_scale: sqrt(area(hull(source))) / sqrt(area(hull(initial driving)))  (demo.py:114-117)
    * use_relative_movement: value = (driving - initial) * scale + source                  (demo.py:123-126)
    * use_relative_jacobian: jacobian = driving @ inverse(initial) @ source                (demo.py:128-130)
    With both flags off the driving key points pass through unchanged (`relative=False`, demo.py:558).
    """
    scale = 1.0
    if adapt_movement_scale:
        scale = np.sqrt(_hull_area(kp_source["value"][0])) / np.sqrt(_hull_area(kp_driving_initial["value"][0]))
    out = dict(kp_driving)
    if use_relative_movement:
        diff = (kp_driving["value"] - kp_driving_initial["value"]) * scale
        out["value"] = diff + kp_source["value"]
        if use_relative_jacobian:
            jd = torch.matmul(kp_driving["jacobian"], torch.inverse(kp_driving_initial["jacobian"]))
            out["jacobian"] = torch.matmul(jd, kp_source["jacobian"])
    return out
