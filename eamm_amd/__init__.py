"""eamm_amd -- MI355X (gfx950) implementation of EAMM's dense-motion + OcclusionAwareGenerator path.

Python host code (this package) mirrors the reference's module interface for the path and drives
libeamm_hip.so, a C-ABI library of hand-written HIP kernels (include/eamm_hip.h).  See DESIGN.md.
"""
from .config import hot_path_config, tiny_config  # noqa: F401
from .generator import OcclusionAwareGenerator  # noqa: F401
from .engine import Engine  # noqa: F401
from .clip import EngineBackend, animate_clip, driving_keypoints, shard_bounds  # noqa: F401
from .keypoints import apply_emotion_offsets, normalize_kp, one_euro_smooth, smooth_keypoints  # noqa: F401
from .pipeline import animate_from_features  # noqa: F401
from .keypoint_detector import KPDetector, KPDetector_a  # noqa: F401
from .deconv_tail import DeconvTail, SplitFeatureMap  # noqa: F401
from .sync_batchnorm import SynchronizedBatchNorm2d  # noqa: F401
from .data_parallel import all_reduce_gradients  # noqa: F401
from .config import kp_detector_config, kp_detector_a_config, tiny_kp_config  # noqa: F401

__all__ = ["OcclusionAwareGenerator", "Engine", "EngineBackend", "animate_clip", "animate_from_features", "driving_keypoints", "shard_bounds", "normalize_kp", "apply_emotion_offsets", "smooth_keypoints", "one_euro_smooth", "KPDetector", "KPDetector_a", "DeconvTail", "SplitFeatureMap", "SynchronizedBatchNorm2d",
           "all_reduce_gradients", "hot_path_config", "tiny_config"]
