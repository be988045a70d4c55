"""Hot-path configuration of the dense-motion + OcclusionAwareGenerator path.

The reference builds the generator by splatting two YAML sections into the constructor
(reference demo.py:54-55, run.py:60-61):

    OcclusionAwareGenerator(**model_params.generator_params, **model_params.common_params)

``hot_path_config()`` is the literal content of
reference config/MEAD_emo_video_aug_delta_4_crop_random_crop.yaml:26-52 for those two sections;
``tiny_config()`` is a structurally identical but small network used for full-tensor golden
fixtures and fast CPU tests.
"""
from __future__ import annotations

import copy

_HOT = {
    "generator_params": {
        "block_expansion": 64,
        "max_features": 512,
        "num_down_blocks": 2,
        "num_bottleneck_blocks": 6,
        "estimate_occlusion_map": True,
        "dense_motion_params": {
            "block_expansion": 64,
            "max_features": 1024,
            "num_blocks": 5,
            "scale_factor": 0.25,
        },
    },
    "common_params": {"num_kp": 10, "num_channels": 3, "estimate_jacobian": True},
}

_TINY = {
    "generator_params": {
        "block_expansion": 32,
        "max_features": 128,
        "num_down_blocks": 2,
        "num_bottleneck_blocks": 2,
        "estimate_occlusion_map": True,
        "dense_motion_params": {
            "block_expansion": 32,
            "max_features": 128,
            "num_blocks": 3,
            "scale_factor": 0.25,
        },
    },
    "common_params": {"num_kp": 10, "num_channels": 3, "estimate_jacobian": True},
}


def hot_path_config() -> dict:
    """Constructor kwargs of the full-size generator (flat dict, ready to splat)."""
    c = copy.deepcopy(_HOT)
    return {**c["generator_params"], **c["common_params"]}


def tiny_config() -> dict:
    """Small generator with the same topology (64x64 frames, 16x16 motion grid)."""
    c = copy.deepcopy(_TINY)
    return {**c["generator_params"], **c["common_params"]}


# key-point detectors (N1): reference config/MEAD_emo_video_aug_delta_4_crop_random_crop.yaml:31-41,
# built as KPDetector(**kp_detector_params, **common_params) / KPDetector_a(**kp_detector_params, **audio_params)
# (reference demo.py:59-72)
_KP = {"temperature": 0.1, "block_expansion": 32, "max_features": 1024, "scale_factor": 0.25, "num_blocks": 5}


def kp_detector_config() -> dict:
    return {**_KP, "num_kp": 10, "num_channels": 3, "estimate_jacobian": True}


def kp_detector_a_config() -> dict:
    return {**_KP, "num_kp": 10, "num_channels": 3, "num_channels_a": 3, "estimate_jacobian": True}


def tiny_kp_config(audio: bool = False) -> dict:
    c = {"temperature": 0.1, "block_expansion": 32, "max_features": 128, "scale_factor": 0.25, "num_blocks": 3,
         "num_kp": 10, "num_channels": 3, "estimate_jacobian": True}
    if audio:
        c["num_channels_a"] = 3
    return c
