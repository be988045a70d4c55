"""ctypes binding of libeamm_hip.so (C ABI declared in include/eamm_hip.h).

There is deliberately NO fallback: if the HIP library has not been built, importing a symbol from
here raises, and nothing in ``eamm_amd`` computes on the CPU or through stock PyTorch ops.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeamm_hip.so")

ABI_VERSION = 5   # round 6: split NHWC hand-over DeconvTail -> KPDetector_a (eamm_deconv_forward_split, eamm_kp_detect_features_split)

EAMM_OK = 0
ERR_ARG, ERR_STATE, ERR_KEY, ERR_HIP, ERR_NUMERIC = -1, -2, -3, -4, -5


class EammConfig(C.Structure):
    _fields_ = [
        ("num_channels", C.c_int32), ("num_kp", C.c_int32), ("block_expansion", C.c_int32),
        ("max_features", C.c_int32), ("num_down_blocks", C.c_int32), ("num_bottleneck_blocks", C.c_int32),
        ("estimate_occlusion_map", C.c_int32), ("dm_block_expansion", C.c_int32),
        ("dm_max_features", C.c_int32), ("dm_num_blocks", C.c_int32), ("dm_inv_scale", C.c_int32),
        ("kp_variance", C.c_float), ("height", C.c_int32), ("width", C.c_int32),
        ("max_frames", C.c_int32), ("max_sources", C.c_int32),
    ]


class EammOutputs(C.Structure):
    _fields_ = [
        ("prediction", C.c_void_p), ("mask", C.c_void_p), ("sparse_deformed", C.c_void_p),
        ("occlusion_map", C.c_void_p), ("deformed", C.c_void_p), ("deformation", C.c_void_p),
        ("frames_u8", C.c_void_p),
    ]


class EammKpConfig(C.Structure):
    _fields_ = [
        ("num_kp", C.c_int32), ("num_channels", C.c_int32), ("in_features", C.c_int32),
        ("block_expansion", C.c_int32), ("max_features", C.c_int32), ("num_blocks", C.c_int32),
        ("temperature", C.c_float), ("estimate_jacobian", C.c_int32), ("single_jacobian_map", C.c_int32),
        ("inv_scale", C.c_int32), ("pad", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("max_batch", C.c_int32), ("with_predictor", C.c_int32),
    ]


class EammDeconvConfig(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("channels", C.c_int32 * 9), ("max_batch", C.c_int32)]


class EammBnSite(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


class EammKpOutputs(C.Structure):
    _fields_ = [("value", C.c_void_p), ("jacobian", C.c_void_p), ("heatmap", C.c_void_p)]


# name -> (restype, argtypes); kept in one table so tests can check that every symbol declared in
# include/eamm_hip.h is exported and bound.
SIGNATURES = {
    "eamm_abi_version": (C.c_int, []),
    "eamm_create": (C.c_int, [C.POINTER(EammConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "eamm_destroy": (None, [C.c_void_p]),
    "eamm_last_error": (C.c_char_p, [C.c_void_p]),
    "eamm_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "eamm_finalize_weights": (C.c_int, [C.c_void_p]),
    "eamm_encode_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eamm_forward_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(EammOutputs), C.c_void_p]),
    "eamm_check_numeric": (C.c_int, [C.c_void_p, C.c_void_p]),
    "eamm_source_cache_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "eamm_export_source_cache": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eamm_import_source_cache": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eamm_flops_per_frame": (C.c_double, [C.c_void_p]),
    "eamm_bottleneck_form": (C.c_int, [C.c_void_p, C.c_int]),
    "eamm_encode_flops": (C.c_double, [C.c_void_p]),
    "eamm_bottleneck_chains": (C.c_int, [C.c_void_p, C.c_int]),
    "eamm_pass_chains": (C.c_int, [C.c_void_p, C.c_int]),
    "eamm_op_one_euro": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "eamm_last_stream_set": (C.c_int, [C.c_void_p]),
    "eamm_describe_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "eamm_knobs_json": (C.c_int, [C.c_char_p, C.c_int]),
    "eamm_build_experiments": (C.c_int, []),
    "eamm_total_mfma_flops": (C.c_double, []),
    "eamm_kp_create": (C.c_int, [C.POINTER(EammKpConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "eamm_kp_destroy": (None, [C.c_void_p]),
    "eamm_kp_last_error": (C.c_char_p, [C.c_void_p]),
    "eamm_kp_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "eamm_kp_finalize_weights": (C.c_int, [C.c_void_p]),
    "eamm_kp_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(EammKpOutputs), C.c_void_p]),
    "eamm_kp_detect_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(EammKpOutputs), C.c_void_p]),
    "eamm_kp_split_channels": (C.c_int, [C.c_void_p]),
    "eamm_kp_detect_features_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(EammKpOutputs), C.c_void_p]),
    "eamm_deconv_create": (C.c_int, [C.POINTER(EammDeconvConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "eamm_deconv_destroy": (None, [C.c_void_p]),
    "eamm_deconv_last_error": (C.c_char_p, [C.c_void_p]),
    "eamm_deconv_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "eamm_deconv_finalize_weights": (C.c_int, [C.c_void_p]),
    "eamm_deconv_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "eamm_deconv_split_channels": (C.c_int, [C.c_void_p]),
    "eamm_deconv_forward_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "eamm_bn_local_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_backward_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "eamm_bn_backward_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_backward_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "eamm_bn_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p]),
    "eamm_bn_last_error": (C.c_char_p, []),
    "eamm_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "eamm_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.c_int]),
    "eamm_op_conv": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "eamm_set_training": (C.c_int, [C.c_void_p, C.c_int]),
    "eamm_train_num_sites": (C.c_int, [C.c_void_p]),
    "eamm_train_site_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "eamm_train_max_channels": (C.c_int, [C.c_void_p]),
    "eamm_train_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(EammBnSite), C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.POINTER(EammOutputs),
                                   C.c_void_p]),
    "eamm_train_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "eamm_op_warp": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "eamm_op_warp_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_op_antialias_down": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eamm_op_antialias_down_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eamm_op_kp_records": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_op_kp_records_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_op_motion_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "eamm_op_motion_front": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_op_motion_front_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_motion_head": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_op_motion_head_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.c_void_p]),
    "eamm_bn_nhwc_workspace_floats": (C.c_size_t, [C.c_longlong, C.c_int]),
    "eamm_bn_nhwc_local_sums": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_nhwc_local_stats": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_nhwc_backward_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_nhwc_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "eamm_bn_nhwc_backward_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eamm_bn_nhwc_backward_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eamm_op_conv_dev_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "eamm_op_conv_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_conv7_thin_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "eamm_op_conv7_thin": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_conv7_thin_wgrad": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_final_conv_sigmoid": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_conv_wgrad_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "eamm_op_conv_wgrad": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eamm_op_conv_saved_transform_offset": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "eamm_op_conv_wgrad_saved": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


class EammError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libeamm_hip error {code}: {message}")
        self.code = code


def lib() -> C.CDLL:
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C eamm_amd/csrc`). "
            "eamm_amd has no CPU or PyTorch fallback for this path.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = symbol missing from the .so
        fn.restype = res
        fn.argtypes = args
    got = handle.eamm_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libeamm_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild the extension")
    _lib = handle
    return _lib


def check(code: int, ctx=None, kp: bool = False, deconv: bool = False):
    if code != EAMM_OK:
        msg = (lib().eamm_deconv_last_error if deconv else lib().eamm_kp_last_error if kp else lib().eamm_last_error)(ctx)
        raise EammError(code, msg.decode() if msg else "?")
