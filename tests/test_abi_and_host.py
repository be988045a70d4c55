"""CPU-side checks: the C-ABI library loads and exports what include/eamm_hip.h declares, the host
module mirrors the reference's checkpoint layout, and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from eamm_amd import OcclusionAwareGenerator, hot_path_config, shard_bounds, tiny_config
from eamm_amd import _lib
from eamm_amd.engine import config_struct
from eamm_amd.weights import (antialias_kernel, state_dict_spec, synthetic_keypoints, synthetic_source,
                              synthetic_state_dict)

HEADER = os.path.join(ROOT, "include", "eamm_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eamm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert "eamm_forward_frames" in names and "eamm_encode_source" in names and len(names) >= 14
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in eamm_hip.h but not exported by libeamm_hip.so"
    # and the ctypes binding covers exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().eamm_abi_version() == _lib.ABI_VERSION


def test_struct_layout_matches_header():
    # 12 int32 + float + 4 int32, no padding; 7 pointers
    assert ctypes.sizeof(_lib.EammConfig) == 16 * 4
    assert ctypes.sizeof(_lib.EammOutputs) == 7 * ctypes.sizeof(ctypes.c_void_p)
    fields = [f[0] for f in _lib.EammConfig._fields_]
    text = open(HEADER).read()
    body = text[text.index("typedef struct eamm_config {"):text.index("} eamm_config;")]
    in_header = re.findall(r"\b(?:int32_t|float)\s+([a-z_, ]+);", body)
    flat = [x.strip() for grp in in_header for x in grp.split(",")]
    assert flat == fields


def test_create_rejects_bad_config_without_gpu_compute():
    L = _lib.lib()
    cs = config_struct(tiny_config(), 64, 64, 4, 1)
    cs.num_kp = 40
    ctx = ctypes.c_void_p()
    rc = L.eamm_create(ctypes.byref(cs), 0, ctypes.byref(ctx))
    assert rc == _lib.ERR_ARG and not ctx.value
    assert b"num_kp" in L.eamm_last_error(None)
    cs = config_struct(tiny_config(), 72, 64, 4, 1)  # 72 not divisible by 4 * 2^3
    assert L.eamm_create(ctypes.byref(cs), 0, ctypes.byref(ctx)) == _lib.ERR_ARG
    cs = config_struct(hot_path_config(), 512, 512, 256, 1)  # 256 frames of 512x512: a > 4 GiB activation tensor
    assert L.eamm_create(ctypes.byref(cs), 0, ctypes.byref(ctx)) == _lib.ERR_ARG
    assert b"4 GiB" in L.eamm_last_error(None)
    cs = config_struct({**tiny_config(), "num_channels": 7}, 64, 64, 4, 1)  # one to six image channels (include/eamm_hip.h)
    assert L.eamm_create(ctypes.byref(cs), 0, ctypes.byref(ctx)) == _lib.ERR_ARG
    assert b"num_channels must be 1 .. 6" in L.eamm_last_error(None)


@pytest.mark.parametrize("cfg_fn,nkeys,nparams", [(tiny_config, 112, None), (hot_path_config, 196, 45593205)])
def test_module_state_dict_layout(cfg_fn, nkeys, nparams):
    cfg = cfg_fn()
    gen = OcclusionAwareGenerator(**cfg)
    sd = gen.state_dict()
    spec = state_dict_spec(cfg)
    assert [k for k, *_ in spec] and sorted(sd) == sorted(k for k, *_ in spec)
    for key, shape, _, _ in spec:
        assert tuple(sd[key].shape) == tuple(shape), key
    assert len(sd) == nkeys
    if nparams:
        assert sum(v.numel() for v in sd.values()) == nparams  # 45,575,391 params + buffers
    # the key names the reference checkpoint uses
    for k in ("first.conv.weight", "first.norm.running_var", "down_blocks.1.norm.num_batches_tracked",
              "bottleneck.r0.conv1.weight", "bottleneck.r1.norm2.bias", "final.bias",
              "dense_motion_network.hourglass.encoder.down_blocks.0.conv.weight",
              "dense_motion_network.hourglass.decoder.up_blocks.2.norm.weight",
              "dense_motion_network.mask.weight", "dense_motion_network.occlusion.bias",
              "dense_motion_network.down.weight"):
        assert k in sd, k


def test_strict_load_and_error_behaviour():
    cfg = tiny_config()
    gen = OcclusionAwareGenerator(**cfg)
    sd = synthetic_state_dict(cfg)
    assert not gen.load_state_dict(sd, strict=True).missing_keys
    bad = dict(sd)
    bad.pop("final.bias")
    with pytest.raises(RuntimeError):
        gen.load_state_dict(bad, strict=True)  # same exception type as the reference's strict load
    gen.eval()
    with pytest.raises(RuntimeError, match="GPU"):  # no CPU fallback: fail loudly
        gen(torch.zeros(1, 3, 64, 64), {"value": torch.zeros(1, 10, 2)}, {"value": torch.zeros(1, 10, 2)})
    # constructor accepts (and ignores) estimate_jacobian like the reference (generator.py:15)
    OcclusionAwareGenerator(**{**cfg, "estimate_jacobian": False})
    # ... and dense_motion_params=None: a generator without a motion network (generator.py:18-23), fewer checkpoint keys
    plain = OcclusionAwareGenerator(**{**cfg, "dense_motion_params": None, "estimate_occlusion_map": False})
    assert plain.dense_motion_network is None
    assert sorted(plain.state_dict()) == sorted(k for k in gen.state_dict() if not k.startswith("dense_motion_network."))


def test_product_package_does_not_import_the_oracle():
    import subprocess, sys
    code = "import sys; import eamm_amd, eamm_amd.clip, eamm_amd.engine; print(any(m.startswith('oracle') for m in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "False"
    for fn in os.listdir(os.path.join(ROOT, "eamm_amd")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "eamm_amd", fn)).read().replace("the oracle", "")


def test_seeded_generators_are_frozen():
    cfg = tiny_config()
    a, b = synthetic_state_dict(cfg, seed=1234), synthetic_state_dict(cfg, seed=1234)
    assert all(torch.equal(a[k], b[k]) for k in a)
    # first draws of the RandomState stream: a changed generator would silently un-pin every fixture
    w = a["dense_motion_network.hourglass.encoder.down_blocks.0.conv.weight"].flatten()[:3].numpy()
    np.testing.assert_allclose(w, np.random.RandomState(1234).standard_normal(3) * np.sqrt(2.0 / (44 * 9)), rtol=1e-6)
    assert float(synthetic_source(8, seed=1).sum()) == float(synthetic_source(8, seed=1).sum())
    kp = synthetic_keypoints(3, 10, seed=2)
    assert kp["value"].shape == (3, 10, 2) and kp["jacobian"].shape == (3, 10, 2, 2)
    assert torch.equal(kp["value"][1], synthetic_keypoints(1, 10, seed=3)["value"][0])  # frame t <-> seed 2+t
    k = antialias_kernel(3)
    assert k.shape == (3, 1, 13, 13) and abs(float(k[0].sum()) - 1) < 1e-6


def test_shard_bounds_partition():
    for total in (0, 1, 7, 16, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(2048, 8, 3) == (768, 1024)  # BASELINE config 4: 256 frames per GPU
