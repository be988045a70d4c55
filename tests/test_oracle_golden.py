"""The oracle against the fixtures captured from the imported reference (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from eamm_amd.config import hot_path_config, tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc



def gray_config():
    """num_channels = 1 (reference modules/generator.py:14,25,46; modules/dense_motion.py:17-18,27 accept any)."""
    return {**tiny_config(), "num_channels": 1}


def two_channel_config():
    return {**tiny_config(), "num_channels": 2}


def rgba_config():
    """num_channels = 4: beyond the kernels' float4 pixel record -- two groups of three channels (csrc/motion.hip)."""
    return {**tiny_config(), "num_channels": 4}


def six_channel_config():
    return {**tiny_config(), "num_channels": 6}


def num_kp_config(k, full=False):
    """num_kp != 10 (reference dense_motion.py:15-18, generator.py:14 accept any; round 6)."""
    def make():
        return {**(hot_path_config() if full else tiny_config()), "num_kp": k}
    make.__name__ = f"{'full' if full else 'tiny'}_kp{k}_config"
    return make


NUM_KP_CASES = [("tiny64_kp1", num_kp_config(1)), ("tiny64_kp5", num_kp_config(5)), ("tiny64_kp15", num_kp_config(15)),
                ("tiny64_kp30", num_kp_config(30)), ("tiny64_kp30_nojac", num_kp_config(30)),
                ("full256_kp15", num_kp_config(15, full=True)), ("full256_kp5", num_kp_config(5, full=True))]

CASES = [("tiny64_clip3", tiny_config), ("tiny64_batch2", tiny_config), ("tiny64_nojac", tiny_config),
         ("full256_clip2", hot_path_config), ("full512_clip1", hot_path_config),
         ("tiny64_gray", gray_config), ("tiny64_two_channels", two_channel_config),
         ("tiny64_rgba", rgba_config), ("tiny64_six_channels", six_channel_config)] + NUM_KP_CASES


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def inputs_from_fixture(fx, cfg):
    """Rebuild (state_dict, source batch, kp_driving, kp_source batch) exactly as make_golden.py did."""
    n = fx["kp_driving_value"].shape[0]
    size = int(fx["size"])
    per_frame = bool(fx["per_frame_source"])
    sd = synthetic_state_dict(cfg, seed=int(fx["weight_seed"]))
    nsrc = n if per_frame else 1
    src = synthetic_source(size, seed=int(fx["source_seed"]), batch=nsrc, channels=cfg["num_channels"])
    kp_s = {"value": torch.from_numpy(fx["kp_source_value"]), "jacobian": torch.from_numpy(fx["kp_source_jacobian"])}
    kp_d = {"value": torch.from_numpy(fx["kp_driving_value"])}
    if "kp_driving_jacobian" in fx:
        kp_d["jacobian"] = torch.from_numpy(fx["kp_driving_jacobian"])
    return sd, src, kp_d, kp_s, n, per_frame


def sample(t, key, fx):
    s = int(fx[key + "_stride"])
    return t[:, ::s, ::s] if key == "deformation" else t[..., ::s, ::s]


@pytest.mark.parametrize("name,cfg_fn", CASES)
def test_oracle_matches_reference_fixture(name, cfg_fn):
    cfg = cfg_fn()
    fx = load_case(name)
    sd, src, kp_d, kp_s, n, per_frame = inputs_from_fixture(fx, cfg)
    # the fixture inputs must be what the seeded generators produce (both sides regenerate them)
    ref_kp_s = synthetic_keypoints(src.shape[0], cfg["num_kp"], seed=0)
    ref_kp_d = synthetic_keypoints(n, cfg["num_kp"], seed=2)
    assert np.array_equal(ref_kp_s["value"].numpy(), fx["kp_source_value"])
    assert np.array_equal(ref_kp_d["value"].numpy(), fx["kp_driving_value"])
    if not per_frame:
        src = src.expand(n, -1, -1, -1).contiguous()
        kp_s = {k: v.expand(n, *v.shape[1:]).contiguous() for k, v in kp_s.items()}
    with torch.no_grad():
        out = orc.generator_forward(sd, cfg, src, kp_d, kp_s)
    summary = json.load(open(os.path.join(GOLDEN, "summary.json")))["cases"][name]
    for key in ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed", "deformation"):
        want = torch.from_numpy(fx[key])
        got = sample(out[key], key, fx)
        assert got.shape == want.shape, (key, got.shape, want.shape)
        err = float((got - want).abs().max())
        # same ATen kernels, same order: agreement to the fp32 noise floor of the reference itself
        floor = summary[key]["fp32_vs_fp64_floor"]
        assert err <= max(2e-6, 2 * floor), (name, key, err, floor)


def test_clip_harness_matches_fixture():
    """animate_clip (counterpart of demo.py:251-281) returns HWC float32 frames == prediction."""
    cfg = tiny_config()
    fx = load_case("tiny64_clip3")
    sd, src, kp_d, kp_s, n, _ = inputs_from_fixture(fx, cfg)
    frames = orc.animate_clip(sd, cfg, src, kp_s, kp_d)
    assert len(frames) == n and frames[0].shape == (64, 64, 3) and frames[0].dtype == np.float32
    want = np.transpose(fx["prediction"], [0, 2, 3, 1])
    for t in range(n):
        assert np.abs(frames[t] - want[t]).max() <= 2e-6


def test_oracle_edge_semantics():
    """Appendix-A facts the kernels rely on, checked on the oracle's building blocks."""
    g = orc.coordinate_grid(4, 6, torch.float32)
    assert g.shape == (4, 6, 2)
    assert torch.allclose(g[0, :, 0], torch.linspace(-1, 1, 6)) and torch.allclose(g[:, 0, 1], torch.linspace(-1, 1, 4))
    # identity grid + align_corners=False does NOT reproduce the input: border attenuation (FOMM quirk)
    x = torch.ones(1, 1, 4, 6)
    y = orc.warp_by_flow(x, g[None])
    assert float(y[0, 0, 0, 0]) < 1.0 and abs(float(y[0, 0, 2, 3]) - 1.0) < 1e-6
    # key points far outside [-1,1]: every warp samples zero padding, nothing is NaN
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1)
    kp_s = synthetic_keypoints(1, 10, seed=0)
    kp_d = {"value": torch.full((1, 10, 2), 7.0), "jacobian": torch.eye(2).expand(1, 10, 2, 2).contiguous()}
    with torch.no_grad():
        out = orc.generator_forward(sd, cfg, synthetic_source(64), kp_d, kp_s)
    assert all(torch.isfinite(v).all() for v in out.values())
    # singular driving jacobian -> torch.inverse raises (the error contract of dense_motion.py:56)
    kp_bad = {"value": torch.zeros(1, 10, 2), "jacobian": torch.zeros(1, 10, 2, 2)}
    with pytest.raises(RuntimeError):
        orc.generator_forward(sd, cfg, synthetic_source(64), kp_bad, kp_s)


def test_oracle_generator_without_motion_network():
    """dense_motion_params=None (generator.py:18-23, 64): 'prediction' only, against the reference's output."""
    cfg = tiny_config()
    cfg["dense_motion_params"], cfg["estimate_occlusion_map"] = None, False
    fx = load_case("tiny64_nomotion")
    sd = synthetic_state_dict(cfg, seed=int(fx["weight_seed"]))
    assert not any(k.startswith("dense_motion_network") for k in sd)
    with torch.no_grad():
        out = orc.generator_forward(sd, cfg, synthetic_source(64, seed=int(fx["source_seed"]), batch=2), None, None)
    assert sorted(out) == ["prediction"]
    assert float((out["prediction"] - torch.from_numpy(fx["prediction"])).abs().max()) <= 2e-6
