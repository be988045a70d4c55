"""Key-point normalisation of the clip loop against fixtures produced by the reference's own function
(demo.py:112-132, extracted by oracle/make_golden.py), every flag combination, batched over frames."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from eamm_amd import normalize_kp


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(GOLDEN, "normalize_kp.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("adapt", [0, 1])
@pytest.mark.parametrize("rel", [0, 1])
@pytest.mark.parametrize("relj", [0, 1])
def test_normalize_kp_matches_reference(fx, adapt, rel, relj):
    kp_s = {"value": fx["kp_source_value"], "jacobian": fx["kp_source_jacobian"]}
    kp_i = {"value": fx["kp_initial_value"], "jacobian": fx["kp_initial_jacobian"]}
    kp_d = {"value": fx["kp_driving_value"].clone(), "jacobian": fx["kp_driving_jacobian"].clone()}
    out = normalize_kp(kp_s, kp_d, kp_i, adapt_movement_scale=bool(adapt), use_relative_movement=bool(rel),
                       use_relative_jacobian=bool(relj))            # all 5 frames in one call
    assert torch.allclose(out["value"], fx[f"value_a{adapt}r{rel}j{relj}"], atol=1e-6, rtol=0)
    assert torch.allclose(out["jacobian"], fx[f"jacobian_a{adapt}r{rel}j{relj}"], atol=2e-6, rtol=0)
    # inputs are not modified (the reference mutates a temporary only)
    assert torch.equal(kp_d["value"], fx["kp_driving_value"])
    if not rel:   # `relative=False` is the identity (demo.py:558 default) whatever adapt_movement_scale says
        assert torch.equal(out["value"], kp_d["value"]) and torch.equal(out["jacobian"], kp_d["jacobian"])


@pytest.mark.parametrize("name,kw", [("kp", {}), ("emo", dict(mincutoff=1.0, beta=0.2, dcutoff=1.0, freq=100.0, scale=100.0))])
def test_smooth_keypoints_matches_reference_filter(name, kw):
    """Temporal smoothing between the two loops of make_animation_smooth (demo.py:231-250) against the reference's own
    filter1.OneEuroFilter driven per frame (fixture from oracle/make_golden.py): defaults = the key-point filter,
    the second parameter set is the emotion-displacement filter."""
    from eamm_amd import one_euro_smooth, smooth_keypoints
    z = np.load(os.path.join(GOLDEN, "one_euro.npz"))
    seq = {"value": torch.from_numpy(z["value"]), "jacobian": torch.from_numpy(z["jacobian"])}
    out = smooth_keypoints(seq, **kw)
    for k in ("value", "jacobian"):
        ref = torch.from_numpy(z[f"{name}_{k}"])
        assert out[k].shape == ref.shape and torch.equal(out[k], ref), (name, k)     # bit-equal to the reference filter (round 6)
    assert torch.allclose(out["value"][0], seq["value"][0], atol=1e-7, rtol=0)   # the first frame passes through (x*s/s)
    assert float((out["value"] - seq["value"]).abs().max()) > 1e-3       # ... and later ones are really filtered
    flat = one_euro_smooth(seq["value"].reshape(24, -1), **(kw or dict(mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)))
    assert torch.equal(flat.reshape(out["value"].shape), out["value"])  # element-wise: the shape does not matter


def test_emotion_offsets_match_reference_loop():
    """The `--add_emo` step of the clip loop (demo.py:263-271) followed by normalize_kp (demo.py:276), whole clip at once,
    against the reference's own statements run frame by frame (fixture: oracle/make_golden.py emotion_case)."""
    from eamm_amd import apply_emotion_offsets
    z = np.load(os.path.join(GOLDEN, "emotion_offsets.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    kp_d = {"value": g["kp_driving_value"].clone(), "jacobian": g["kp_driving_jacobian"].clone()}
    emo = {"value": g["emo_value"], "jacobian": g["emo_jacobian"]}
    out = apply_emotion_offsets(kp_d, emo)
    assert torch.equal(out["value"], g["offset_value"]) and torch.equal(out["jacobian"], g["offset_jacobian"])  # same fp32 ops
    assert torch.equal(kp_d["value"], g["kp_driving_value"])          # returns new tensors
    touched = (out["value"] != kp_d["value"]).any(dim=-1).any(dim=0)
    assert touched.nonzero().flatten().tolist() == [1, 4, 6]           # key points 1, 4, 6 only (demo.py:266-271)
    kp_s = {"value": g["kp_source_value"], "jacobian": g["kp_source_jacobian"]}
    kp_i = {"value": g["kp_initial_value"], "jacobian": g["kp_initial_jacobian"]}
    nrm = normalize_kp(kp_s, out, kp_i, adapt_movement_scale=True, use_relative_movement=True, use_relative_jacobian=True)
    assert torch.allclose(nrm["value"], g["normalized_value"], atol=1e-6, rtol=0)
    assert torch.allclose(nrm["jacobian"], g["normalized_jacobian"], atol=2e-6, rtol=0)
    with pytest.raises(ValueError):
        apply_emotion_offsets(kp_d, emo, kind="linear_4")
    with pytest.raises(RuntimeError):
        apply_emotion_offsets(kp_d, {"value": emo["value"][:, :2], "jacobian": emo["jacobian"][:, :2]})


def test_oracle_clip_harness_restatements_match_the_reference_fixtures():
    """oracle/eamm_oracle.py restates filter1.OneEuroFilter and demo.py:normalize_kp statement for statement (the checker of the
    end-to-end GPU test, tests/test_gpu_pipeline.py): both against the fixtures the reference's own code produced."""
    from oracle import eamm_oracle as orc
    z = np.load(os.path.join(GOLDEN, "one_euro.npz"))
    for name, p in (("kp", (0.05, 8, 1.0, 100, 10)), ("emo", (1, 0.2, 1.0, 100, 100))):
        for k in ("value", "jacobian"):
            out = orc.smooth_sequence(torch.from_numpy(z[k]), *p)
            assert torch.equal(out, torch.from_numpy(z[f"{name}_{k}"])), (name, k)      # same statements, same floats
    z = np.load(os.path.join(GOLDEN, "normalize_kp.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    kp_s = {"value": g["kp_source_value"], "jacobian": g["kp_source_jacobian"]}
    kp_i = {"value": g["kp_initial_value"], "jacobian": g["kp_initial_jacobian"]}
    for a in (0, 1):
        for r in (0, 1):
            for j in (0, 1):
                for t in range(g["kp_driving_value"].shape[0]):
                    kd = {"value": g["kp_driving_value"][t:t + 1], "jacobian": g["kp_driving_jacobian"][t:t + 1]}
                    o = orc.normalize_kp(kp_s, kd, kp_i, bool(a), bool(r), bool(j))
                    assert torch.equal(o["value"], g[f"value_a{a}r{r}j{j}"][t:t + 1])
                    assert torch.equal(o["jacobian"], g[f"jacobian_a{a}r{r}j{j}"][t:t + 1])


def test_trained_like_detector_weights_give_well_conditioned_jacobians():
    from eamm_amd import kp_detector_a_config
    from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, synthetic_state_dict, trained_like_kp_state_dict
    from oracle import eamm_oracle as orc
    cfg = kp_detector_a_config()
    sd = trained_like_kp_state_dict(cfg, 77)
    sd_d = synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec())
    with torch.no_grad():
        kp = orc.kp_detector_a_forward(sd, cfg, orc.deconv_tail(sd_d, synthetic_lstm_features(4)))
    assert float(torch.linalg.cond(kp["jacobian"]).max()) < 2.0 and float(torch.det(kp["jacobian"]).min()) > 0.5
    assert float((kp["value"][1:] - kp["value"][:-1]).abs().max()) > 1e-3        # consecutive frames move
