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
