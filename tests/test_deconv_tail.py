"""Audio-to-feature deconvolution tail: oracle vs the fixture captured from the reference's
`AT_net2().decon` (CPU), and the HIP module vs the same fixture and the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from eamm_amd import DeconvTail, KPDetector_a, driving_keypoints, kp_detector_a_config
from eamm_amd.weights import DECONV_CHANNELS, deconv_state_dict_spec, kp_state_dict_spec, synthetic_state_dict
from oracle import eamm_oracle as orc

TOL = 2e-5   # stated fp32 tolerance (max abs, outputs are O(1)); the reference's own fp32-vs-fp64 floor is 1.6e-6


def fixture():
    z = np.load(os.path.join(GOLDEN, "deconv_tail.npz"))
    sd = synthetic_state_dict(None, seed=int(z["weight_seed"]), spec=deconv_state_dict_spec())
    return z, sd, torch.from_numpy(z["x"])


def test_oracle_matches_reference_fixture():
    z, sd, x = fixture()
    out = orc.deconv_tail(sd, x)
    assert tuple(out.shape) == (x.shape[0], 35, 64, 64)
    assert float((out - torch.from_numpy(z["out"])).abs().max()) <= max(2e-6, 2 * float(z["floor"]))
    # frames are independent: the reference's per-frame calls (util.py:603-607) equal one batched call
    one = torch.cat([orc.deconv_tail(sd, x[i:i + 1]) for i in range(x.shape[0])], 0)
    assert float((one - out).abs().max()) <= 5e-6


def test_module_layout_and_errors():
    m = DeconvTail()
    spec = deconv_state_dict_spec()
    sd = m.state_dict()
    # nn.Sequential numbering of the reference (util.py:559-574): 0,1,(2) 3,4,(5) ... 12
    assert sorted(sd) == sorted(k for k, *_ in spec) and all(tuple(sd[k].shape) == tuple(s) for k, s, *_ in spec)
    assert "12.weight" in sd and tuple(sd["12.weight"].shape) == (128, 35, 4, 4) and "13.weight" not in sd
    # drop-in inside a parent module: the checkpoint's `decon.N.*` keys load unchanged
    parent = torch.nn.Module()
    parent.decon = DeconvTail()
    parent.load_state_dict({"decon." + k: v for k, v in synthetic_state_dict(None, seed=1, spec=spec).items()}, strict=True)
    m.eval()
    with pytest.raises(RuntimeError, match="GPU"):        # no CPU fallback
        m(torch.zeros(1, 256, 1, 1))
    with pytest.raises(ValueError):
        DeconvTail(channels=(256, 35))


@pytest.mark.gpu
def test_hip_module_matches_reference_fixture():
    z, sd, x = fixture()
    m = DeconvTail().eval()
    m.load_state_dict(sd, strict=True)
    m.cuda()
    out = m(x.cuda()[:, :, None, None])                    # the reference's call shape [B,256,1,1]
    assert tuple(out.shape) == (x.shape[0], 35, 64, 64)
    err = float((out.cpu() - torch.from_numpy(z["out"])).abs().max())
    assert err <= TOL, err


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 7, 40])
def test_hip_module_matches_oracle_any_batch(batch):
    _, sd, _ = fixture()
    m = DeconvTail(max_batch=8).eval()                     # 40 > max_batch: the handle is rebuilt larger
    m.load_state_dict(sd, strict=True)
    m.cuda()
    g = torch.Generator().manual_seed(batch)
    x = torch.randn(batch, 256, generator=g)
    out = m(x.cuda()).cpu()
    ref = orc.deconv_tail(sd, x)
    assert float((out - ref).abs().max()) <= TOL
    again = m(x.cuda()).cpu()
    assert torch.equal(out, again)                         # deterministic: no atomics in the split-K path
    one = m(x[batch // 2:batch // 2 + 1].cuda()).cpu()     # batch independence
    assert float((one - out[batch // 2:batch // 2 + 1]).abs().max()) <= 5e-6


@pytest.mark.gpu
def test_hip_module_small_config_and_weight_update():
    ch = (64, 32, 32, 19)                                  # 3 layers: 1x1 -> 4x4 -> 8x8 -> 16x16, odd output width
    spec = deconv_state_dict_spec(ch)
    sd = synthetic_state_dict(None, seed=3, spec=spec)
    m = DeconvTail(channels=ch).eval()
    m.load_state_dict(sd, strict=True)
    m.cuda()
    x = torch.randn(5, 64, generator=torch.Generator().manual_seed(0))
    out = m(x.cuda()).cpu()
    assert tuple(out.shape) == (5, 19, 16, 16)
    assert float((out - orc.deconv_tail(sd, x)).abs().max()) <= TOL
    sd2 = synthetic_state_dict(None, seed=4, spec=spec)    # in-place weight update is picked up (version counters)
    m.load_state_dict(sd2, strict=True)
    assert float((m(x.cuda()).cpu() - orc.deconv_tail(sd2, x)).abs().max()) <= TOL


@pytest.mark.gpu
def test_driving_keypoints_chain_matches_oracle():
    """LSTM features -> decon -> KPDetector_a, the per-frame front end of demo.py:219, batched on the device, against
    the oracle's composition of the two reference restatements."""
    _, sd_d, _ = fixture()
    cfg = kp_detector_a_config()
    sd_k = synthetic_state_dict(cfg, seed=77, spec=kp_state_dict_spec(cfg))
    tail = DeconvTail().eval()
    tail.load_state_dict(sd_d, strict=True)
    kpa = KPDetector_a(**cfg).eval()
    kpa.load_state_dict(sd_k, strict=True)
    tail.cuda(), kpa.cuda()
    x = 0.3 * torch.randn(1, 11, 256, generator=torch.Generator().manual_seed(5))   # [1,T,256] as lstm_out
    kp = driving_keypoints(tail, kpa, x.cuda(), batch=4)                             # 4 + 4 + 3: ragged last batch
    with torch.no_grad():
        ref = orc.kp_detector_a_forward(sd_k, cfg, orc.deconv_tail(sd_d, x[0]))
    assert tuple(kp["value"].shape) == (11, 10, 2) and tuple(kp["jacobian"].shape) == (11, 10, 2, 2)
    assert float((kp["value"].cpu() - ref["value"]).abs().max()) <= 5e-5
    assert float((kp["jacobian"].cpu() - ref["jacobian"]).abs().max()) <= 2e-4


@pytest.mark.gpu
def test_split_hand_over_to_kp_detector_a_is_the_nchw_path_bit_for_bit():
    """Round 6: DeconvTail.forward_split writes the 35 channels in the layout KPDetector_a's heads read -- NHWC wide
    [B,64,64,32] + one float4 per pixel -- instead of the reference's NCHW [B,35,64,64] (util.py:604-607 -> keypoint_detector.py:180-205).
    The values are forward()'s, the key points are those of the NCHW path, bit for bit, at one frame and at a clip-harness batch."""
    from eamm_amd import KPDetector_a, SplitFeatureMap, driving_keypoints, kp_detector_a_config
    from eamm_amd.weights import synthetic_lstm_features, trained_like_kp_state_dict
    tail = DeconvTail()
    tail.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()), strict=True)
    ca = kp_detector_a_config()
    kpa = KPDetector_a(**ca)
    kpa.load_state_dict(trained_like_kp_state_dict(ca, 77), strict=True)
    tail, kpa = tail.cuda().eval(), kpa.cuda().eval()
    assert tail.split_channels() == 32 and kpa.accepts_split(64, 64) == 32
    feats = synthetic_lstm_features(70, seed=5).cuda()
    with torch.no_grad():
        for b in (1, 3, 70):
            x = feats[:b].contiguous()
            nchw = tail(x)
            sp = tail.forward_split(x)
            assert isinstance(sp, SplitFeatureMap) and sp.shape == (b, 35, 64, 64)
            assert torch.equal(sp.to_nchw(), nchw)                      # same values, other layout
            assert float(sp.thin[..., 3].abs().max()) == 0.0             # the float4's fourth slot
            a, s = kpa(nchw), kpa(sp)
            for k in ("value", "jacobian", "heatmap"):
                assert torch.equal(a[k], s[k]), (b, k)

        class Plain:                                                     # hides forward_split: the reference's hand-over
            def __call__(self, t):
                return tail(t)
        ref, got = driving_keypoints(Plain(), kpa, feats, batch=16), driving_keypoints(tail, kpa, feats, batch=16)
        assert all(torch.equal(ref[k], got[k]) for k in ref)
    # a tail whose output is not 32 m + 3 channels wide has no split form
    other = DeconvTail(channels=(64, 64, 32)).cuda().eval()
    assert other.split_channels() == 0
    with pytest.raises(RuntimeError, match="32 m \\+ 3"):
        other.forward_split(torch.zeros(1, 64, device="cuda"))
