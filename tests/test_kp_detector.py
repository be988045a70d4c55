"""Key-point detectors: oracle vs fixtures captured from the reference's KPDetector /
KPDetector_a (CPU), and the HIP modules vs the same fixtures and the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from eamm_amd import KPDetector, KPDetector_a, kp_detector_a_config, kp_detector_config, tiny_kp_config
from eamm_amd.weights import kp_state_dict_spec, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc

# stated tolerances (max abs); the reference's own fp32-vs-fp64 floor is <= 3.6e-6 / 8.4e-6 / 4.5e-6
TOL_KP = {"value": 5e-5, "jacobian": 1e-4, "heatmap": 5e-5}
CASES = [("kp_tiny64", lambda: tiny_kp_config(), False), ("kp_full256", kp_detector_config, False),
         ("kpa_tiny", lambda: tiny_kp_config(audio=True), True), ("kpa_full", kp_detector_a_config, True),
         ("kp_tiny64_gray", lambda: {**tiny_kp_config(), "num_channels": 1}, False),   # one image channel (keypoint_detector.py:17-21)
         ("kp_tiny64_rgba", lambda: {**tiny_kp_config(), "num_channels": 4}, False),   # four / six: four channels per float4 slot
         ("kp_tiny64_six_channels", lambda: {**tiny_kp_config(), "num_channels": 6}, False)]
# round 6: num_kp != 10 (reference keypoint_detector.py:24-41 sizes the heat-map and jacobian heads by it)
for _k in (1, 5, 15, 30):
    CASES.append((f"kp_tiny64_k{_k}", (lambda k=_k: {**tiny_kp_config(), "num_kp": k}), False))
    CASES.append((f"kpa_tiny_k{_k}", (lambda k=_k: {**tiny_kp_config(audio=True), "num_kp": k}), True))


def load(name, cfg, audio):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fx = {k: z[k] for k in z.files}
    sd = synthetic_state_dict(cfg, seed=int(fx["weight_seed"]), spec=kp_state_dict_spec(cfg))
    x = torch.from_numpy(fx["feature_map"]) if audio else synthetic_source(int(fx["size"]), seed=3, batch=int(fx["batch"]),
                                                                            channels=cfg["num_channels"])
    return fx, sd, x


@pytest.mark.parametrize("name,cfg_fn,audio", CASES)
def test_oracle_matches_reference_fixture(name, cfg_fn, audio):
    cfg = cfg_fn()
    fx, sd, x = load(name, cfg, audio)
    with torch.no_grad():
        out = (orc.kp_detector_a_forward if audio else orc.kp_detector_forward)(sd, cfg, x)
    for k in ("value", "jacobian", "heatmap"):
        err = float((out[k] - torch.from_numpy(fx[k])).abs().max())
        assert err <= max(2e-6, 2 * float(fx[k + "_floor"])), (name, k, err)
    # soft-argmax of a spatial softmax: heat-maps sum to one, values stay inside the [-1,1] grid
    assert torch.allclose(out["heatmap"].sum(dim=(2, 3)), torch.ones(x.shape[0], cfg["num_kp"]), atol=1e-5)
    assert float(out["value"].abs().max()) <= 1.0


def test_module_layout_and_errors():
    cfg = kp_detector_config()
    m = KPDetector(**cfg)
    spec = kp_state_dict_spec(cfg)
    sd = m.state_dict()
    assert sorted(sd) == sorted(k for k, *_ in spec) and all(tuple(sd[k].shape) == tuple(s) for k, s, *_ in spec)
    # reference init of the jacobian head: zero weights, identity bias (keypoint_detector.py:27-28)
    assert float(m.jacobian.weight.abs().max()) == 0 and m.jacobian.bias[:4].tolist() == [1, 0, 0, 1]
    a = KPDetector_a(**kp_detector_a_config())
    assert sorted(a.state_dict()) == sorted(sd)           # same checkpoint layout, predictor included
    m.eval()
    with pytest.raises(RuntimeError, match="GPU"):        # no CPU fallback
        m(torch.zeros(1, 3, 256, 256))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg_fn,audio", CASES)
def test_hip_modules_match_reference_fixture(name, cfg_fn, audio):
    cfg = cfg_fn()
    fx, sd, x = load(name, cfg, audio)
    mod = (KPDetector_a if audio else KPDetector)(**cfg)
    mod.load_state_dict(sd, strict=True)
    mod = mod.to("cuda:0").eval()
    out = mod(x.to("cuda:0"))
    assert set(out) == {"value", "jacobian", "heatmap"}
    errs = {k: float((out[k].cpu() - torch.from_numpy(fx[k])).abs().max()) for k in out}
    print("\n" + name + "  " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    for k, e in errs.items():
        assert out[k].shape == fx[k].shape and e <= TOL_KP[k], (name, k, e)
    # batch of one == row of the batch (split-K plans may differ with M: tolerance, not bit-exactness)
    one = mod(x[1:2].to("cuda:0"))
    assert float((one["value"][0] - out["value"][1]).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg_fn", [("kpa_tiny", lambda: tiny_kp_config(audio=True)), ("kpa_full", kp_detector_a_config)])
def test_kp_detector_a_padded_head_form(name, cfg_fn, monkeypatch):
    """EAMM_KPA_THIN=0: the heads of KPDetector_a as ONE convolution over the feature map zero-padded from 35 to 64 channels (rounds
    1-4) instead of the wide (32) + thin (3) split of round 5 -- the same fixture, and the two forms against each other."""
    cfg = cfg_fn()
    fx, sd, x = load(name, cfg, True)
    split = KPDetector_a(**cfg)
    split.load_state_dict(sd, strict=True)
    a = split.to("cuda:0").eval()(x.to("cuda:0"))
    monkeypatch.setenv("EAMM_KPA_THIN", "0")
    padded = KPDetector_a(**cfg)
    padded.load_state_dict(sd, strict=True)
    b = padded.to("cuda:0").eval()(x.to("cuda:0"))
    for k in ("value", "jacobian", "heatmap"):
        assert float((b[k].cpu() - torch.from_numpy(fx[k])).abs().max()) <= TOL_KP[k], (name, k)
        assert float((a[k] - b[k]).abs().max()) <= TOL_KP[k] / 2, (name, k)


@pytest.mark.gpu
def test_kp_detector_feeds_generator_contract():
    """demo.py:206,219,279: kp_source = kp_detector(source); kp_driving = kp_detector_a(feature map); both dicts go
    straight into the generator (extra 'heatmap' key ignored)."""
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gcfg = tiny_config()
    gen = OcclusionAwareGenerator(**gcfg)
    gen.load_state_dict(synthetic_state_dict(gcfg, seed=1234))
    gen = gen.cuda().eval()
    kcfg, acfg = tiny_kp_config(), tiny_kp_config(audio=True)
    kp = KPDetector(**kcfg)
    kp.load_state_dict(synthetic_state_dict(kcfg, seed=77, spec=kp_state_dict_spec(kcfg)))
    kpa = KPDetector_a(**acfg)
    kpa.load_state_dict(synthetic_state_dict(acfg, seed=78, spec=kp_state_dict_spec(acfg)))
    kp, kpa = kp.cuda().eval(), kpa.cuda().eval()
    src = synthetic_source(64, seed=1).cuda()
    fmap = torch.randn(1, 35, 16, 16, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():                 # demo.py:195: the whole animation loop runs without autograd
        kp_source, kp_driving = kp(src), kpa(fmap)
        out = gen(src, kp_source=kp_source, kp_driving=kp_driving)
    assert out["prediction"].shape == (1, 3, 64, 64) and torch.isfinite(out["prediction"]).all()
    sd_g = synthetic_state_dict(gcfg, seed=1234)
    with torch.no_grad():
        ref = orc.generator_forward(sd_g, gcfg, src.cpu(), {k: v.cpu() for k, v in kp_driving.items() if k != "heatmap"},
                                    {k: v.cpu() for k, v in kp_source.items() if k != "heatmap"})
    assert float((out["prediction"].cpu() - ref["prediction"]).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("audio,k", [(True, 10), (False, 10), (True, 15)])
def test_batched_heads_on_the_lds_dma_tile_match_the_oracle_and_the_small_call(audio, k):
    """Round 6: from B*h*w >= 16384 pixels (4 frames at 256x256) the heads' 7x7 convolution runs on the LDS-DMA tile (512 x 64 columns;
    256 x 128 when K + 4 K > 64) instead of the register-staged 128 x 64 one -- the clip harness's 64-frame front-end batches.  Eight
    frames (DMA tile) against the oracle, and their first two against a two-frame call (register-staged tile) of the same module."""
    cfg = {**(kp_detector_a_config() if audio else kp_detector_config()), "num_kp": k}
    sd = synthetic_state_dict(cfg, seed=77, spec=kp_state_dict_spec(cfg))
    mod = (KPDetector_a if audio else KPDetector)(**cfg)
    mod.load_state_dict(sd, strict=True)
    mod = mod.to("cuda:0").eval()
    if audio:
        rs = np.random.RandomState(11)
        x = torch.from_numpy(rs.standard_normal((8, cfg["block_expansion"] + cfg["num_channels_a"], 64, 64)).astype(np.float32))
    else:
        x = synthetic_source(256, seed=4, batch=8)
    with torch.no_grad():
        big = {n: v.cpu() for n, v in mod(x.to("cuda:0")).items()}
        small = {n: v.cpu() for n, v in mod(x[:2].to("cuda:0")).items()}
        ref = (orc.kp_detector_a_forward if audio else orc.kp_detector_forward)(sd, cfg, x)
        lean = mod.detect(x.to("cuda:0"), heatmap=False)       # what the clip harness asks for: no heat-map pass
    assert set(lean) == {"value", "jacobian"} and all(torch.equal(lean[n].cpu(), big[n]) for n in lean)
    assert float((big["heatmap"].sum(dim=(2, 3)) - 1).abs().max()) <= 1e-5     # the sliced form's heat-maps are normalised by the combined sum
    errs = {n: float((big[n] - ref[n]).abs().max()) for n in big}
    print(f"\nbatched heads audio={audio} K={k}: " + "  ".join(f"{n}={e:.2e}" for n, e in errs.items()))
    for n, e in errs.items():
        assert e <= TOL_KP[n], (n, e)
        assert float((big[n][:2] - small[n]).abs().max()) <= TOL_KP[n] / 2, n


@pytest.mark.gpu
@pytest.mark.parametrize("audio", [False, True])
def test_sliced_head_reductions_on_small_maps(audio):
    """The pixel-sliced reductions (B >= 8) where a slice holds a dozen pixels: the tiny detectors' 10 x 10 heat-map windows, nine images,
    against the oracle and against the per-(image, key point) kernel's results for the same images (calls of two)."""
    cfg = tiny_kp_config(audio=audio)
    sd = synthetic_state_dict(cfg, seed=77, spec=kp_state_dict_spec(cfg))
    mod = (KPDetector_a if audio else KPDetector)(**cfg)
    mod.load_state_dict(sd, strict=True)
    mod = mod.to("cuda:0").eval()
    if audio:
        x = torch.from_numpy(np.random.RandomState(12).standard_normal((9, cfg["block_expansion"] + cfg["num_channels_a"], 16, 16)).astype(np.float32))
    else:
        x = synthetic_source(64, seed=6, batch=9)
    with torch.no_grad():
        big = {n: v.cpu() for n, v in mod(x.to("cuda:0")).items()}
        pairs = [mod(x[i:i + 2].to("cuda:0")) for i in (0, 2, 4, 6)]
        ref = (orc.kp_detector_a_forward if audio else orc.kp_detector_forward)(sd, cfg, x)
    for n in ("value", "jacobian", "heatmap"):
        assert float((big[n] - ref[n]).abs().max()) <= TOL_KP[n], n
        small = torch.cat([p[n].cpu() for p in pairs])
        assert float((big[n][:8] - small).abs().max()) <= TOL_KP[n] / 2, n
