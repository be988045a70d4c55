"""The reference's make_animation_smooth (demo.py:194-282) end to end on the GPU -- eamm_amd.animate_from_features: KPDetector on
the source, DeconvTail + KPDetector_a per frame, One-Euro smoothing, emotion offsets, normalize_kp, the generator, uint8 frames
in pinned host memory -- against the oracle's chain (oracle/eamm_oracle.py::animation_keypoints + generator_forward: the
reference's statements frame by frame, its smoothing / normalisation pinned to the reference fixtures in test_normalize_kp.py).

Staged, because a key-point error is amplified on its way through the generator (measured with the oracle at 256x256: the N1
tolerances 5e-5 / 2e-4 on value / jacobian move a pixel by up to 1.7e-3): (i) every key-point stage against the oracle at its
own tolerance, (ii) the frames against the oracle generator fed the PRODUCT's normalised key points at the path's tolerance
(1e-4), (iii) the whole chain against the whole oracle chain, printed, bound 2e-3 and one uint8 level."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, TOL
from eamm_amd import (DeconvTail, KPDetector, KPDetector_a, OcclusionAwareGenerator, animate_from_features, hot_path_config,
                      kp_detector_a_config, kp_detector_config, one_euro_smooth)
from eamm_amd.weights import (deconv_state_dict_spec, synthetic_lstm_features, synthetic_source, synthetic_state_dict,
                              trained_like_kp_state_dict)
from oracle import eamm_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build():
    cfg = hot_path_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(sd, strict=True)
    cfg_k, cfg_a = kp_detector_config(), kp_detector_a_config()
    sd_k, sd_a = trained_like_kp_state_dict(cfg_k, 78), trained_like_kp_state_dict(cfg_a, 77)
    sd_d = synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec())
    kp, kpa, tail = KPDetector(**cfg_k), KPDetector_a(**cfg_a), DeconvTail()
    kp.load_state_dict(sd_k, strict=True)
    kpa.load_state_dict(sd_a, strict=True)
    tail.load_state_dict(sd_d, strict=True)
    mods = [m.to(DEV).eval() for m in (gen, kp, tail, kpa)]
    return cfg, sd, cfg_k, sd_k, cfg_a, sd_a, sd_d, mods


@pytest.mark.parametrize("with_emo", [False, True])
def test_whole_chain_against_the_oracle_chain(with_emo):
    cfg, sd, cfg_k, sd_k, cfg_a, sd_a, sd_d, (gen, kp, tail, kpa) = build()
    T = 20
    src = synthetic_source(256, seed=1)
    feats = synthetic_lstm_features(T, seed=5)
    emo = None
    if with_emo:
        g = torch.Generator().manual_seed(9)
        emo = {"value": 0.02 * torch.randn(T, 4, 2, generator=g), "jacobian": 0.02 * torch.randn(T, 4, 2, 2, generator=g)}
    timings = {}
    frames, span, kps = animate_from_features(gen, kp, tail, kpa, src, feats, emo_driving=emo, batch=8, front_batch=7,
                                              uint8=True, to_host=True, timings=timings, return_keypoints=True)
    assert span == (0, T) and frames.shape == (T, 256, 256, 3) and frames.dtype == torch.uint8
    assert frames.device.type == "cpu" and frames.is_pinned()
    assert {"front_ms", "smooth_ms", "normalize_ms", "encode_ms", "compute_ms", "d2h_tail_ms"} <= set(timings)
    # ---- the oracle's chain, frame by frame as the reference runs it
    kp_s, norm, raw, smooth = orc.animation_keypoints(sd_k, cfg_k, sd_d, sd_a, cfg_a, src, feats, emo_driving=emo)
    nv, nj = torch.cat([n["value"] for n in norm]), torch.cat([n["jacobian"] for n in norm])
    stage = {"kp_source": (kps["kp_source"], kp_s), "raw": (kps["kp_driving_raw"], raw),
             "normalised": (kps["kp_norm"], {"value": nv, "jacobian": nj})}
    if not with_emo:
        stage["smoothed"] = (kps["kp_driving_smoothed"], smooth)
    print()
    for name, (got, want) in stage.items():
        ev = float((got["value"].cpu() - want["value"]).abs().max())
        ej = float((got["jacobian"].cpu() - want["jacobian"]).abs().max())
        print(f"{name:11s} value {ev:.2e}  jacobian {ej:.2e}")
        assert ev <= 5e-5 and ej <= 2e-4, (name, ev, ej)        # the N1 tolerances (tests/test_kp_detector.py)
    # the filter really did something, and frame 0 is the un-smoothed initial pose
    assert float((kps["kp_driving_smoothed"]["value"] - kps["kp_driving_raw"]["value"]).abs().max()) > 1e-3 or with_emo
    # ---- (ii) frames vs the oracle generator on the PRODUCT's normalised key points
    pick = [0, 7, 8, 19]                                         # batch boundaries of the 8-frame calls, first and last
    srcb = src.expand(len(pick), -1, -1, -1).contiguous()
    ksb = {k: v.cpu().expand(len(pick), *v.shape[1:]).contiguous() for k, v in kps["kp_source"].items()}
    with torch.no_grad():
        ref_own = orc.generator_forward(sd, cfg, srcb, {k: v.cpu()[pick] for k, v in kps["kp_norm"].items()}, ksb)["prediction"]
        ref_all = orc.generator_forward(sd, cfg, srcb, {"value": nv[pick], "jacobian": nj[pick]},
                                        {k: v.expand(len(pick), *v.shape[1:]).contiguous() for k, v in kp_s.items()})["prediction"]
    want_own = torch.clamp(torch.round(ref_own * 255), 0, 255).permute(0, 2, 3, 1)
    d_own = float((frames[pick].float() - want_own).abs().max())
    # float frames of the same clip for the float comparison
    f32, _ = animate_from_features(gen, kp, tail, kpa, src, feats, emo_driving=emo, batch=8, front_batch=7, uint8=False, to_host=True)
    assert f32.shape == (T, 3, 256, 256) and f32.is_pinned()
    e_own = float((f32[pick] - ref_own).abs().max())
    e_all = float((f32[pick] - ref_all).abs().max())
    d_all = float((frames[pick].float() - torch.clamp(torch.round(ref_all * 255), 0, 255).permute(0, 2, 3, 1)).abs().max())
    print(f"frames vs oracle generator on the product's key points: {e_own:.2e} (uint8 levels {d_own:.0f});  "
          f"whole chain vs whole oracle chain: {e_all:.2e} (uint8 levels {d_all:.0f})")
    assert e_own <= TOL["prediction"] and d_own <= 1
    assert e_all <= 2e-3 and d_all <= 1
    # uint8 frames are the float frames rounded (same kernels, packing on the device)
    assert float((frames.float() - torch.clamp(torch.round(f32 * 255), 0, 255).permute(0, 2, 3, 1)).abs().max()) <= 1


@pytest.mark.parametrize("with_emo", [False, True])
def test_streamed_front_end_is_bit_identical(with_emo):
    """animate_from_features(stream=True): the front end of later frames on its own stream beside the generator of earlier ones, the
    One-Euro filter resumed from batch to batch.  Same kernels on the same batches: every key point and every frame bit-equal to
    the un-streamed path; 37 frames with 8 per front batch and 5 per generator call (ragged everywhere)."""
    cfg, sd, cfg_k, sd_k, cfg_a, sd_a, sd_d, (gen, kp, tail, kpa) = build()
    T = 37
    src, feats = synthetic_source(256, seed=1), synthetic_lstm_features(T, seed=6)
    emo = None
    if with_emo:
        g = torch.Generator().manual_seed(3)
        emo = {"value": 0.02 * torch.randn(T, 3, 2, generator=g), "jacobian": 0.02 * torch.randn(T, 3, 2, 2, generator=g)}
    kw = dict(emo_driving=emo, batch=5, front_batch=8, uint8=False, to_host=True, return_keypoints=True)
    f0, s0, k0 = animate_from_features(gen, kp, tail, kpa, src, feats, stream=False, **kw)
    f1, s1, k1 = animate_from_features(gen, kp, tail, kpa, src, feats, stream=True, **kw)
    assert s0 == s1 == (0, T)
    for name in ("kp_source", "kp_driving_raw", "kp_driving_smoothed", "kp_norm"):
        for k in ("value", "jacobian"):
            assert torch.equal(k0[name][k], k1[name][k]), (name, k)
    assert torch.equal(f0, f1)
    # the resumable filter on its own: a sequence in three chunks == the sequence whole
    seq = k0["kp_driving_raw"]["jacobian"]
    whole = one_euro_smooth(seq, mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)
    state = torch.zeros(3, 40, device=seq.device)
    parts = [one_euro_smooth(seq[a:b], mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0, state=state, resume=a > 0)
             for a, b in ((0, 1), (1, 20), (20, T))]
    assert torch.equal(torch.cat(parts), whole)


def test_one_euro_on_the_device_matches_the_reference_filter_and_is_fast():
    """eamm_op_one_euro against the reference's own filter1.OneEuroFilter outputs (fixture one_euro.npz, both parameter sets) and
    against the host filter on a 2048-frame clip; VERDICT r04: <= 2 ms per 2048 frames (the round-4 host loop took 608 ms)."""
    import os
    z = np.load(os.path.join(GOLDEN, "one_euro.npz"))
    for name, kw in (("kp", dict(mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)),
                     ("emo", dict(mincutoff=1.0, beta=0.2, dcutoff=1.0, freq=100.0, scale=100.0))):
        for k in ("value", "jacobian"):
            got = one_euro_smooth(torch.from_numpy(z[k]).to(DEV), **kw).cpu()
            err = float((got - torch.from_numpy(z[f"{name}_{k}"])).abs().max())
            print(f"\none-euro {name}/{k}: max |device - reference filter| = {err:.2e}")
            assert err == 0.0, (name, k, err)     # round 6: true divisions where ATen's CPU kernels divide -> the reference filter's bits
    g = torch.Generator().manual_seed(0)
    seq = (0.3 * torch.randn(1, 10, 2, 2, generator=g) + 0.02 * torch.cumsum(torch.randn(2048, 10, 2, 2, generator=g), 0))
    kw = dict(mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)
    host = one_euro_smooth(seq, **kw)
    dev = one_euro_smooth(seq.to(DEV), **kw)
    assert torch.equal(dev.cpu(), host)            # device filter == host filter, bit for bit, over 2048 frames
    inplace = seq.to(DEV).clone()                  # include/eamm_hip.h: out may alias x
    import ctypes as C
    from eamm_amd import _lib
    _lib.check(_lib.lib().eamm_op_one_euro(0, C.c_void_p(inplace.data_ptr()), 2048, 40, 0.05, 8.0, 1.0, 100.0, 10.0,
                                           C.c_void_p(inplace.data_ptr()), None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), None)
    assert torch.equal(inplace, dev)
    # device time of the clip's two filter launches (values [2048,20] + jacobians [2048,40]), HIP events on the launch stream, best of
    # ten (a wall-clock bound here was flaky: the host's jitter is several milliseconds on a busy box)
    xv, xj = seq[:, :, 0].contiguous().to(DEV), seq.to(DEV)
    best = float("inf")
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one_euro_smooth(xv, **kw)
        one_euro_smooth(xj, **kw)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"one-euro, 2048 frames, values + jacobians on the device: {best:.3f} ms (HIP events, best of 10)")
    assert best <= 2.0, best
    assert dev.is_cuda and one_euro_smooth(seq[:0].to(DEV), **kw).shape[0] == 0


def test_whole_chain_with_15_key_points():
    """Round 6: make_animation_smooth whole at num_kp = 15 -- the detectors' heads span 128 logit columns there, so KPDetector_a is NOT in
    the wide + thin form and the tail's split hand-over falls back to the reference's NCHW tensor (SplitFeatureMap.to_nchw) inside
    driving_keypoints -- key points and frames against the oracle's chain."""
    cfg = {**hot_path_config(), "num_kp": 15}
    sd = synthetic_state_dict(cfg, seed=1234)
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(sd, strict=True)
    cfg_k, cfg_a = {**kp_detector_config(), "num_kp": 15}, {**kp_detector_a_config(), "num_kp": 15}
    sd_k, sd_a = trained_like_kp_state_dict(cfg_k, 78), trained_like_kp_state_dict(cfg_a, 77)
    sd_d = synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec())
    kp, kpa, tail = KPDetector(**cfg_k), KPDetector_a(**cfg_a), DeconvTail()
    kp.load_state_dict(sd_k, strict=True)
    kpa.load_state_dict(sd_a, strict=True)
    tail.load_state_dict(sd_d, strict=True)
    gen, kp, tail, kpa = [m.to(DEV).eval() for m in (gen, kp, tail, kpa)]
    T = 9
    src, feats = synthetic_source(256, seed=1), synthetic_lstm_features(T, seed=5)
    f32, span, kps = animate_from_features(gen, kp, tail, kpa, src, feats, batch=4, front_batch=5, uint8=False, to_host=True,
                                           return_keypoints=True)
    assert span == (0, T) and kpa.accepts_split(64, 64) == 0 and tail.split_channels() == 32
    kp_s, norm, raw, smooth = orc.animation_keypoints(sd_k, cfg_k, sd_d, sd_a, cfg_a, src, feats)
    nv, nj = torch.cat([n["value"] for n in norm]), torch.cat([n["jacobian"] for n in norm])
    for name, got, want in (("raw", kps["kp_driving_raw"], raw), ("smoothed", kps["kp_driving_smoothed"], smooth),
                            ("normalised", kps["kp_norm"], {"value": nv, "jacobian": nj})):
        ev = float((got["value"].cpu() - want["value"]).abs().max())
        ej = float((got["jacobian"].cpu() - want["jacobian"]).abs().max())
        print(f"\nK = 15 {name:11s} value {ev:.2e}  jacobian {ej:.2e}")
        assert ev <= 5e-5 and ej <= 2e-4, (name, ev, ej)
    pick = [0, 4, 8]
    srcb = src.expand(len(pick), -1, -1, -1).contiguous()
    ksb = {k: v.cpu().expand(len(pick), *v.shape[1:]).contiguous() for k, v in kps["kp_source"].items()}
    with torch.no_grad():
        ref = orc.generator_forward(sd, cfg, srcb, {k: v.cpu()[pick] for k, v in kps["kp_norm"].items()}, ksb)["prediction"]
    assert float((f32[pick] - ref).abs().max()) <= TOL["prediction"]
