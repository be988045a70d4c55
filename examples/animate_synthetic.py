#!/usr/bin/env python
"""The reference's `make_animation_smooth` (demo.py:194-282) on the MI355X path, end to end, with synthetic weights and inputs -- what a
maintainer's `demo.py` looks like after the swap of INTEGRATION.md section 2 (real use: `load_state_dict(checkpoint[...])` instead of the
seeded weights, the audio network's LSTM output instead of `synthetic_lstm_features`).

    python examples/animate_synthetic.py [frames] [out.npy]

Two forms of the same clip:
  1. the reference's loop shape, one `generator(source, kp_source=..., kp_driving=...)` call per frame (demo.py:251-281), unchanged;
  2. `eamm_amd.animate_from_features`: the whole function as one call (detectors batched, smoothing on the device, source encoded once,
     frames batched, uint8 frames in pinned host memory) -- and checks that both give the same frames to one uint8 level."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import (DeconvTail, KPDetector, KPDetector_a, OcclusionAwareGenerator, animate_from_features, hot_path_config,  # noqa: E402
                      kp_detector_a_config, kp_detector_config, normalize_kp, smooth_keypoints)
from eamm_amd.weights import (deconv_state_dict_spec, synthetic_lstm_features, synthetic_source, synthetic_state_dict,  # noqa: E402
                              trained_like_kp_state_dict)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    dev = "cuda:0"
    cfg, ck, ca = hot_path_config(), kp_detector_config(), kp_detector_a_config()
    # demo.py:54-95: build the modules from the YAML's kwargs, load the checkpoints (here: seeded synthetic weights), .cuda().eval()
    generator = OcclusionAwareGenerator(**cfg)
    kp_detector, kp_detector_a, decon = KPDetector(**ck), KPDetector_a(**ca), DeconvTail()
    generator.load_state_dict(synthetic_state_dict(cfg, seed=1234), strict=True)
    kp_detector.load_state_dict(trained_like_kp_state_dict(ck, 78), strict=True)
    kp_detector_a.load_state_dict(trained_like_kp_state_dict(ca, 77), strict=True)
    decon.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()), strict=True)
    generator, kp_detector, kp_detector_a, decon = [m.to(dev).eval() for m in (generator, kp_detector, kp_detector_a, decon)]
    source = synthetic_source(256, seed=1).to(dev)                  # [1,3,256,256] in [0,1]
    lstm_out = synthetic_lstm_features(T, seed=5).to(dev)           # [T,256]: AT_net2's LSTM output, one row per frame

    def reference_loop():
        """demo.py:206-281 as written there: every module called once per frame."""
        kp_source = kp_detector(source)                                                          # demo.py:206
        raw = [kp_detector_a(decon(lstm_out[t:t + 1])) for t in range(T)]                        # util.py:603-607 + demo.py:219
        seq = smooth_keypoints({k: torch.cat([r[k] for r in raw]) for k in ("value", "jacobian")})   # demo.py:241-250 (One-Euro)
        initial = {k: raw[0][k] for k in ("value", "jacobian")}                                   # demo.py:207
        frames = []
        for t in range(T):
            kp_norm = normalize_kp(kp_source, {k: v[t:t + 1] for k, v in seq.items()}, initial, adapt_movement_scale=True,
                                   use_relative_movement=True, use_relative_jacobian=True)      # demo.py:276
            out = generator(source, kp_source=kp_source, kp_driving=kp_norm)                    # demo.py:279
            frames.append(np.transpose(out["prediction"].data.cpu().numpy(), [0, 2, 3, 1])[0])   # demo.py:281
        return frames

    with torch.no_grad():
        # ---- 1. the reference's structure, per frame (first pass: the modules build their library handles and pack the weights)
        reference_loop()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frames_loop = reference_loop()
        torch.cuda.synchronize(); t_loop = time.perf_counter() - t0
        # ---- 2. the same function as one call
        animate_from_features(generator, kp_detector, decon, kp_detector_a, source, lstm_out)     # warm-up (handles, pinned buffer)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frames_u8, span = animate_from_features(generator, kp_detector, decon, kp_detector_a, source, lstm_out)
        torch.cuda.synchronize(); t_clip = time.perf_counter() - t0
    loop_u8 = np.clip(np.rint(np.stack(frames_loop) * 255), 0, 255)
    worst = float(np.abs(loop_u8 - frames_u8.numpy().astype(np.float64)).max())
    print(f"{T} frames at 256x256: per-frame loop {t_loop * 1e3:.1f} ms ({T / t_loop:.0f} frames/s), one call {t_clip * 1e3:.1f} ms "
          f"({T / t_clip:.0f} frames/s); uint8 levels between the two: {worst:.0f}")
    assert span == (0, T) and frames_u8.shape == (T, 256, 256, 3) and worst <= 1
    if out_path:
        np.save(out_path, frames_u8.numpy())
        print("frames written to", out_path)


if __name__ == "__main__":
    main()
