#!/usr/bin/env python
"""Experiment: what the device-to-host delivery costs the end-to-end path -- animate_from_features over 2048 frames with uint8 frames delivered to
pinned host memory (copies on a copy stream) against the same clip with the frames left on the device; HSA_ENABLE_SDMA as given in the environment."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eamm_amd import (DeconvTail, EngineBackend, KPDetector, KPDetector_a, OcclusionAwareGenerator, animate_from_features, hot_path_config,
                      kp_detector_a_config, kp_detector_config)
from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, synthetic_source, synthetic_state_dict, trained_like_kp_state_dict
torch.set_grad_enabled(False)
T = 2048; dev = "cuda:0"
cfg, ck, ca = hot_path_config(), kp_detector_config(), kp_detector_a_config()
g = OcclusionAwareGenerator(**cfg, max_frames=128); kd, ka, de = KPDetector(**ck), KPDetector_a(**ca), DeconvTail()
g.load_state_dict(synthetic_state_dict(cfg, seed=1234)); kd.load_state_dict(trained_like_kp_state_dict(ck, 78)); ka.load_state_dict(trained_like_kp_state_dict(ca, 77))
de.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()))
g, kd, ka, de = [m.to(dev).eval() for m in (g, kd, ka, de)]
be = EngineBackend(g, batch=128)
src = synthetic_source(256, seed=1).to(dev); feats = synthetic_lstm_features(T, seed=5).to(dev)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
for to_host in (True, False, True, False):
    def run():
        return animate_from_features(g, kd, de, ka, src, feats, batch=128, uint8=True, to_host=to_host, backend=be)
    run(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f, _ = run(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"to_host={to_host}: {T / best:.1f} frames/s ({best * 1e3:.1f} ms)")
    del f
