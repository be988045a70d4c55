#!/usr/bin/env python
"""The reference caller's shape (demo.py:251-281): ONE driving frame per forward call, source cached -- `frames` calls of the
engine back to back (no D2H), for a rocprofv3 kernel trace of the one-frame launch sequence.  tools/one_frame_loop.py [frames]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = hot_path_config()
gen = OcclusionAwareGenerator(**cfg); gen.load_state_dict(synthetic_state_dict(cfg)); gen = gen.cuda().eval()
eng = gen.encode_source(synthetic_source(256).cuda(), max_frames=1)
kp_s = {k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=0).items()}
kps = [{k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=2 + t).items()} for t in range(T)]
for t in range(8): eng.forward_frames(kps[t], kp_s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T): out = eng.forward_frames(kps[t], kp_s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / T
print(f"one frame per call, engine, no D2H: {dt*1e3:.3f} ms per frame ({T} frames)")
