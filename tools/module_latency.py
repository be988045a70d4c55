#!/usr/bin/env python
"""Per-call latency of the drop-in module in the reference's loop shape (demo.py:251-281): one generator(source,
kp_source, kp_driving) call per frame, prediction copied to the host each frame."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict

torch.set_grad_enabled(False)     # demo.py:195 runs the frame loop under torch.no_grad()
cfg = hot_path_config()
sd = synthetic_state_dict(cfg)
for cache in (False, True):
    gen = OcclusionAwareGenerator(**cfg, cache_source=cache); gen.load_state_dict(sd); gen = gen.cuda().eval()
    src = synthetic_source(256).cuda()
    kp_s = {k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=0).items()}
    kps = [{k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=2 + t).items()} for t in range(128)]
    for t in range(32): gen(src, kp_source=kp_s, kp_driving=kps[t])       # allocator / clock warm-up
    passes = []
    for rep in range(5):                                                   # five passes of 128 frames: median and best
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(128):
            out = gen(src, kp_source=kp_s, kp_driving=kps[t])
            frame = out["prediction"].data.cpu().numpy()
        passes.append((time.perf_counter() - t0) / 128)
    passes.sort()
    dt = passes[len(passes) // 2]
    print(f"module forward per frame incl. D2H, cache_source={cache}: {dt*1e3:.3f} ms = {1/dt:.0f} frames/s "
          f"(median of 5 passes of 128 frames; best {passes[0]*1e3:.3f} ms)")
