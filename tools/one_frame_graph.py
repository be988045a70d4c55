#!/usr/bin/env python
"""One frame per call (BASELINE configs[1]): eager launches vs one HIP-graph replay per frame, with and without a host synchronisation
per frame.  tools/micro/launch_floor.hip shows that inside a replayed graph a small dependent kernel costs ~1.6 - 1.9 us where an eager
launch costs ~3 us -- does the one-frame pass (~75 dependent launches) see that?   tools/one_frame_graph.py [frames]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = hot_path_config()
gen = OcclusionAwareGenerator(**cfg); gen.load_state_dict(synthetic_state_dict(cfg)); gen = gen.cuda().eval()
eng = gen.encode_source(synthetic_source(256).cuda(), max_frames=1)
kp_s = {k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=0).items()}
static = {k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=2).items()}
for _ in range(8): eng.forward_frames(static, kp_s)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): eng.forward_frames(static, kp_s)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    g_out = eng.forward_frames(static, kp_s)["prediction"]
graph.replay(); torch.cuda.synchronize()
ref = eng.forward_frames(static, kp_s)["prediction"]
assert torch.equal(ref, g_out)
for sync in (False, True):
    for mode in ("eager", "graph"):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for t in range(T):
                if mode == "graph": graph.replay()
                else: eng.forward_frames(static, kp_s)
                if sync: torch.cuda.current_stream().synchronize()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / T)
        print(f"one frame per call, {mode:5s}, {'host sync per frame' if sync else 'frames back to back':20s}: {best*1e3:.3f} ms per frame")
