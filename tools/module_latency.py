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


# ---- what a HIP graph of the one-frame pass would buy: the same loop with the engine's launch sequence
# captured once (static key-point buffers, refreshed by a copy per frame) and replayed -- host launches and inter-kernel
# dispatch gaps are what a graph can remove; kernel time is not
gen = OcclusionAwareGenerator(**cfg); gen.load_state_dict(sd); gen = gen.cuda().eval()
src = synthetic_source(256).cuda()
kp_s = {k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=0).items()}
kps = [{k: v.cuda() for k, v in synthetic_keypoints(1, 10, seed=2 + t).items()} for t in range(128)]
eng = gen.encode_source(src, max_frames=1)
static = {k: v.clone() for k, v in kps[0].items()}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): eng.forward_frames(static, kp_s)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    g_out = eng.forward_frames(static, kp_s)["prediction"]
ref = eng.forward_frames(kps[5], kp_s)["prediction"].clone()
for k in static: static[k].copy_(kps[5][k])
graph.replay(); torch.cuda.synchronize()
assert torch.equal(g_out, ref)
for mode in ("eager engine", "graph replay"):
    passes = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(128):
            if mode == "graph replay":
                for k in static: static[k].copy_(kps[t][k], non_blocking=True)
                graph.replay()
                frame = g_out.cpu().numpy()
            else:
                frame = eng.forward_frames(kps[t], kp_s)["prediction"].cpu().numpy()
        passes.append((time.perf_counter() - t0) / 128)
    passes.sort()
    print(f"one frame per call incl. D2H, {mode}: {passes[2]*1e3:.3f} ms (median of 5 passes of 128 frames; best {passes[0]*1e3:.3f} ms)")
