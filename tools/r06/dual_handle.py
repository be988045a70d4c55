#!/usr/bin/env python
"""Experiment: a clip's 128-frame calls alternating between TWO engine handles on two streams (two workspaces, calls overlap freely: the
hourglass phase of one call beside the bottleneck / decoder of the other) against one handle, calls back to back."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
torch.set_grad_enabled(False)
T, CB = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = hot_path_config(); sd = synthetic_state_dict(cfg, seed=1234); dev = torch.device("cuda:0")
gens = []
for _ in range(2):
    g = OcclusionAwareGenerator(**cfg, max_frames=CB); g.load_state_dict(sd); gens.append(g.to(dev).eval())
src = synthetic_source(256, seed=1).to(dev)
engs = [g.encode_source(src, max_frames=CB) for g in gens]
kp_s = {k: v.to(dev) for k, v in synthetic_keypoints(1, 10, seed=0).items()}
kp_d = {k: v.to(dev) for k, v in synthetic_keypoints(T, 10, seed=2).items()}
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(dual):
    outs = []
    for i, s0 in enumerate(range(0, T, CB)):
        j = (i & 1) if dual else 0
        with torch.cuda.stream(streams[j]):
            outs.append(engs[j].forward_frames({k: v[s0:s0 + CB] for k, v in kp_d.items()}, kp_s, outputs=("prediction",))["prediction"])
    torch.cuda.synchronize()
    return outs
for dual in (False, True, False, True):
    run(dual); best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = run(dual); best = min(best, time.perf_counter() - t0)
    print(f"{'two handles, two streams' if dual else 'one handle, one stream  '}: {T / best:.1f} frames/s ({CB} frames per call)")
a, b = run(False), run(True)
print("same frames:", all(torch.equal(x, y) for x, y in zip(a, b)))
