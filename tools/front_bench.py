#!/usr/bin/env python
"""The clip's front end alone (reference util.py:604-607 + demo.py:219: DeconvTail + KPDetector_a for every frame), `batch` frames per
call over a 2048-frame clip, in both hand-over forms: the reference's NCHW tensor between the two modules and the private split NHWC
form (round 6).  tools/front_bench.py [frames] [batch]   (run it under rocprofv3 --kernel-trace for the per-kernel table)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import DeconvTail, KPDetector_a, driving_keypoints, kp_detector_a_config
from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, synthetic_state_dict, trained_like_kp_state_dict
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ca = kp_detector_a_config()
kpa, tail = KPDetector_a(**ca), DeconvTail()
kpa.load_state_dict(trained_like_kp_state_dict(ca, 77), strict=True)
tail.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()), strict=True)
kpa, tail = kpa.cuda().eval(), tail.cuda().eval()
feats = synthetic_lstm_features(T, seed=5).cuda()


class NchwTail:      # the reference's hand-over: hides forward_split from driving_keypoints
    def __init__(self, m): self.m = m
    def __call__(self, x): return self.m(x)


res = {}
for name, t in (("nchw", NchwTail(tail)), ("split", tail)):
    for _ in range(2):
        out = driving_keypoints(t, kpa, feats, batch=B)
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); out = driving_keypoints(t, kpa, feats, batch=B); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    res[name] = out
    print(f"front end, {T} frames, {B} per call, hand-over {name}: {best*1e3:.2f} ms ({best/T*1e6:.2f} us per frame)")
print("split == nchw bit for bit:", all(torch.equal(res["nchw"][k], res["split"][k]) for k in res["nchw"]))
