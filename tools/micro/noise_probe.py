"""How far does ONE cancelling-sum gradient (occlusion.bias, no-jacobian variant, .train()) move under fp32-rounding-sized
perturbations?  (round 4: the HIP motion operators put it 3.9 % from the oracle's double value, the torch composition < 2 %.)"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eamm_amd import OcclusionAwareGenerator, tiny_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
warnings.simplefilter("ignore")
cfg = tiny_config(); n = 3; DEV = "cuda:0"
src = synthetic_source(64, seed=3, batch=n)
kp_s, kp_d = synthetic_keypoints(n, 10, seed=4, jacobian=False), synthetic_keypoints(n, 10, seed=5, jacobian=False)
KEYS = ("mask", "sparse_deformed", "occlusion_map", "deformed", "prediction")
def run(route, eps=0.0):
    os.environ["EAMM_MOTION_TORCH"] = "1" if route == "torch" else "0"
    gen = OcclusionAwareGenerator(**cfg); gen.load_state_dict(synthetic_state_dict(cfg, seed=77)); gen = gen.to(DEV).train()
    s = (src * (1 + eps)).to(DEV)
    out = gen(s, kp_driving={k: v.to(DEV) for k, v in kp_d.items()}, kp_source={k: v.to(DEV) for k, v in kp_s.items()})
    g = torch.Generator().manual_seed(9)
    w = {k: torch.randn(out[k].shape, generator=g) for k in out}
    sum((out[k] * w[k].to(DEV)).sum() for k in out).backward()
    return {k: p.grad.detach().cpu().double() for k, p in gen.named_parameters()}
base = run("torch")
for tag, other in (("hip route", run("hip")), ("torch route, source * (1 + 1e-7)", run("torch", 1e-7)), ("torch route, source * (1 + 3e-7)", run("torch", 3e-7)),
                   ("hip route, source * (1 + 1e-7)", run("hip", 1e-7))):
    k = "dense_motion_network.occlusion.bias"
    worst = max(((float((other[q] - base[q]).abs().max()) / max(float(base[q].abs().max()), 1e-12)), q) for q in base)
    print(f"{tag:36s} occlusion.bias {float(other[k]):+.6f} (torch route {float(base[k]):+.6f})   worst relative change of any tensor {worst[0]:.2e} at {worst[1]}")
