#!/usr/bin/env python
"""One fine-tuning step of the generator at the benchmark configuration (256 x 256, hot_path_config): .train() forward with an
autograd graph (eamm_amd/train_graph.py: HIP convolution / BatchNorm / warp operators) + loss.backward().  Wall clock per step
with the stream drained, forward and backward separately.  Usage: python tools/train_step_bench.py [pairs] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import OcclusionAwareGenerator, hot_path_config  # noqa: E402
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict  # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    size = int(os.environ.get("TRAIN_BENCH_SIZE", "256"))
    dev = torch.device("cuda:0")
    cfg = hot_path_config()
    gen = OcclusionAwareGenerator(**cfg)
    gen.load_state_dict(synthetic_state_dict(cfg, seed=1234), strict=True)
    gen = gen.to(dev).train()
    gen.requires_grad_(True)
    src = synthetic_source(size, seed=1, batch=pairs).to(dev)
    kp_s = {k: v.to(dev) for k, v in synthetic_keypoints(pairs, cfg["num_kp"], seed=0).items()}
    kp_d = {k: v.to(dev).requires_grad_() for k, v in synthetic_keypoints(pairs, cfg["num_kp"], seed=2).items()}
    target = torch.rand(pairs, 3, size, size, device=dev)
    fwd, bwd = [], []
    for it in range(steps + 1):
        for p in gen.parameters():
            p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen(src, kp_driving=kp_d, kp_source=kp_s)
        loss = (out["prediction"] - target).abs().mean()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:   # the first step pays allocator warm-up
            fwd.append((t1 - t0) * 1e3)
            bwd.append((t2 - t1) * 1e3)
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in gen.parameters())))
    print(json.dumps({"op": "train_step", "size": size, "pairs": pairs, "steps": steps, "forward_ms": min(fwd), "backward_ms": min(bwd),
                      "step_ms": min(f + b for f, b in zip(fwd, bwd)), "pairs_per_s": pairs / (min(f + b for f, b in zip(fwd, bwd)) / 1e3),
                      "loss": float(loss.detach()), "grad_norm": gnorm, "finite": bool(torch.isfinite(torch.tensor(gnorm)))}))


if __name__ == "__main__":
    main()
