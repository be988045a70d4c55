#!/usr/bin/env python
"""Throughput of the key-point detectors on MI355X (row N1): KPDetector_a (per driving frame, demo.py:219) at a
given batch and KPDetector (once per clip, demo.py:206); CPU oracle timed beside them on a bounded sample."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eamm_amd import KPDetector, KPDetector_a, kp_detector_a_config, kp_detector_config  # noqa: E402
from eamm_amd.weights import kp_state_dict_spec, synthetic_source, synthetic_state_dict  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    from oracle import eamm_oracle as orc
    acfg, kcfg = kp_detector_a_config(), kp_detector_config()
    asd = synthetic_state_dict(acfg, seed=77, spec=kp_state_dict_spec(acfg))
    ksd = synthetic_state_dict(kcfg, seed=77, spec=kp_state_dict_spec(kcfg))
    kpa = KPDetector_a(**acfg, max_batch=B); kpa.load_state_dict(asd); kpa = kpa.cuda().eval()
    kp = KPDetector(**kcfg, max_batch=B); kp.load_state_dict(ksd); kp = kp.cuda().eval()
    fmap = torch.randn(B, 35, 64, 64, generator=torch.Generator().manual_seed(0))
    img = synthetic_source(256, seed=3, batch=B)
    dfm, dimg = fmap.cuda(), img.cuda()
    ta = timed(lambda: kpa(dfm), 50)
    tk = timed(lambda: kp(dimg), 50)
    t1 = timed(lambda: kp(dimg[:1]), 50)
    torch.set_num_threads(16)
    with torch.no_grad():
        orc.kp_detector_a_forward(asd, acfg, fmap[:1]); t0 = time.perf_counter()
        for i in range(B): orc.kp_detector_a_forward(asd, acfg, fmap[i:i + 1])
        ca = (time.perf_counter() - t0) / B
        orc.kp_detector_forward(ksd, kcfg, img[:1]); t0 = time.perf_counter()
        for i in range(4): orc.kp_detector_forward(ksd, kcfg, img[i:i + 1])
        ck = (time.perf_counter() - t0) / 4
    print(f"KPDetector_a heads, batch {B}: {ta*1e3:.3f} ms/call = {B/ta:.0f} frames/s   (CPU oracle 16 thr: {1/ca:.1f} frames/s)")
    print(f"KPDetector (image->kp), batch {B}: {tk*1e3:.3f} ms/call = {B/tk:.0f} images/s; batch 1: {t1*1e3:.3f} ms"
          f"   (CPU oracle 16 thr: {1/ck:.1f} images/s)")


if __name__ == "__main__":
    main()
