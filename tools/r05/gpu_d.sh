#!/bin/bash
# round 5, call d (final): whole GPU suite, default line, 512x512 line, module latency / backward / train-step tools, profiles at both
# sizes (tools/gpu_round.sh), then two sweeps of the clip legs: frames per call of the clip leg, frames per call of the front end
cd $GRAFT_REPO_ROOT
ROUND=r05 bash tools/gpu_round.sh d
O=gpurun_out/r05_d
for cb in 32 128; do
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --train-pairs 0 --e2e-frames 0 --clip-batch $cb 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('clip-batch $cb', d['clip']['frames_per_s'], d['clip']['plan']['pass_chains'], d['clip']['verify']['ok'])" | tee -a $O/sweeps.txt
done
python - <<'P' | tee -a gpurun_out/r05_d/sweeps.txt
import time, torch, sys
sys.path.insert(0, '.')
from eamm_amd import DeconvTail, KPDetector_a, driving_keypoints, kp_detector_a_config
from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, synthetic_state_dict, trained_like_kp_state_dict
ca = kp_detector_a_config()
kpa, tail = KPDetector_a(**ca), DeconvTail()
kpa.load_state_dict(trained_like_kp_state_dict(ca, 77)); tail.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()))
kpa, tail = kpa.cuda().eval(), tail.cuda().eval()
f = synthetic_lstm_features(2048).cuda()
for fb in (16, 32, 64, 128, 256):
    driving_keypoints(tail, kpa, f, batch=fb); torch.cuda.synchronize()
    t0 = time.perf_counter(); driving_keypoints(tail, kpa, f, batch=fb); torch.cuda.synchronize()
    print(f'front end, 2048 frames, {fb} per call: {(time.perf_counter() - t0) * 1e3:.1f} ms')
P
