#!/bin/bash
# round 5, call m: KPDetector_a wide + thin heads -- parity, then the front end and the end-to-end leg
mkdir -p gpurun_out/r05_m
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kp_detector.py tests/test_deconv_tail.py tests/test_gpu_pipeline.py -x -q -m gpu -s > gpurun_out/r05_m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_m/pytest.log; grep -n "^kpa" gpurun_out/r05_m/pytest.log
for thin in 1 0; do
EAMM_KPA_THIN=$thin python - <<'P'
import os, time, torch, sys
sys.path.insert(0, '.')
from eamm_amd import DeconvTail, KPDetector_a, driving_keypoints, kp_detector_a_config
from eamm_amd.weights import deconv_state_dict_spec, synthetic_lstm_features, synthetic_state_dict, trained_like_kp_state_dict
ca = kp_detector_a_config()
kpa, tail = KPDetector_a(**ca), DeconvTail()
kpa.load_state_dict(trained_like_kp_state_dict(ca, 77)); tail.load_state_dict(synthetic_state_dict(None, seed=3, spec=deconv_state_dict_spec()))
kpa, tail = kpa.cuda().eval(), tail.cuda().eval()
f = synthetic_lstm_features(2048).cuda()
for fb in (64, 128):
    driving_keypoints(tail, kpa, f, batch=fb); torch.cuda.synchronize()
    t0 = time.perf_counter(); driving_keypoints(tail, kpa, f, batch=fb); torch.cuda.synchronize()
    print(f"EAMM_KPA_THIN={os.environ['EAMM_KPA_THIN']}: front end, 2048 frames, {fb} per call: {(time.perf_counter() - t0) * 1e3:.1f} ms")
P
done
timeout 400 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --clip-frames 0 --train-pairs 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        e=json.loads(l)['e2e_clip']; print('e2e', e['frames_per_s'], e['verify'], e['phases_ms_rank0'])"
