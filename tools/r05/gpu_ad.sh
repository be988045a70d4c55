#!/bin/bash
# round 5, call ad (= call z again after the per-call variant rule): profiles at 256 / 512, smoke, the GPU suite, the committed lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_ad; mkdir -p $O
ROUND=r05 timeout 600 bash tools/gpu_profile.sh 256 16 r05ad_256_b16 > $O/profile_256.log 2>&1; tail -3 $O/profile_256.log
ROUND=r05 timeout 600 bash tools/gpu_profile.sh 512 8 r05ad_512_b8 > $O/profile_512.log 2>&1; tail -3 $O/profile_512.log
python tools/collect_profiles.py ad r05 > $O/collect.log 2>&1   # on the box: the bench lines below read the traffic record of THESE kernels
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 400 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json
timeout 400 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > $O/bench_512_b8.log 2>&1; grep '^{' $O/bench_512_b8.log > $O/bench_512_b8.json
python - <<'P'
import json
for n in ("256_b16","512_b8"):
    d=json.load(open(f"gpurun_out/r05_ad/bench_{n}.json"))
    e=d.get("e2e_clip") or {}
    print(n, d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "clip", d["clip"]["frames_per_s"], "e2e", e.get("frames_per_s"), e.get("verify"))
P
