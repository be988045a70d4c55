#!/bin/bash
# round 5, call ab: the GEMM variant chosen per call (6 while the call's GEMM workgroups fit the CUs, 3 above): plan tests + the line's legs
mkdir -p gpurun_out/r05_ab
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_plan64.py tests/test_gpu_bench.py -x -q > gpurun_out/r05_ab/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r05_ab/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --cpu-frames 0 --train-pairs 0 > gpurun_out/r05_ab/d$i.json 2> gpurun_out/r05_ab/d$i.err
  python - d$i <<'P'
import json,sys
n=sys.argv[1]
d=json.load(open(f"gpurun_out/r05_ab/{n}.json")); e=d.get("e2e_clip") or {}
print(f"{n:6s} value {d['value']:8.1f}  clip {d['clip']['frames_per_s']:8.1f}  e2e {e.get('frames_per_s')}  frac {d['roofline']['frac']:.3f} plan16 v{d['knobs']['plan'].get('wino4_variant')} clip plan v{d['clip']['plan'].get('wino4_variant')}")
P
done
