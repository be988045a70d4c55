#!/bin/bash
# round 5, call aa: GEMM variant 3 vs 6 (non-temporal V loads) on the CLIP and END-TO-END legs (four chains of 32 frames per call), same box, alternating
mkdir -p gpurun_out/r05_aa
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-frames 0 --train-pairs 0 > gpurun_out/r05_aa/$name.json 2> gpurun_out/r05_aa/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_aa/{n}.json")); e=d.get("e2e_clip") or {}
    print(f"{n:6s} value {d['value']:8.1f}  clip {d['clip']['frames_per_s']:8.1f}  e2e {e.get('frames_per_s')}  frac {d['roofline']['frac']:.3f}")
except Exception as ex:
    print(n, "FAILED", ex, open(f"gpurun_out/r05_aa/{n}.err").read()[-400:])
P
}
run v3a EAMM_WINO4_VARIANT=3
run v6a EAMM_WINO4_VARIANT=6
run v3b EAMM_WINO4_VARIANT=3
run v6b EAMM_WINO4_VARIANT=6
