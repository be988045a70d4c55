#!/bin/bash
# round 5, call ac: GEMM variant pinned 3 / pinned 6 / chosen per call (default), alternating on one box, three rounds
mkdir -p gpurun_out/r05_ac
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-frames 0 --train-pairs 0 > gpurun_out/r05_ac/$name.json 2> gpurun_out/r05_ac/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_ac/{n}.json")); e=d.get("e2e_clip") or {}
    print(f"{n:6s} value {d['value']:8.1f}  clip {d['clip']['frames_per_s']:8.1f}  e2e {e.get('frames_per_s')}  frac {d['roofline']['frac']:.3f}  variants {d['knobs']['plan'].get('wino4_variant')}/{d['clip']['plan'].get('wino4_variant')}")
except Exception as ex:
    print(n, "FAILED", ex, open(f"gpurun_out/r05_ac/{n}.err").read()[-400:])
P
}
for r in 1 2 3; do
  run v3_$r EAMM_WINO4_VARIANT=3
  run v6_$r EAMM_WINO4_VARIANT=6
  run def_$r EAMM_UNUSED=0
done
