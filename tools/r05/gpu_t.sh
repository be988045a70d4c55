#!/bin/bash
# round 5, call t: the round-4 stagger knob again on top of the XCD-aware transform
mkdir -p gpurun_out/r05_t
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_t/$name.json 2> gpurun_out/r05_t/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
d=json.load(open(f"gpurun_out/r05_t/{n}.json")); s=d["stage_ms_per_step"]; r=d["roofline"]
print(f"{n:10s} {d['value']:8.1f} f/s  T {s['bneck_transform']:.3f} G {s['bneck_conv']:.3f}  union {r['bneck_union_ms_per_step']:.3f} frac {r['frac']:.3f} T/launch {r['avg_input_transform_ms']*1e3:.1f}us")
P
}
run base A=1
run stagger EAMM_BNECK_STAGGER=1
run base2 A=1
run stagger2 EAMM_BNECK_STAGGER=1
