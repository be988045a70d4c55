#!/bin/bash
# round 5, call u: non-temporal stores of V in the input transform (EAMM_WINO4_TR_XCD bit 1), on top of the XCD-aware order (bit 0)
mkdir -p gpurun_out/r05_u
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env EAMM_TUNING=1 "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_u/$name.json 2> gpurun_out/r05_u/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_u/{n}.json")); s=d["stage_ms_per_step"]; r=d["roofline"]
    print(f"{n:8s} {d['value']:8.1f} f/s  parity {d['parity_check']['max_abs_err']:.2e}  T {s['bneck_transform']:.3f} G {s['bneck_conv']:.3f}  union {r['bneck_union_ms_per_step']:.3f} frac {r['frac']:.3f} T/launch {r['avg_input_transform_ms']*1e3:.1f}us G/launch {r['per_launch']['avg_launch_ms']*1e3:.1f}us")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r05_u/{n}.err").read()[-400:])
P
}
run xcd1 EAMM_WINO4_TR_XCD=1
run nt3 EAMM_WINO4_TR_XCD=3
run xcd1b EAMM_WINO4_TR_XCD=1
run nt3b EAMM_WINO4_TR_XCD=3
run nt2 EAMM_WINO4_TR_XCD=2
