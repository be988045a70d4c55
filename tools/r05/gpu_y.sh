#!/bin/bash
# round 5, call y: non-temporal V LOADS in the bottleneck GEMM (EAMM_WINO4_VARIANT=6 = variant 3 + nt on the V stream)
mkdir -p gpurun_out/r05_y
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_y/$name.json 2> gpurun_out/r05_y/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_y/{n}.json")); s=d["stage_ms_per_step"]; r=d["roofline"]
    print(f"{n:8s} {d['value']:8.1f} f/s  parity {d['parity_check']['max_abs_err']:.2e}  T {s['bneck_transform']:.3f} G {s['bneck_conv']:.3f} union {r['bneck_union_ms_per_step']:.3f} frac {r['frac']:.3f} G/launch {r['per_launch']['avg_launch_ms']*1e3:.1f}us T/launch {r['avg_input_transform_ms']*1e3:.1f}us")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r05_y/{n}.err").read()[-400:])
P
}
run v7 EAMM_WINO4_VARIANT=7
run v7 EAMM_WINO4_VARIANT=7
run v8 EAMM_WINO4_VARIANT=8
run v6b EAMM_WINO4_VARIANT=6
