#!/bin/bash
# round 5, call e: streaming (persistent, double-buffered) input transform -- EAMM_WINO4_TR_STREAM = workgroups per CU -- in the pipeline
mkdir -p gpurun_out/r05_e
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_e/$name.json 2> gpurun_out/r05_e/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_e/{n}.json"))
    s=d["stage_ms_per_step"]; r=d["roofline"]
    print(f"{n:26s} {d['value']:8.1f} f/s  parity {d['parity_check']['max_abs_err']:.2e}  T {s['bneck_transform']:.3f} G {s['bneck_conv']:.3f} hg {s['hg_enc']+s['hg_dec']:.3f} up {s['up']:.3f}  union {r['bneck_union_ms_per_step']:.3f} frac {r['frac']:.3f} T/launch {r['avg_input_transform_ms']*1e3:.1f}us G/launch {r['per_launch']['avg_launch_ms']*1e3:.1f}us")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r05_e/{n}.err").read()[-600:])
P
}
run base A=1
run s1_50 EAMM_WINO4_TR_STREAM=1
run s2_50 EAMM_WINO4_TR_STREAM=2
run s2_100 EAMM_WINO4_TR_STREAM=2 EAMM_WINO4_TR_STREAM_CU_PCT=100
run s1_100 EAMM_WINO4_TR_STREAM=1 EAMM_WINO4_TR_STREAM_CU_PCT=100
run s2_50_stag EAMM_WINO4_TR_STREAM=2 EAMM_BNECK_STAGGER=1
run s2_100_stag EAMM_WINO4_TR_STREAM=2 EAMM_WINO4_TR_STREAM_CU_PCT=100 EAMM_BNECK_STAGGER=1
run base_stag EAMM_BNECK_STAGGER=1
run s2_25 EAMM_WINO4_TR_STREAM=2 EAMM_WINO4_TR_STREAM_CU_PCT=25
run base2 A=1
