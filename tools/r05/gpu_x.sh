#!/bin/bash
# round 5, call x: non-temporal output stores in the up-block kernel (EAMM_PATCH_NT, experiment)
mkdir -p gpurun_out/r05_x
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  env EAMM_TUNING=1 "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_x/$name.json 2> gpurun_out/r05_x/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_x/{n}.json")); s=d["stage_ms_per_step"]
    print(f"{n:8s} {d['value']:8.1f} f/s  parity {d['parity_check']['max_abs_err']:.2e}  up {s['up']:.3f} final {s['final']:.3f} hg_dec {s['hg_dec']:.3f} bneck {s['bneck_transform']+s['bneck_conv']:.3f}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r05_x/{n}.err").read()[-400:])
P
}
run base EAMM_PATCH_NT=0
run nt EAMM_PATCH_NT=1
run base2 EAMM_PATCH_NT=0
run nt2 EAMM_PATCH_NT=1
