#!/bin/bash
# round 5, call b: bottleneck sub-chains inside the whole-pass chains (EAMM_BNECK_SUB) and chain-count variants; each line is one
# short bench run (contract region only) whose parity_check must pass
mkdir -p gpurun_out/r05_b
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > gpurun_out/r05_b/$name.json 2> gpurun_out/r05_b/$name.err
  python - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r05_b/{n}.json"))
    s=d["stage_ms_per_step"]
    print(f"{n:28s} {d['value']:8.1f} f/s  parity {d['parity_check']['max_abs_err']:.1e}  bneck {s['bneck_transform']+s['bneck_conv']:.3f} hg {s['hg_enc']+s['hg_dec']:.3f} up {s['up']:.3f} final {s['final']:.3f} head {s['head']:.3f}  union {d['roofline']['bneck_union_ms_per_step']:.3f} frac {d['roofline']['frac']:.3f} plan sub={d['knobs']['plan'].get('bneck_subchains')} pc={d['knobs']['plan']['pass_chains']}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r05_b/{n}.err").read()[-600:])
P
}
run base A=1
run sub2 EAMM_BNECK_SUB=2
run sub2_q8 EAMM_BNECK_SUB=2 GPU_MAX_HW_QUEUES=8
run sub4_q8 EAMM_BNECK_SUB=4 GPU_MAX_HW_QUEUES=8
run sub4 EAMM_BNECK_SUB=4
run pc3 EAMM_PASS_CHAINS=3 EAMM_FINAL_FUSED_MIN_ROWS=64
run pc4 EAMM_PASS_CHAINS=4 EAMM_PASS_CHAINS_MIN_BLOCKS=1 EAMM_FINAL_FUSED_MIN_ROWS=64
run pc4_q8 EAMM_PASS_CHAINS=4 EAMM_PASS_CHAINS_MIN_BLOCKS=1 EAMM_FINAL_FUSED_MIN_ROWS=64 GPU_MAX_HW_QUEUES=8
run base2 A=1
