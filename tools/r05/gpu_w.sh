#!/bin/bash
mkdir -p gpurun_out/r05_w
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r05_w/bench_256_b16.log 2>&1; grep '^{' gpurun_out/r05_w/bench_256_b16.log > gpurun_out/r05_w/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > gpurun_out/r05_w/bench_512_b8.log 2>&1; grep '^{' gpurun_out/r05_w/bench_512_b8.log > gpurun_out/r05_w/bench_512_b8.json
python - <<'P'
import json
for n in ("256_b16","512_b8"):
    d=json.load(open(f"gpurun_out/r05_w/bench_{n}.json"))
    e=d.get("e2e_clip") or {}
    print(n, d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "clip", d["clip"]["frames_per_s"], "e2e", e.get("frames_per_s"), e.get("seconds"), e.get("verify"), e.get("phases_ms_rank0"))
P
