#!/bin/bash
# round 5, call g: the two committed bench lines again, now carrying the PMC record of the kernel source they ran; the 128-frame plan test
mkdir -p gpurun_out/r05_g
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_plan64.py tests/test_gpu_bench.py -x -q -m gpu -s > gpurun_out/r05_g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_g/pytest.log; grep -n "128 frames" gpurun_out/r05_g/pytest.log
timeout 600 python bench.py > gpurun_out/r05_g/bench_256_b16.log 2>&1; grep '^{' gpurun_out/r05_g/bench_256_b16.log > gpurun_out/r05_g/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > gpurun_out/r05_g/bench_512_b8.log 2>&1; grep '^{' gpurun_out/r05_g/bench_512_b8.log > gpurun_out/r05_g/bench_512_b8.json
python - <<'P'
import json
for n in ("256_b16","512_b8"):
    d=json.load(open(f"gpurun_out/r05_g/bench_{n}.json"))
    print(n, d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "clip", d["clip"]["frames_per_s"], d["clip"]["batch"], (d.get("e2e_clip") or {}).get("frames_per_s"))
P
