#!/bin/bash
# round 5, call s: final tree -- smoke, whole GPU suite, the two committed bench lines
mkdir -p gpurun_out/r05_s
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_s/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r05_s/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_s/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_s/pytest.log
timeout 600 python bench.py > gpurun_out/r05_s/bench_256_b16.log 2>&1; grep '^{' gpurun_out/r05_s/bench_256_b16.log > gpurun_out/r05_s/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > gpurun_out/r05_s/bench_512_b8.log 2>&1; grep '^{' gpurun_out/r05_s/bench_512_b8.log > gpurun_out/r05_s/bench_512_b8.json
python - <<'P'
import json
for n in ("256_b16","512_b8"):
    d=json.load(open(f"gpurun_out/r05_s/bench_{n}.json"))
    e=d.get("e2e_clip") or {}
    print(n, d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "clip", d["clip"]["frames_per_s"], "e2e", e.get("frames_per_s"), e.get("seconds"), e.get("verify"), e.get("phases_ms_rank0"))
P
