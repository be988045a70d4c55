#!/bin/bash
# round 4, call b: all GPU tests (no -x), MFMA 4x4x1 probe, bench with the new final-layer epilogue, profile with graph replay
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export ROUND=r04
O=$R/gpurun_out/r04_b; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -15
timeout 120 tools/micro/mfma4x4_probe > $O/mfma4x4_probe.txt 2>&1; cat $O/mfma4x4_probe.txt
timeout 600 python bench.py --graph > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; python - <<PY
import json
d=json.load(open("$O/bench_256_b16.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("graph"), d["stage_ms_per_step"], d["train_step"].get("checks"))
PY
timeout 900 bash tools/gpu_profile.sh 256 16 r04b_256_b16 > $O/profile_256.log 2>&1; tail -4 $O/profile_256.log
cat $R/gpurun_out/prof_r04b_256_b16/bneck_timeline.txt | head -12
