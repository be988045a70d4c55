#!/bin/bash
# round 6, call c: KPDetector at num_kp 15 / 30 (wider head), the bench line's new records (all_outputs, latency_b1, cores split)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kp_detector.py tests/test_gpu_bench.py tests/test_deconv_tail.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^(kp_tiny64_k|kpa_tiny_k)|passed|failed|Error|FAILED|assert" $O/pytest.log | head -40
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_c/bench_256_b16.json'))
print(d['value'], d['all_outputs'], d['latency_b1'], {k:d['cpu_baseline'][k] for k in ('value','cores','threads_used','physical_cores','logical_cpus')}, d['e2e_clip']['verify'])
P
