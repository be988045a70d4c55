#!/bin/bash
# round 6, call h: pixel-sliced kp head + vectorised split epilogue: parity, front-end time and per-kernel table, e2e leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_deconv_tail.py tests/test_kp_detector.py tests/test_gpu_pipeline.py tests/test_gpu_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/front_bench.py 2048 64 > $O/kt.log 2>&1
cd $R; python tools/rocpd_summary.py $O/kt/kt_results.db > $O/front_kernel_trace_stats.txt 2>&1; rm -rf $O/kt
grep "front end\|bit for bit" $O/kt.log; head -16 $O/front_kernel_trace_stats.txt | cut -c1-140
timeout 300 python tools/front_bench.py 2048 64 2>&1 | grep -v amdgpu.ids | tee $O/front_bench.txt
timeout 300 python tools/front_bench.py 2048 128 2>&1 | grep -v amdgpu.ids | tee -a $O/front_bench.txt
timeout 600 python bench.py --cpu-frames 0 --clip-frames 0 --train-pairs 0 > $O/bench_e2e.log 2>&1; grep '^{' $O/bench_e2e.log > $O/bench_e2e.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_h/bench_e2e.json'))
print(d['value'], d['e2e_clip']['frames_per_s'], d['e2e_clip']['phases_ms_rank0'], d['e2e_clip']['verify'])
P
