#!/bin/bash
# round 6, call e: split NHWC hand-over DeconvTail -> KPDetector_a: parity, front-end time, per-kernel table of the front end alone
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_deconv_tail.py tests/test_kp_detector.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/front_bench.py 2048 64 2>&1 | grep -v amdgpu.ids | tee $O/front_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/front_bench.py 2048 64 > $O/kt.log 2>&1
cd $R; python tools/rocpd_summary.py $O/kt/kt_results.db > $O/front_kernel_trace_stats.txt 2>&1; rm -rf $O/kt
head -30 $O/front_kernel_trace_stats.txt | cut -c1-140
