#!/bin/bash
# round 6, call a: baseline of the inherited tree on this round's box + kernel trace of the ONE-FRAME launch sequence
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; cut -c1-500 $O/bench_256_b16.json
timeout 300 python tools/module_latency.py 2>&1 | grep -v amdgpu.ids > $O/module_latency.txt; cat $O/module_latency.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 -- python $R/tools/one_frame_loop.py 64 > $O/kt1.log 2>&1
cd $R
python tools/rocpd_summary.py $O/kt1/kt1_results.db > $O/one_frame_kernel_trace_stats.txt 2>&1
rm -rf $O/kt1
head -70 $O/one_frame_kernel_trace_stats.txt
