#!/bin/bash
# round 6, call f: half-row split of the one-frame bottleneck GEMM (groups 12), grouped kp head kernel, DMA tile for the deconv tail's last
# layer: full GPU suite, one-frame latency, front-end time, launch-floor microbenchmark
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_f; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
tools/micro/launch_floor 2>&1 | grep -v amdgpu.ids | tee $O/launch_floor.txt
timeout 300 python tools/front_bench.py 2048 64 2>&1 | grep -v amdgpu.ids | tee $O/front_bench.txt
timeout 300 python tools/module_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/module_latency.txt
for g in 0 6; do EAMM_TUNING=1 EAMM_WINO4_GROUPS=$g timeout 120 python tools/one_frame_loop.py 256 2>&1 | grep -v amdgpu.ids | sed "s/^/EAMM_WINO4_GROUPS=$g: /" | tee -a $O/one_frame.txt; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 -- python $R/tools/one_frame_loop.py 64 > $O/kt1.log 2>&1
cd $R; python tools/rocpd_summary.py $O/kt1/kt1_results.db > $O/one_frame_kernel_trace_stats.txt 2>&1; rm -rf $O/kt1
head -12 $O/one_frame_kernel_trace_stats.txt | cut -c1-130
