#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_h; mkdir -p $O
python tools/one_frame_loop.py 64 2>&1 | grep -v amdgpu.ids | tee $O/one_frame.txt
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/one_frame_loop.py 64 > $O/kt.log 2>&1
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/one_frame_kernel_trace_stats.txt 2>&1
rm -rf $O/kt
head -70 $O/one_frame_kernel_trace_stats.txt | cut -c1-150
