#!/bin/bash
# round 6, call g: per-kernel table of the front end after the grouped head kernel / DMA last layer; graph replay vs eager at one frame per call
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_g; mkdir -p $O; cd $R
timeout 300 python tools/one_frame_graph.py 256 2>&1 | grep -v amdgpu.ids | tee $O/one_frame_graph.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/front_bench.py 2048 64 > $O/kt.log 2>&1
cd $R; python tools/rocpd_summary.py $O/kt/kt_results.db > $O/front_kernel_trace_stats.txt 2>&1; rm -rf $O/kt
grep "front end" $O/kt.log; head -16 $O/front_kernel_trace_stats.txt | cut -c1-140
