#!/bin/bash
# round 6, call t: deeper LDS-DMA ring for the narrow (one-frame) bottleneck GEMM
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_t; mkdir -p $O; cd $R
EAMM_TUNING=1 EAMM_WINO4_NARROW_RING=3 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wino4" 2>&1 | tail -2
for ring in 2 3 2 3; do EAMM_TUNING=1 EAMM_WINO4_NARROW_RING=$ring timeout 120 python tools/one_frame_loop.py 256 2>&1 | grep -v amdgpu.ids | sed "s/^/ring $ring: /" | tee -a $O/one_frame_ring.txt; done
cd /tmp && export TMPDIR=/tmp
EAMM_TUNING=1 EAMM_WINO4_NARROW_RING=3 rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 -- python $R/tools/one_frame_loop.py 64 > $O/kt1.log 2>&1
cd $R; python tools/rocpd_summary.py $O/kt1/kt1_results.db > $O/one_frame_kernel_trace_stats_ring3.txt 2>&1; rm -rf $O/kt1
head -6 $O/one_frame_kernel_trace_stats_ring3.txt | cut -c1-130
