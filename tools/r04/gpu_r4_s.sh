#!/bin/bash
# kernel trace of the training step (8 pairs, 6 steps incl. warm-up) after the saved-transform change + the new ABI test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_s; mkdir -p $O
python -m pytest tests/test_gpu_backward.py -q -x -m gpu -k "saved or wgrad" 2>&1 | tail -3 > $O/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/train_step_bench.py 8 5 > $O/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/train_step_kernel_trace_stats.txt 2>&1
rm -rf $O/kt
cat $O/tests.txt; tail -2 $O/kt.log; head -50 $O/train_step_kernel_trace_stats.txt | cut -c1-150
