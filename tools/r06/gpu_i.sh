#!/bin/bash
# round 6, call i: training-step timeline by stage, the >= 200-step contract line, the profile set incl. the per-launch roofline derivation
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_i; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/train_step_bench.py 8 6 > $O/train_kt.log 2>&1
cd $R; python tools/rocpd_summary.py --train-timeline $O/kt/kt_results.db > $O/train_step_timeline.txt 2>&1
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/train_step_kernel_trace_stats.txt 2>&1; rm -rf $O/kt
grep '^{' $O/train_kt.log; cat $O/train_step_timeline.txt
timeout 300 python tools/train_step_bench.py 8 5 2>&1 | grep '^{' | tee $O/train_step.txt
timeout 600 python bench.py --steps 200 --warmup 50 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 > $O/bench_steps200.log 2>&1; grep '^{' $O/bench_steps200.log > $O/bench_256_b16_steps200.json; cut -c1-300 $O/bench_256_b16_steps200.json
ROUND=r06 timeout 1200 bash tools/gpu_profile.sh 256 16 r06i_256_b16 > $O/profile_256.log 2>&1; tail -5 $O/profile_256.log
cat gpurun_out/prof_r06i_256_b16/per_launch_frac.txt
