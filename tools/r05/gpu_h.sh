#!/bin/bash
# round 5, call h: kernel trace of the end-to-end leg (make_animation_smooth whole: detectors, smoothing, generator, D2H)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --clip-frames 0 --train-pairs 0 > $O/bench.log 2>&1
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/e2e_kernel_trace_stats.txt 2>&1
grep '^{' $O/bench.log > $O/bench_under_trace.json
rm -rf $O/kt
head -40 $O/e2e_kernel_trace_stats.txt | cut -c1-160
python -c "
import json; d=json.load(open('$O/bench_under_trace.json')); print(d['e2e_clip']['frames_per_s'], d['e2e_clip']['phases_ms_rank0'])"
