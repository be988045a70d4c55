#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace of bench.py + PMC passes over bench.py itself at one
# (size, batch).  Text summaries land in gpurun_out/prof_<tag>/*.txt; copy the ones to be judged into profiles/.
#   tools/gpu_profile.sh [size] [batch] [tag]
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r06}
SIZE=${1:-256}; BATCH=${2:-16}; TAG=${3:-${SIZE}_b${BATCH}}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
BENCH="python bench.py --size $SIZE --batch $BATCH --steps 10 --warmup 3 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 --latency-frames 0 --no-all-outputs"   # the contract line's kernels only
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BENCH --graph > $O/kt_bench.log 2>&1
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/kernel_trace_stats.txt 2>&1
grep '^{' $O/kt_bench.log > $O/bench_under_kernel_trace.json
# the bottleneck stage's union window re-derived from the trace's kernel timestamps (vs bench.py's HIP events, same run)
$BENCH > $O/bench_unprofiled.log 2>&1; grep '^{' $O/bench_unprofiled.log > $O/bench_unprofiled.json   # the same command without the profiler
python tools/rocpd_summary.py --bneck-timeline $O/bench_under_kernel_trace.json $O/kt/kt_results.db $O/bench_unprofiled.json > $O/bneck_timeline.txt 2>&1
# roofline.frac re-derived from the trace's per-kernel averages alone (the chains drift apart under the profiler; durations do not)
python tools/rocpd_summary.py --per-launch-frac $O/bench_under_kernel_trace.json $O/kt/kt_results.db $O/bench_unprofiled.json > $O/per_launch_frac.txt 2>&1
pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name -- $BENCH > $O/$name.log 2>&1; python tools/rocpd_summary.py $O/$name/${name}_results.db | grep -v rocclr > $O/pmc_$name.txt; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU
CHAINS=$(python -c "import json;print(json.load(open('$O/bench_under_kernel_trace.json'))['roofline'].get('chains',1))" 2>/dev/null || echo 1)
python tools/pmc_traffic.py $O/fetch/fetch_results.db $O/write/write_results.db $SIZE $BATCH "profiles/${ROUND}_${SIZE}_b${BATCH}_pmc_{fetch,write}.txt" $O/pmc_traffic.json $CHAINS > $O/pmc_traffic.log 2>&1
# only gpurun_out/ travels back and it is capped at 64 MiB: keep the text summaries, drop the raw rocprofv3 databases
rm -rf $O/kt $O/fetch $O/write $O/sq $O/lds
head -30 $O/kernel_trace_stats.txt; cat $O/pmc_traffic.log
