#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace of bench.py + PMC passes on the dominant
# convolution (bottleneck 3x3 256->256 @64x64, 16 frames, LDS-DMA 256x256 tile).  Text summaries land in
# gpurun_out/prof/*.txt; copy the ones to be judged into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
TILE=${1:-1001}
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 10 --warmup 3 --cpu-frames 0 > $O/kt_bench.log 2>&1
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/kernel_trace_stats.txt 2>&1
pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name -- python tools/conv_bench.py 16 bottleneck $TILE > $O/$name.log 2>&1; python tools/rocpd_summary.py $O/$name/${name}_results.db | grep -E "PMC|conv_mfma|wino" > $O/pmc_$name.txt; }
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cat $O/kernel_trace_stats.txt | head -30; cat $O/pmc_*.txt
