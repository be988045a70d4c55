#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace of the bench + PMC passes on the
# dominant convolution.  Results land under gpurun_out/; summaries are copied to profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 10 --warmup 3 --cpu-frames 0 > $O/kt_bench.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o sq -- python tools/conv_bench.py 16 bottleneck > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $O/pmc_lds -o lds -- python tools/conv_bench.py 16 bottleneck > $O/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python tools/conv_bench.py 16 bottleneck > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write -- python tools/conv_bench.py 16 bottleneck > $O/pmc_write.log 2>&1
rocprofv3 -L > $O/counters_list.txt 2>&1
find $O -name "*.csv" | head -50
