#!/bin/bash
# PMC passes on one layer of tools/conv_bench.py:  tools/pmc_conv.sh <layer> <tile> [batch]   -> gpurun_out/pmc_conv_<layer>_<tile>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; L=$1; T=$2; B=${3:-16}
O=$R/gpurun_out/pmc_conv_${L}_${T}; rm -rf $O; mkdir -p $O; cd $R
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n -- python tools/conv_bench.py $B $L $T > $O/$n.log 2>&1; python tools/rocpd_summary.py $O/$n/${n}_results.db | grep -v rocclr | grep -E "PMC|kernel|patch|wino|conv" ; rm -rf $O/$n; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA
