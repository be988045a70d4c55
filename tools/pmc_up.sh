cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_up
rm -rf $O; mkdir -p $O; cd $R
for t in 3000 3001; do
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p$t -o p$t -- python tools/conv_bench.py 16 up0 $t > $O/p$t.log 2>&1
python tools/rocpd_summary.py $O/p$t/p${t}_results.db | grep -E "patch" | head -12
done
