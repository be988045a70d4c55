mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/t_all.log 2>&1; tail -4 gpurun_out/t_all.log
python tools/conv_bench.py 16 bottleneck,up0,up1 0,1001,1002,1003 > gpurun_out/convbench_tiles2.log 2>&1; cat gpurun_out/convbench_tiles2.log
for b in 16 4 1; do
  for m in 1024 -1; do
    echo "== batch $b EAMM_DMA_MIN_M=$m"
    EAMM_DMA_MIN_M=$m timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  done
done
echo "== batch 16 N256 cfg1"; EAMM_DMA_CFG_N256=1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
echo "== batch 16 N256 cfg3"; EAMM_DMA_CFG_N256=3 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2; mkdir -p $O; cd $R
for t in 1001 1002; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/sq_$t -o sq -- python tools/conv_bench.py 16 bottleneck $t > $O/sq_$t.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT -d $O/lds_$t -o lds -- python tools/conv_bench.py 16 bottleneck $t > $O/lds_$t.log 2>&1
done
ls $O
