mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/t_all.log 2>&1; tail -4 gpurun_out/t_all.log
python tools/conv_bench.py 16 bottleneck,up0,up1,hg_dec2,hg_enc1 0,1001,1002,1003 > gpurun_out/convbench_tiles3.log 2>&1; cat gpurun_out/convbench_tiles3.log
for cfg in 2 1; do
echo "== batch 16 N256 cfg$cfg"; EAMM_DMA_CFG_N256=$cfg timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'])"
done
