mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -k "wino" 2>&1 | tail -15
python tools/conv_bench.py 16 bottleneck 1001,2000 2>&1 | tail -4
python -m pytest tests/test_gpu_generator.py -m gpu -q --tb=short -s 2>&1 | tail -25
for b in 16 32; do echo "== batch $b (winograd bottleneck)"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'])"; done
echo "== batch 16 (direct)"; EAMM_WINO_MIN_M=-1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'])"
