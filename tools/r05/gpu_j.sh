#!/bin/bash
# round 5, call j: streamed front end -- bit-equality tests, then the end-to-end leg streamed vs phased
mkdir -p gpurun_out/r05_j
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s > gpurun_out/r05_j/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05_j/pytest.log
timeout 400 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --clip-frames 0 --train-pairs 0 > gpurun_out/r05_j/bench.json 2> gpurun_out/r05_j/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r05_j/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r05_j/bench.json")); e=d["e2e_clip"]
print("e2e streamed", e["frames_per_s"], e["seconds"], e["verify"], e["phases_ms_rank0"])
P
