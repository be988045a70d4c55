#!/bin/bash
# round 5, call c: whole GPU suite (new: pipeline, adversarial, plan64, bench legs) + the default bench line
mkdir -p gpurun_out/r05_c
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -s > gpurun_out/r05_c/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r05_c/pytest.log
tail -5 gpurun_out/r05_c/pytest.log
grep -n "x floor\|whole chain\|one-euro\|value .* jacobian" gpurun_out/r05_c/pytest.log | head -40
timeout 400 python bench.py > gpurun_out/r05_c/bench.json 2> gpurun_out/r05_c/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r05_c/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r05_c/bench.json"))
print(d["value"], d["clip"]["frames_per_s"], d["e2e_clip"])
P
