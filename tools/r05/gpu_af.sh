#!/bin/bash
# round 5, call af: the final tree once more on a fresh box -- smoke, whole GPU suite, the default line (nothing installed from it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_af; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 400 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json
python - <<'P'
import json
d=json.load(open("gpurun_out/r05_af/bench_256_b16.json")); e=d.get("e2e_clip") or {}
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "clip", d["clip"]["frames_per_s"], "e2e", e.get("frames_per_s"), "cpu", d["cpu_baseline"]["value"])
P
