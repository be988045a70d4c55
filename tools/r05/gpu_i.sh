#!/bin/bash
# round 5, call i: smoke + whole GPU suite on the final tree
mkdir -p gpurun_out/r05_i
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_i/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05_i/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_i/pytest.log
