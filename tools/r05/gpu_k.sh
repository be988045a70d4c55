#!/bin/bash
mkdir -p gpurun_out/r05_k
cd $GRAFT_REPO_ROOT
python tools/r05/dbg_euro.py 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s > gpurun_out/r05_k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_k/pytest.log; grep -n "one-euro" gpurun_out/r05_k/pytest.log
