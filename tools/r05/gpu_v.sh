#!/bin/bash
# round 5, call v: final kernels (XCD-aware order + non-temporal V stores in the input transform): 256x256 profile, suite, committed lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_v; mkdir -p $O
ROUND=r05 timeout 900 bash tools/gpu_profile.sh 256 16 r05v_256_b16 > $O/profile_256.log 2>&1; tail -3 $O/profile_256.log
ROUND=r05 timeout 900 bash tools/gpu_profile.sh 512 8 r05v_512_b8 > $O/profile_512.log 2>&1; tail -3 $O/profile_512.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
