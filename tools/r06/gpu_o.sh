#!/bin/bash
# round 6, call o: repeatability of the GPU suite on the final tree (three runs in a row), the example script, smoke()
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_o; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_$i.log 2>&1; echo "run $i rc=$?"; tail -1 $O/pytest_$i.log; done
timeout 300 python examples/animate_synthetic.py 64 2>&1 | grep -v amdgpu.ids | tee $O/example.txt
