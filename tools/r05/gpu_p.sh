#!/bin/bash
# round 5, call p: the whole GPU suite twice more on the final tree (repeatability)
mkdir -p gpurun_out/r05_p
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_p/run$i.log 2>&1; tail -1 gpurun_out/r05_p/run$i.log
  grep -n "^FAILED\|^E  " gpurun_out/r05_p/run$i.log | head -6 | cut -c1-300
done
