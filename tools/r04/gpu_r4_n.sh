#!/bin/bash
# round 4: sweep of the hourglass / decoder tile knobs in the two-chain regime (bench.py 256x256 x 16)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_n; mkdir -p $O
timeout 900 bash tools/exp_env.sh "EAMM_PATCH_SPLIT_MAX=4" "EAMM_PATCH_SPLIT_MAX=2" "EAMM_PATCH_SPLIT_MAX=8" "EAMM_PATCH_SPLIT_MAX=1" \
  "EAMM_SKINNY_MAX_M=0" "EAMM_SKINNY_MAX_M=65536" "EAMM_DMA_MIN_M=1024" "EAMM_DMA_MIN_M=16384" "EAMM_HEAD_COL7_MIN_TILES=1000000" \
  "EAMM_WINO4_VARIANT=0" "EAMM_ENC_WINO_MIN_TILES=100000" "EAMM_PATCH_SPLIT_MAX=4" 2>&1 | tee $O/exp_knobs.txt
