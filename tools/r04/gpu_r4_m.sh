#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_m; mkdir -p $O
timeout 500 bash tools/exp_env.sh "EAMM_ENC_CUS_PCT=100" "EAMM_ENC_CUS_PCT=200" "EAMM_ENC_CUS_PCT=100" "EAMM_ENC_CUS_PCT=200" "EAMM_ENC_CUS_PCT=400" 2>&1 | tee $O/exp_enc_cus2.txt
