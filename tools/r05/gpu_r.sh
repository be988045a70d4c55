#!/bin/bash
# round 5, call r: profiles of the final kernels (the input transform's workgroup order changed), then the committed bench lines and the suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_r; mkdir -p $O
ROUND=r05 timeout 900 bash tools/gpu_profile.sh 256 16 r05r_256_b16 > $O/profile_256.log 2>&1; tail -4 $O/profile_256.log
ROUND=r05 timeout 900 bash tools/gpu_profile.sh 512 8 r05r_512_b8 > $O/profile_512.log 2>&1; tail -4 $O/profile_512.log
cp gpurun_out/prof_r05r_256_b16/pmc_traffic.json $O/pmc_traffic_256.json; cp gpurun_out/prof_r05r_512_b8/pmc_traffic.json $O/pmc_traffic_512.json
