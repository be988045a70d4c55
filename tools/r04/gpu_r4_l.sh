#!/bin/bash
# re-run the two bench lines after the PMC passes so that they carry the traffic record of the kernel source they ran
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r04_z; mkdir -p $O
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; cut -c1-200 $O/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > $O/bench_512_b8.log 2>&1; grep '^{' $O/bench_512_b8.log > $O/bench_512_b8.json; cut -c1-200 $O/bench_512_b8.json
