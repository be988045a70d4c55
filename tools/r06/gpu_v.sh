#!/bin/bash
# round 6, call v: deep hourglass levels on the skinny tiles by split-K factor (more workgroups per CU to hide the LDS-DMA latency of one-wave-per-SIMD tiles?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_v; mkdir -p $O; cd $R
for B in 8 16; do
for sk in 0 16 24 32 48; do
  echo "== B=$B splitk=$sk"
  CONV_BENCH_SPLITK=$sk python tools/conv_bench.py $B hg_enc4 1005 2>/dev/null | grep -v sum
  T0=1005; [ $B -le 8 ] && T0=1004
  CONV_BENCH_SPLITK=$sk python tools/conv_bench.py $B hg_dec0 $T0 2>/dev/null | grep -v sum
  CONV_BENCH_SPLITK=$sk python tools/conv_bench.py $B hg_dec1 1005 2>/dev/null | grep -v sum
done; done 2>&1 | tee $O/skinny_splitk2.txt
