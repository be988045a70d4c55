#!/bin/bash
# round 5, call ae: FETCH_SIZE of the bottleneck GEMM by launch size (8-frame launches of the contract line, 32-frame launches of the clip leg) with the
# V stream loaded normally (variant 3) and non-temporally (variant 6) -- evidence for profiles/r05_experiments.txt 14
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ae; mkdir -p $O; cd $R
BENCH="python bench.py --steps 4 --warmup 1 --cpu-frames 0 --train-pairs 0 --e2e-frames 0"
for v in 3 6; do
  EAMM_WINO4_VARIANT=$v timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$v -o f$v -- $BENCH > $O/f$v.log 2>&1
  python tools/pmc_by_grid.py $O/f$v/f${v}_results.db FETCH_SIZE wino4_gemm_kernel > $O/fetch_by_launch_size_variant$v.txt 2>&1
  cat $O/fetch_by_launch_size_variant$v.txt
  rm -rf $O/f$v
done
