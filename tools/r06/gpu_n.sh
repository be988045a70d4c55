#!/bin/bash
# round 6, call n: the two bench lines re-run so that they carry the PMC record (roofline.traffic) of the kernel source they ran;
# per-launch roofline derivation with launches grouped by grid (512x512: the encoder levels share the bottleneck's GEMM instantiation)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_n; mkdir -p $O; cd $R
timeout 600 python bench.py > $O/bench_256_b16.log 2>&1; grep '^{' $O/bench_256_b16.log > $O/bench_256_b16.json; cut -c1-200 $O/bench_256_b16.json
timeout 600 python bench.py --size 512 --cpu-frames 0 --clip-frames 512 > $O/bench_512_b8.log 2>&1; grep '^{' $O/bench_512_b8.log > $O/bench_512_b8.json; cut -c1-200 $O/bench_512_b8.json
cd /tmp && export TMPDIR=/tmp
for cfg in "256 16" "512 8"; do set -- $cfg
BENCH="python $R/bench.py --size $1 --batch $2 --steps 10 --warmup 3 --cpu-frames 0 --clip-frames 0 --train-pairs 0 --e2e-frames 0 --latency-frames 0 --no-all-outputs"
cd $R; rocprofv3 --kernel-trace --stats -d $O/kt$1 -o kt -- $BENCH --graph > $O/kt_$1.log 2>&1
grep '^{' $O/kt_$1.log > $O/under_$1.json
python tools/rocpd_summary.py --per-launch-frac $O/under_$1.json $O/kt$1/kt_results.db $O/bench_$( [ $1 = 256 ] && echo 256_b16 || echo 512_b8 ).json > $O/per_launch_frac_$1.txt 2>&1
sqlite3 $O/kt$1/kt_results.db "pragma table_info('kernels')" 2>/dev/null | head -40 > $O/kernels_columns_$1.txt || python -c "
import sqlite3;c=sqlite3.connect('$O/kt$1/kt_results.db');print([d[0] for d in c.execute('select * from kernels limit 1').description])" > $O/kernels_columns_$1.txt
rm -rf $O/kt$1; cat $O/per_launch_frac_$1.txt
done
cat $O/kernels_columns_256.txt | head -5
