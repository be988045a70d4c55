#!/bin/bash
# Per-level timing of the dense-motion hourglass at 8 and 16 frames in the forms the pipeline launches (tools/conv_bench.py):
# encoder levels 0-3 in F(4x4) split form (2152/2153/2156 = 2/3/6 workgroups per block; un-pooled shapes), level 4 direct;
# decoder levels 0-2 on the im2col phase kernels (the pipeline's skinny tiles), levels 3-4 on the polyphase patch kernel (3003, split via CONV_BENCH_SPLITK).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 8 16; do
  echo "== $B frames"
  python tools/conv_bench.py $B hg_enc0_nopool,hg_enc1_nopool,hg_enc2_nopool,hg_enc3_nopool 2152,2153,2156 2>/dev/null | grep -v sum
  # the deep levels on the tiles the pipeline picks for them (skinny LDS-DMA tiles: 1004 = 32 x 128 for per-phase M <= 32,
  # 1005 = 64 x 128 above; tile 0 of conv_bench.py is the register-staged 128 x N kernel, which the pipeline does not use here)
  T4=1005; [ $B -le 2 ] && T4=1004
  python tools/conv_bench.py $B hg_enc4 $T4 2>/dev/null | grep -v sum
  T0=1005; [ $B -le 8 ] && T0=1004
  python tools/conv_bench.py $B hg_dec0 $T0 2>/dev/null | grep -v sum
  T1=1005; [ $B -le 2 ] && T1=1004
  python tools/conv_bench.py $B hg_dec1 $T1 2>/dev/null | grep -v sum
  python tools/conv_bench.py $B hg_dec2 1005 2>/dev/null | grep -v sum
  for sk in 1 2 4; do CONV_BENCH_SPLITK=$sk python tools/conv_bench.py $B hg_dec3,hg_dec4 3003 2>/dev/null | grep -v sum | sed "s/$/  splits=$sk/"; done
done
