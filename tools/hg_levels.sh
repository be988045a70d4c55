#!/bin/bash
# Per-level timing of the dense-motion hourglass at 8 and 16 frames in the forms the pipeline launches (tools/conv_bench.py):
# encoder levels 0-3 in F(4x4) split form (2152/2153/2156 = 2/3/6 workgroups per block; un-pooled shapes), level 4 direct;
# decoder levels 0-2 on the im2col phase kernels (auto tile), levels 3-4 on the polyphase patch kernel (3003, split via CONV_BENCH_SPLITK).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 8 16; do
  echo "== $B frames"
  python tools/conv_bench.py $B hg_enc0_nopool,hg_enc1_nopool,hg_enc2_nopool,hg_enc3_nopool 2152,2153,2156 2>/dev/null | grep -v sum
  python tools/conv_bench.py $B hg_enc0,hg_enc1,hg_enc2,hg_enc3,hg_enc4,hg_dec0,hg_dec1,hg_dec2,hg_dec3,hg_dec4 0 2>/dev/null | grep -v sum
  for sk in 1 2 4; do CONV_BENCH_SPLITK=$sk python tools/conv_bench.py $B hg_dec3,hg_dec4 3003 2>/dev/null | grep -v sum | sed "s/$/  splits=$sk/"; done
done
