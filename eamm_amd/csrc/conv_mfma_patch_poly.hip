// Polyphase minimal-filtering form of the UpBlock2d convolution (reference modules/util.py:883-900: nearest x2 -> 3x3 ->
// BN -> ReLU) on the spatial-patch kernel (conv_mfma_patch.hip).
//
// In one dimension the two outputs that share a low-resolution pixel are
//     out[2y]   = w0 e[y-1] + (w1 + w2) e[y]          out[2y+1] = (w0 + w1) e[y] + w2 e[y+1]
// -- four multiplies in the collapsed two-phase form.  With the CENTRE pixel as the common term they need three:
//     m0 = e[y] (w0 + w1 + w2),   m1 = (e[y-1] - e[y]) w0,   m2 = (e[y+1] - e[y]) w2;   out[2y] = m0 + m1,  out[2y+1] = m0 + m2.
// In two dimensions: nine products per low-resolution pixel and input channel give its four output phases (the collapsed
// phase form needs 16, the reference's convolution of the up-sampled map 36):
//     V = T e T^T,  T = [0 1 0; 1 -1 0; 0 -1 1]      (e: the 3x3 low-resolution neighbourhood; differences with the centre)
//     U = G w G^T,  G = [1 1 1; 1 0 0; 0 0 1]        (w: the BatchNorm-folded 3x3 kernel; computed in double on the host)
//     out(py,px) = sum_ij A[py][i] A[px][j] (U_ij . V_ij),  A = [1 1 0; 1 0 1]
// Every coefficient is 0 or +-1 on the input side and 0 or +1 on the output side.  Same multiply count as the Winograd
// F(2x2,2x2) form of conv_mfma_patch_wino.hip (9 per low-resolution pixel), but the nine products are shared by the four
// PHASES of one pixel instead of the four pixels of one phase, which changes the kernel:
//   * a transform point's weights are ONE [64][32] tile for all four outputs (8 KiB instead of 4 x 8 KiB): a barrier
//     interval carries a ROW of three points (24 KiB) = 96 MFMAs per wave -- three barriers per channel chunk, not nine;
//   * (i, j) are compile-time inside an interval, so the fragment of a point is 1 (centre), 2 (edge) or 4 (corner) LDS
//     reads with literal signs, and its patch offsets are immediates;
//   * the four corner points contribute to exactly one output each with coefficient +1: their MFMAs accumulate straight
//     into that output -- no product register, no fold; the other five go through ONE product accumulator M, and the points
//     are ordered M, direct, M, direct, ... so that the fold of a product (adds only) is spread over the MFMA gaps of the
//     direct point behind it, whose output it does not touch (schedule below); 96 of the 384 fold adds per chunk remain
//     exposed.
// fp32 throughout; only the summation order differs from the reference (measured at the prediction: DESIGN.md 5.2d).
#include "conv_common.h"

#include <cstdlib>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int YT = 16;                   // tile side in low-resolution pixels
constexpr int YW = YT + 2;               // patch side (halo 1)
constexpr int YPIX = YW * YW;            // 324 patch pixels
constexpr int YPAD = (YPIX + 7) / 8 * 8; // rounded to whole DMA instructions (8 pixels each)
constexpr int YBN = 64;                  // output channels per workgroup
constexpr int YNT = 2;                   // 32-wide MFMA tiles along N per wave
constexpr int YWAVES = 8;
// processing order of the nine points of a chunk, three per barrier interval (see the schedule in the kernel): (i, j)
constexpr int Y_ORDER[9][2] = {{0, 0}, {1, 1}, {0, 1}, {1, 2}, {0, 2}, {2, 1}, {1, 0}, {2, 2}, {2, 0}};
}  // namespace

struct PatchPolyArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] (hourglass skip concatenation) or null
    int C0, C1;
    unsigned in0_bytes, in1_bytes, w_bytes;
    int B, H, W;           // low-resolution input size; output is [B,2H,2W,Cout]
    int tiles_x, tiles_y, ntiles;
    const float* w;        // packed [ntile][cchunk][point 9 in Y_ORDER][64][32], swizzled
    const float* bias;     // [ntiles*64]
    int Cout, act;
    float* out;
};

__global__ __launch_bounds__(YWAVES * 64) void conv_patch_poly_kernel(const PatchPolyArgs p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = YPAD * BK;            // floats
    constexpr int B_STAGE = 3 * YBN * BK;         // one row of transform points
    constexpr int A_INSTR = (YPAD / 8 + YWAVES - 1) / YWAVES;   // patch DMA instructions per wave per chunk (6)
    constexpr int B_INSTR = 3 * YBN / 8 / YWAVES;               // weight DMA instructions per wave per interval (3)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][A_STAGE] [2][B_STAGE]
    float* const As = smem;
    float* const Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    L /= p.ntiles;
    const int tx0 = (L % p.tiles_x) * YT;
    L /= p.tiles_x;
    const int ty0 = (L % p.tiles_y) * YT;
    const int b = L / p.tiles_y;
    const int cchunks = (p.C0 + p.C1) / BK;

    // ---- patch loader (as conv_patch_phase_kernel): DMA instruction j of this wave covers patch pixels (wave + 8j)*8 .. +8
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto dma_patch_piece = [&](auto jc, int cc, int st) {
        constexpr int j = decltype(jc)::value;
        if ((wave + YWAVES * j) * 8 < YPAD) {
            int lane_l = lane;                                     // opaque: the piece's address arithmetic is redone per use
            asm volatile("" : "+v"(lane_l));                       // (hoisted out of the chunk loop it pins ~20 VGPRs -> scratch)
            const int i = (wave + YWAVES * j) * 8 + (lane_l >> 3);   // patch pixel
            const int pyy = i / YW, pxx = i - pyy * YW;
            const int y = ty0 + pyy - 1, x = tx0 + pxx - 1;
            const bool ok = i < YPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const int slot = ((lane_l & 7) ^ ((i >> 1) & 7)) << 2;
            const int c0 = cc * BK;
            const bool first = c0 < p.C0;
            const int C = first ? p.C0 : p.C1;
            const int coff = first ? c0 : c0 - p.C0;
            const unsigned off = ok ? (unsigned)(((b * p.H + y) * p.W + x) * C + coff + slot) * 4u : OOB;
            float* dst = As + st * A_STAGE + (wave + YWAVES * j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };
    // weights of interval it = cc*3 + i: 24 KiB contiguous; per-lane part fixed, interval part wave-uniform
    const unsigned w_lane = (unsigned)lane * 16u;
    auto dma_weight_piece = [&](auto jc, int it, int st) {
        constexpr int j = decltype(jc)::value;
        const unsigned so = (unsigned)(((ntile * cchunks * 3 + it) * 3 * YBN + (wave * B_INSTR + j) * 8) * BK) * 4u;
        float* dst = Bs + st * B_STAGE + (wave * B_INSTR + j) * (8 * BK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)dst, 16, w_lane, so, 0, 0);
    };

    f32x16 Y[4][YNT];      // the four output phases of each low-resolution pixel, o = 2*py + px
    f32x16 M[YNT];         // product of a point that feeds more than one output
    static_for<YNT>([&](auto jc) {
        static_for<16>([&](auto rc) {
            constexpr int j = decltype(jc)::value, r = decltype(rc)::value;
            M[j][r] = 0.f;
            static_for<4>([&](auto oc) { Y[decltype(oc)::value][j][r] = 0.f; });
        });
    });

    // this lane's pixel inside the tile: wave w owns rows 2w, 2w+1; LDS float address of the 16 bytes of patch pixel
    // idx at K step s: (idx*32 + ((half ^ (idx>>1)&7) << 2)) ^ (8 s)  (the step only flips bits of the swizzled slot)
    const int trow = 2 * wave + (l31 >> 4), tcol = l31 & 15;
    const int idx0 = (trow + 1) * YW + tcol + 1;
    auto addr_of = [&](int idx) { return idx * BK + ((half ^ ((idx >> 1) & 7)) << 2); };
    const int bw0 = l31 * BK + ((half ^ ((l31 >> 1) & 7)) << 2);   // weight row of this lane inside a point's [64][32] tile

    // One transform point (I, J) of the chunk whose patch is in a_stage and whose row of weights is in b_stage (slot S):
    // 4 K steps x YNT x 4 MFMAs into DST (0..3: output phase o, accumulated; 4: the product M, overwritten), with the fold
    // M -> outputs FMASK spread over its MFMA gaps (FMASK = 0: none; only with DST < 4) and DMA pieces of interval DMA - 1
    // at the head (DMA = 0: none).
    // Signs of the fragment terms: row terms {(dy, +1)} for I = 0, {(-1|+1, +1), (0, -1)} else; same for columns.
    auto point = [&](auto ic, auto jc, auto sc, auto dstc, auto fmaskc, auto dmac, const float* a_stage,
                     const float* b_stage, int it_next, int cc_next, int b_st_next, int a_st_next, bool more_w, bool more_a) {
        constexpr int I = decltype(ic)::value, J = decltype(jc)::value, S = decltype(sc)::value;
        constexpr int DST = decltype(dstc)::value, FMASK = decltype(fmaskc)::value;
        constexpr int DMA = decltype(dmac)::value;
        static_assert(FMASK == 0 || (DST < 4 && !((FMASK >> DST) & 1)), "a fold rides only under a direct point whose output it does not touch");
        constexpr int dy = I == 1 ? -YW : (I == 2 ? YW : 0), dx = J == 1 ? -1 : (J == 2 ? 1 : 0);
        constexpr int NR = I == 0 ? 1 : 2, NC = J == 0 ? 1 : 2;     // row / column terms
        int idx0_l = idx0, bw0_l = bw0;                             // opaque: keeps the 2-4 term addresses out of loop-invariant VGPRs
        asm volatile("" : "+v"(idx0_l), "+v"(bw0_l));
        int ta[NR * NC];
        static_for<NR>([&](auto rc_) {
            static_for<NC>([&](auto cc_) {
                constexpr int r = decltype(rc_)::value, c = decltype(cc_)::value;
                constexpr int sh = (r == 0 ? dy : 0) + (c == 0 ? dx : 0);   // term 0 = the displaced pixel, term 1 = the centre
                ta[r * NC + c] = addr_of(idx0_l + sh);
            });
        });
        const float* bt0 = b_stage + S * (YBN * BK);
        // the patch terms of the next step are fetched under the current step's MFMAs (two fragment buffers); the weight
        // rows are fetched between steps into ONE buffer (the partner wave owns the matrix pipe meanwhile): 8 VGPRs fewer
        f32x4 raw[NR * NC], a[2], bb[YNT];
        auto fetch_a = [&](int s) {
#pragma unroll
            for (int k = 0; k < NR * NC; ++k) raw[k] = *reinterpret_cast<const f32x4*>(a_stage + (ta[k] ^ (8 * s)));
        };
        auto fetch_b = [&](int s) {
            const float* bt = bt0 + (bw0_l ^ (8 * s));
#pragma unroll
            for (int j = 0; j < YNT; ++j) bb[j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * BK);
        };
        auto combine = [&](int buf) {
            f32x4 v;
            if constexpr (NR == 1 && NC == 1) v = raw[0];
            else if constexpr (NR * NC == 2) v = raw[0] - raw[1];
            else v = (raw[0] - raw[1]) - (raw[2] - raw[3]);         // (e[dy][dx] - e[dy][0]) - (e[0][dx] - e[0][0])
            a[buf] = v;
        };
        constexpr int NFOLD = (FMASK & 1) + ((FMASK >> 1) & 1) + ((FMASK >> 2) & 1) + ((FMASK >> 3) & 1);
        constexpr int FE = NFOLD * YNT * 16;                        // fold adds of this point
        fetch_a(0);
        fetch_b(0);
        combine(0);
        static_for<4>([&](auto stc) {
            constexpr int step = decltype(stc)::value;
            if constexpr (step + 1 < 4) fetch_a(step + 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4 * YNT>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / YNT, j = q % YNT;
                if constexpr (DST < 4) {
                    Y[DST][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][t], bb[j][t], Y[DST][j], 0, 0, 0);
                } else if constexpr (step == 0 && t == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    M[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][t], bb[j][t], zero, 0, 0, 0);
                } else {
                    M[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][t], bb[j][t], M[j], 0, 0, 0);
                }
                constexpr int g = step * 4 * YNT + q;               // MFMA index within the point (32 total)
                // this gap's share of the fold: elements [g*FE/32, (g+1)*FE/32) of the (output, n-tile, register) lattice
                if constexpr (FE > 0) {
                    constexpr int e0 = g * FE / 32, e1 = (g + 1) * FE / 32;
                    static_for<e1 - e0>([&](auto ec) {
                        constexpr int e = e0 + decltype(ec)::value;
                        constexpr int k = e / (YNT * 16), jj = (e / 16) % YNT, r = e % 16;
                        // k-th set bit of FMASK
                        constexpr int o = [] { int n = 0; for (int b2 = 0; b2 < 4; ++b2) if ((FMASK >> b2) & 1) { if (n == k) return b2; ++n; } return 0; }();
                        Y[o][jj][r] += M[jj][r];
                    });
                }
                if constexpr (DMA != 0 && g % 4 == 3 && g / 4 < B_INSTR) {
                    if (more_w) dma_weight_piece(std::integral_constant<int, g / 4>{}, it_next, b_st_next);
                } else if constexpr (DMA != 0 && g % 4 == 3 && g / 4 >= B_INSTR && g / 4 < B_INSTR + 2) {
                    constexpr int jp = 2 * (DMA - 1) + (g / 4 - B_INSTR);   // patch pieces 2k, 2k+1 of this wave ride in interval k
                    if constexpr (jp < A_INSTR) {
                        if (more_a) dma_patch_piece(std::integral_constant<int, jp>{}, cc_next, a_st_next);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (step + 1 < 4) {
                fetch_b(step + 1);
                combine((step + 1) & 1);
            }
        });
    };

    // the part of the product's fold that could not ride under a direct point: M -> outputs of MASK, before M is overwritten
    auto fold_now = [&](auto maskc) {
        constexpr int MASK = decltype(maskc)::value;
        static_for<4>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            if constexpr ((MASK >> o) & 1) {
                static_for<YNT>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<16>([&](auto rc) { constexpr int r = decltype(rc)::value; Y[o][j][r] += M[j][r]; });
                });
            }
        });
    };

    // Schedule of one chunk (o = 2*py + px; "| fold {..}" = M into those outputs, in the MFMA gaps of the point on its line;
    // the running MFMAs never accumulate into an output a fold touches):
    //   interval 0:  (0,0) -> M
    //                (1,1) -> Y0   | fold {1,2,3}            then fold {0} (exposed)
    //                (0,1) -> M
    //   interval 1:  (1,2) -> Y1   | fold {0,2}
    //                (0,2) -> M
    //                (2,1) -> Y2   | fold {1,3}
    //   interval 2:  (1,0) -> M
    //                (2,2) -> Y3   | fold {0,1}
    //                (2,0) -> M                              then fold {2,3} (exposed)
    using IC0 = std::integral_constant<int, 0>;
    using IC1 = std::integral_constant<int, 1>;
    using IC2 = std::integral_constant<int, 2>;
    using IC3 = std::integral_constant<int, 3>;
    using ICM = std::integral_constant<int, 4>;
    auto chunk = [&](int cc, int& b_st) {
        const int a_st = cc & 1;
        const float* a_stage = As + a_st * A_STAGE;
        const bool more_a = cc + 1 < cchunks;
        auto interval = [&](int k, auto body) {
            const int it = cc * 3 + k;
            const bool more_w = it + 1 < cchunks * 3;
            body(Bs + b_st * B_STAGE, it + 1, b_st ^ 1, more_w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            b_st ^= 1;
        };
        interval(0, [&](const float* bs, int itn, int bsn, bool mw) {
            point(IC0{}, IC0{}, IC0{}, ICM{}, IC0{}, IC1{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            point(IC1{}, IC1{}, IC1{}, IC0{}, std::integral_constant<int, 0xE>{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            fold_now(std::integral_constant<int, 0x1>{});
            point(IC0{}, IC1{}, IC2{}, ICM{}, IC0{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
        });
        interval(1, [&](const float* bs, int itn, int bsn, bool mw) {
            point(IC1{}, IC2{}, IC0{}, IC1{}, std::integral_constant<int, 0x5>{}, IC2{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            point(IC0{}, IC2{}, IC1{}, ICM{}, IC0{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            point(IC2{}, IC1{}, IC2{}, IC2{}, std::integral_constant<int, 0xA>{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
        });
        interval(2, [&](const float* bs, int itn, int bsn, bool mw) {
            point(IC1{}, IC0{}, IC0{}, ICM{}, IC0{}, IC3{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            point(IC2{}, IC2{}, IC1{}, IC3{}, std::integral_constant<int, 0x3>{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            point(IC2{}, IC0{}, IC2{}, ICM{}, IC0{}, IC0{}, a_stage, bs, itn, cc + 1, bsn, a_st ^ 1, mw, more_a);
            fold_now(std::integral_constant<int, 0xC>{});
        });
    };

    // ---- main loop over channel chunks
    static_for<A_INSTR>([&](auto jc) { dma_patch_piece(jc, 0, 0); });
    static_for<B_INSTR>([&](auto jc) { dma_weight_piece(jc, 0, 0); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int b_st = 0;
    for (int cc = 0; cc < cchunks; ++cc) chunk(cc, b_st);

    // ---- epilogue: per phase, stage the 256 x 64 tile through LDS and store 16-byte pieces (as conv_patch_phase_kernel)
    constexpr int LDO = YBN + 4, C4 = YBN / 4, NTHR = YWAVES * 64, PER = YT * YT * C4 / NTHR;
    const int OH = 2 * p.H, OW = 2 * p.W;
    static_for<4>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        __syncthreads();
        static_for<YNT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int col = j * 32 + l31;
            const float bias = p.bias[ntile * YBN + col];
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // tile pixel: (row >> 4, row & 15)
                smem[row * LDO + col] = Y[ph][j][r] + bias;
            });
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / C4, c4 = idx - row * C4;
            const int y = ty0 + (row >> 4), x = tx0 + (row & 15);
            const int n = ntile * YBN + c4 * 4;
            if (y < p.H && x < p.W && n < p.Cout) {
                float4 v = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                const size_t o = ((size_t)(b * OH + 2 * y + (ph >> 1)) * OW + 2 * x + (ph & 1)) * p.Cout + n;
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t patch_poly_packed_elems(int Cin_packed, int Cout) {
    return (size_t)((Cout + YBN - 1) / YBN) * (Cin_packed / CONV_BK) * 9 * YBN * CONV_BK;
}

// w: 3x3 weights [Cout][Cin][3][3] (BatchNorm folded) -> U = G w G^T, G = [1 1 1; 1 0 0; 0 0 1], laid out
// [ntile][cchunk][point][64][32] (points in processing order, Y_ORDER), LDS swizzle applied.
void patch_poly_pack_host(const float* w, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst) {
    static const double G[3][3] = {{1, 1, 1}, {1, 0, 0}, {0, 0, 1}};
    const int BK = CONV_BK, cch = cin_packed / BK;
    const size_t total = patch_poly_packed_elems(cin_packed, Cout);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / YBN, nl = o % YBN;
        for (int cp = 0; cp < cin_packed; ++cp) {
            const int c = cin_map ? cin_map[cp] : cp;
            if (c < 0 || c >= Cin) continue;
            const float* s = w + ((size_t)o * Cin + c) * 9;
            const int cc = cp / BK, kl = cp % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int pt = 0; pt < 9; ++pt) {
                const int i = Y_ORDER[pt][0], j = Y_ORDER[pt][1];
                double u = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) u += G[i][a] * G[j][b2] * (double)s[a * 3 + b2];
                const size_t tile = ((size_t)nt * cch + cc) * 9 + pt;
                dst[(tile * YBN + nl) * BK + kk] = (float)u;
            }
        }
    }
}

hipError_t patch_poly_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                             float* out, hipStream_t stream) {
    if ((L.C0 % CONV_BK) || (L.C1 % CONV_BK) || (L.Cout & 3) || L.w_poly == nullptr) return hipErrorInvalidValue;
    PatchPolyArgs a{};
    a.in0 = in0;
    a.in1 = L.C1 ? in1 : nullptr;
    a.C0 = L.C0;
    a.C1 = L.C1;
    const size_t px = (size_t)B * H * W;
    const size_t b0 = px * L.C0 * 4, b1 = px * L.C1 * 4, bw = patch_poly_packed_elems(L.C0 + L.C1, L.Cout) * 4;
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.in0_bytes = (unsigned)b0;
    a.in1_bytes = (unsigned)b1;
    a.w_bytes = (unsigned)bw;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = (W + YT - 1) / YT;
    a.tiles_y = (H + YT - 1) / YT;
    a.ntiles = (L.Cout + YBN - 1) / YBN;
    a.w = L.w_poly;
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.act = act;
    a.out = out;
    constexpr size_t lds_loop = sizeof(float) * 2 * (YPAD * CONV_BK + 3 * YBN * CONV_BK);
    constexpr size_t lds_epi = sizeof(float) * (YT * YT) * (YBN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static unsigned long long configured = 0;
    if (hipError_t e = ensure_dynamic_lds(conv_patch_poly_kernel, lds, &configured); e != hipSuccess) return e;
    const int blocks = a.tiles_x * a.tiles_y * B * a.ntiles;
    hipLaunchKernelGGL(conv_patch_poly_kernel, dim3(blocks), dim3(YWAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace eamm
