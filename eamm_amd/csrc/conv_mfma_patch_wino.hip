// Winograd F(2x2, 2x2) form of the spatial-patch UpBlock2d kernel (conv_mfma_patch.hip; reference modules/util.py:883-900:
// nearest x2 -> 3x3 -> BN -> ReLU).
//
// The collapsed up-convolution is four independent 2x2 correlations over the LOW-resolution input, one per output
// phase (py,px): out[2y+py][2x+px] = sum_{t,u in {0,1}} w_ph[t][u] * d[y-1+py+t][x-1+px+u].  A 2-tap filter admits the
// minimal algorithm F(2,2) -- two outputs from three multiplies instead of four:
//     m1 = (e0 - e1) w0,  m2 = e1 (w0 + w1),  m3 = (e1 - e2) w1;   out0 = m1 + m2,  out1 = m2 - m3
// (B^T = [1 -1 0; 0 1 0; 0 1 -1], G = [1 0; 1 1; 0 1], A^T = [1 1 0; 0 1 -1]: every coefficient is 0 or +-1), so a 2x2
// tile of same-phase outputs costs 9 multiplies per input channel instead of 16 -- 0.5625x the MFMA passes of the
// phase form, 4x fewer than the reference's direct convolution of the up-sampled map.  Plain fp32 arithmetic; only the
// summation order differs (measured at the prediction: DESIGN.md section 5.2c).
//
// Same workgroup geometry as conv_patch_phase_kernel: a 16x16 tile of low-resolution pixels x 64 output channels, the
// 18x18 input patch of a 32-channel chunk DMA'd into LDS once.  What changes:
//   * the GEMM row is a (phase, 2x2 tile) pair: 4 phases x 64 tiles = 256 rows; wave w owns phase w/2 and tiles
//     32(w%2) .. +32, so the four phases' weights of one transform point -- [4][64][32] = 32 KiB, the size of a phase
//     interval before -- form one barrier interval: 9 intervals per chunk instead of 4 x 4 taps;
//   * the transformed input is never materialised: the A fragment of transform point (i,j) is the +-1 combination of
//     1, 2 or 4 patch pixels (B^T e B), built from LDS reads when the fragment is fetched;
//   * after each interval the partial product M_ij of this chunk is folded into the tile's four output accumulators
//     Y[oy][ox] += A^T[oy][i] A^T[ox][j] M_ij (the fold is linear, so folding chunk by chunk is exact).
// U = G w_ph G^T is computed in double on the host from the BatchNorm-folded, phase-collapsed weights.
#include "conv_common.h"

#include <cstdlib>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int QT = 16;                   // tile side in low-resolution pixels
constexpr int QW = QT + 2;               // patch side (halo 1)
constexpr int QPIX = QW * QW;            // 324 patch pixels
constexpr int QPAD = (QPIX + 7) / 8 * 8; // rounded to whole DMA instructions (8 pixels each)
constexpr int QBN = 64;                  // output channels per workgroup
constexpr int QNT = 2;                   // 32-wide MFMA tiles along N per wave
constexpr int QWAVES = 8;
constexpr int QHALF = QW * (QW / 2);      // patch pixels per column parity (162)

// 16-byte-slot swizzle of patch pixel (row r, half-column ch): the lanes of a wave differ in (Ty, Tx) with r = 2 Ty + m,
// ch = Tx + k; 2 (Ty mod 4) + bit 0 of (Tx + k)/2 is distinct over each group of 16 lanes a ds_read_b128 serves at once,
// and the two lanes of a pair sit on different bank halves (position parity) -- conflict-free fragment reads.
__device__ __forceinline__ int patch_swz(int r, int ch) { return ((r & 6) | ((ch >> 1) & 1)) & 7; }
}  // namespace

struct PatchWinoArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] (hourglass skip concatenation) or null
    int C0, C1;
    unsigned in0_bytes, in1_bytes, w_bytes;
    int B, H, W;           // low-resolution input size (even); output is [B,2H,2W,Cout]
    int tiles_x, tiles_y, ntiles;
    const float* w;        // packed [ntile][cchunk][xi 9][phase 4][64][32], swizzled
    const float* bias;     // [ntiles*64]
    int Cout, act;
    int young_first;
    long long* trace;      // diagnostic (tools/pwino_trace.py): per-interval cycle stamps of workgroup 0, or null
    float* out;
};

__global__ __launch_bounds__(QWAVES * 64) void conv_patch_wino_kernel(const PatchWinoArgs p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = QPAD * BK;            // floats
    constexpr int B_STAGE = 4 * QBN * BK;         // one transform point: 4 phases
    constexpr int A_INSTR = (QPAD / 8 + QWAVES - 1) / QWAVES;   // patch DMA instructions per wave per chunk (6)
    constexpr int B_INSTR = 4 * QBN / 8 / QWAVES;               // weight DMA instructions per wave per interval (4)
    static_assert(A_INSTR <= 9, "one patch piece per interval");
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
    const int tx0 = (L % p.tiles_x) * QT;
    L /= p.tiles_x;
    const int ty0 = (L % p.tiles_y) * QT;
    const int b = L / p.tiles_y;
    const int cchunks = (p.C0 + p.C1) / BK;

    // ---- patch loader: DMA instruction j of this wave covers patch pixels (wave + 8j)*8 .. +8
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // piece j (run time, wave-uniform) of this wave: patch pixels (wave + 8j)*8 .. +8; addressing is recomputed per piece
    auto dma_patch_piece = [&](int j, int cc, int st) {
        const int piece = wave + QWAVES * j;
        if (j < A_INSTR && piece * 8 < QPAD) {
            // LDS position q of the patch image: even patch columns first, then odd ones (QHALF positions each), rows of
            // QW/2 -- a wave reads every second column, so this keeps its 32 lanes on both halves of the banks; the
            // 16-byte slot is XOR-swizzled with g(row, half-column) (patch_swz) instead of the position bits
            const int q = piece * 8 + (lane >> 3);
            const int hp = q >= QHALF ? 1 : 0, qq = q - hp * QHALF;
            const int pr = qq / (QW / 2), ch = qq - pr * (QW / 2), pc = 2 * ch + hp;
            const int y = ty0 + pr - 1, x = tx0 + pc - 1;
            const bool ok = (q < QPIX) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
            const int pix = (b * p.H + y) * p.W + x;
            const int slot = ((lane & 7) ^ patch_swz(pr, ch)) << 2;
            const int c0 = cc * BK;
            const bool first = c0 < p.C0;
            const int C = first ? p.C0 : p.C1;
            const int coff = first ? c0 : c0 - p.C0;
            const unsigned off = ok ? (unsigned)(pix * C + coff + slot) * 4u : OOB;
            float* dst = As + st * A_STAGE + piece * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };
    // weights of interval it = cc*9 + xi: 32 KiB contiguous ([phase][64][32])
    auto dma_weight_piece = [&](auto jc, int it, int st) {
        constexpr int j = decltype(jc)::value;
        const unsigned off = (unsigned)(((ntile * cchunks * 9 + it) * 4 * QBN + (wave * B_INSTR + j) * 8) * BK + lane * 4) * 4u;
        float* dst = Bs + st * B_STAGE + (wave * B_INSTR + j) * (8 * BK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)dst, 16, off, 0, 0, 0);
    };

    f32x16 acc[QNT];          // M_ij of the interval in flight (this chunk's share)
    f32x16 Y[4][QNT];         // the four outputs of each 2x2 tile, o = 2*oy + ox
    static_for<QNT>([&](auto jc) {
        static_for<16>([&](auto rc) {
            constexpr int j = decltype(jc)::value, r = decltype(rc)::value;
            acc[j][r] = 0.f;
            static_for<4>([&](auto oc) { Y[decltype(oc)::value][j][r] = 0.f; });
        });
    });

    // GEMM row of this lane: phase = wave / 2, tile t = 32 (wave % 2) + l31 of the 8x8 tiles, (Ty, Tx) = (t / 8, t % 8)
    const int yf = p.young_first;
    const int ph = wave >> 1, py = ph >> 1, px = ph & 1;
    const int tl = (wave & 1) * 32 + l31;

    // One interval = one transform point of one channel chunk: 4 K steps x QNT tiles x 4 MFMAs.  The interval body
    // exists ONCE (transform point at run time, wave-uniform): nine unrolled copies next to the 128 + 32 accumulator
    // registers are more than the register allocator places without spilling.
    //   (B^T e B)[ti][tj] = sum over rows {ra, rb} x columns {ca, cb} of +-e:  ti = 0: e0 - e1, 1: e1, 2: e1 - e2.
    auto compute = [&](int xi, int a_st, int b_st, int it_next, int cc_next, bool more_w, bool more_a) {
        const int ti = xi / 3, tj = xi - 3 * ti;
        const int ra = ti == 0 ? 0 : 1, rb = ti == 2 ? 2 : 1, ca = tj == 0 ? 0 : 1, cb = tj == 2 ? 2 : 1;
        const float si = ti == 1 ? 0.f : -1.f, sj = tj == 1 ? 0.f : -1.f;      // weight of the (rb) row / (cb) column
        const float sij = si * sj;
        const float* a_stage = As + a_st * A_STAGE;
        const float* b_stage = Bs + b_st * B_STAGE + ph * (QBN * BK);
        int ty_l = tl >> 3, tx_l = tl & 7, l31_l = l31;     // opaque per interval: keeps the fragment addresses out of loop-invariant VGPRs
        asm volatile("" : "+v"(ty_l), "+v"(tx_l), "+v"(l31_l));
        // term (a,b) reads patch pixel (2 Ty + py + a, 2 Tx + px + b) -> position parity*QHALF + row*(QW/2) + half-column
        // LDS float index of a term's 16 bytes at K step s: base ^ (8 s) -- the step only flips bits 1-2 of the
        // XOR-swizzled slot, so one v_xor per read replaces the per-step address arithmetic
        auto term = [&](int a_, int b_) {
            const int m = py + a_, pb = px + b_;
            const int r = 2 * ty_l + m, chn = tx_l + (pb >> 1);
            const int pos = (pb & 1) * QHALF + r * (QW / 2) + chn;
            return pos * BK + ((half ^ patch_swz(r, chn)) << 2);
        };
        const int b00 = term(ra, ca), b01 = term(ra, cb), b10 = term(rb, ca), b11 = term(rb, cb);
        const int bw0 = l31_l * BK + ((half ^ ((l31_l >> 1) & 7)) << 2);
        f32x4 a[2], bb[2][QNT], raw[4];
        auto issue = [&](auto sc_, int buf) {
            constexpr int s = decltype(sc_)::value;
            raw[0] = *reinterpret_cast<const f32x4*>(a_stage + (b00 ^ (8 * s)));
            raw[1] = *reinterpret_cast<const f32x4*>(a_stage + (b01 ^ (8 * s)));
            raw[2] = *reinterpret_cast<const f32x4*>(a_stage + (b10 ^ (8 * s)));
            raw[3] = *reinterpret_cast<const f32x4*>(a_stage + (b11 ^ (8 * s)));
            const float* bt = b_stage + (bw0 ^ (8 * s));
#pragma unroll
            for (int j = 0; j < QNT; ++j) bb[buf][j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * BK);
        };
        auto combine = [&](int buf) {
            f32x4 v = raw[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(sij, raw[3][e], fmaf(si, raw[2][e], fmaf(sj, raw[1][e], v[e])));
            a[buf] = v;
        };
        issue(std::integral_constant<int, 0>{}, 0);
        // the younger wave of each SIMD (waves 4-7: the matrix pipe serves the older one first) issues its DMA pieces
        // before its first MFMA, the older wave interleaves them (same placement as wino4_gemm_kernel)
        const bool young = yf && wave >= QWAVES / 2;
        if (young) {
            if (more_w) static_for<B_INSTR>([&](auto jc) { dma_weight_piece(jc, it_next, b_st ^ 1); });
            if (more_a) dma_patch_piece(xi, cc_next, a_st ^ 1);
        }
        combine(0);
        static_for<4>([&](auto stc) {
            constexpr int step = decltype(stc)::value;
            if constexpr (step + 1 < 4) issue(std::integral_constant<int, step + 1>{}, (step + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4 * QNT>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / QNT, j = q % QNT;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][t], bb[step & 1][j][t], acc[j], 0, 0, 0);
                constexpr int g = step * 4 * QNT + q;   // MFMA index within the interval (32 total)
                // DMA pieces for the next interval: 4 weight pieces, then (intervals 0..A_INSTR-1) one piece of the next patch
                if constexpr (g % 4 == 3 && g / 4 < B_INSTR) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more_w && !young) dma_weight_piece(std::integral_constant<int, g / 4>{}, it_next, b_st ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (g == 4 * B_INSTR + 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more_a && !young) dma_patch_piece(xi, cc_next, a_st ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (step + 1 < 4) combine((step + 1) & 1);
        });
        // fold: Y[oy][ox] += A^T[oy][ti] * A^T[ox][tj] * M, A^T = [1 1 0; 0 1 -1]; zero coefficients are skipped (uniform)
        const float r0 = ti < 2 ? 1.f : 0.f, r1 = ti == 0 ? 0.f : (ti == 1 ? 1.f : -1.f);
        const float c0 = tj < 2 ? 1.f : 0.f, c1 = tj == 0 ? 0.f : (tj == 1 ? 1.f : -1.f);
        const float k0 = r0 * c0, k1 = r0 * c1, k2 = r1 * c0, k3 = r1 * c1;
        auto fold_into = [&](auto oc, float k) {
            constexpr int o = decltype(oc)::value;
            if (k != 0.f) {
                static_for<QNT>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<16>([&](auto rc) { constexpr int r = decltype(rc)::value; Y[o][j][r] = fmaf(k, acc[j][r], Y[o][j][r]); });
                });
            }
        };
        fold_into(std::integral_constant<int, 0>{}, k0);
        fold_into(std::integral_constant<int, 1>{}, k1);
        fold_into(std::integral_constant<int, 2>{}, k2);
        fold_into(std::integral_constant<int, 3>{}, k3);
        static_for<QNT>([&](auto jc) {
            static_for<16>([&](auto rc) { acc[decltype(jc)::value][decltype(rc)::value] = 0.f; });
        });
    };

    // ---- main loop over (channel chunk, transform point)
    static_for<A_INSTR>([&](auto jc) { dma_patch_piece(decltype(jc)::value, 0, 0); });
    static_for<B_INSTR>([&](auto jc) { dma_weight_piece(jc, 0, 0); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int b_st = 0;
    const int nint = cchunks * 9;
    int cc = 0, xi = 0;
    for (int it = 0; it < nint; ++it) {
        const int a_st = cc & 1;
        long long ts0 = 0, ts1 = 0;
        if (p.trace) ts0 = __builtin_readcyclecounter();
        compute(xi, a_st, b_st, it + 1, cc + 1, it + 1 < nint, cc + 1 < cchunks);
        if (p.trace) ts1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.trace && blockIdx.x == 0 && lane == 0 && it < 72) {
            long long* d = p.trace + (wave * 72 + it) * 3;
            d[0] = ts0; d[1] = ts1; d[2] = __builtin_readcyclecounter();
        }
        __syncthreads();
        b_st ^= 1;
        if (++xi == 9) {
            xi = 0;
            ++cc;
        }
    }

    // ---- epilogue: per output (oy,ox) of the tiles, stage the 256 x 64 block through LDS and store 16-byte pieces
    constexpr int LDO = QBN + 4, C4 = QBN / 4, NTHR = QWAVES * 64, PER = 256 * C4 / NTHR;
    const int OH = 2 * p.H, OW = 2 * p.W;
    static_for<4>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        if (o) __syncthreads();
        static_for<QNT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int col = j * 32 + l31;
            const float bias = p.bias[ntile * QBN + col];
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // (phase, tile) row of the block
                smem[row * LDO + col] = Y[o][j][r] + bias;
            });
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / C4, c4 = idx - row * C4;
            const int rw = row >> 5, rph = rw >> 1, rt = (rw & 1) * 32 + (row & 31);
            const int y = ty0 + 2 * (rt >> 3) + (o >> 1), x = tx0 + 2 * (rt & 7) + (o & 1);   // low-resolution pixel
            const int n = ntile * QBN + c4 * 4;
            if (y < p.H && x < p.W && n < p.Cout) {
                float4 v = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                const size_t oo = ((size_t)(b * OH + 2 * y + (rph >> 1)) * OW + 2 * x + (rph & 1)) * p.Cout + n;
                *reinterpret_cast<float4*>(p.out + oo) = v;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t patch_wino_packed_elems(int Cin_packed, int Cout) {
    return (size_t)((Cout + QBN - 1) / QBN) * (Cin_packed / CONV_BK) * 36 * QBN * CONV_BK;
}

// w: 3x3 weights [Cout][Cin][3][3] (BatchNorm folded) -> the four 2x2 phase filters (taps pre-summed in double, as
// patch_pack_host does) -> U = G w_ph G^T, G = [1 0; 1 1; 0 1]; laid out [ntile][cchunk][xi][phase][64][32], swizzled.
void patch_wino_pack_host(const float* w, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst) {
    static const double G[3][2] = {{1, 0}, {1, 1}, {0, 1}};
    const int BK = CONV_BK, cch = cin_packed / BK;
    const size_t total = patch_wino_packed_elems(cin_packed, Cout);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / QBN, nl = o % QBN;
        for (int cp = 0; cp < cin_packed; ++cp) {
            const int c = cin_map ? cin_map[cp] : cp;
            if (c < 0 || c >= Cin) continue;
            const float* s = w + ((size_t)o * Cin + c) * 9;
            const int cc = cp / BK, kl = cp % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                double w4[2][2] = {{0, 0}, {0, 0}};
                for (int ky = 0; ky < 3; ++ky) {
                    const int my = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
                    for (int kx = 0; kx < 3; ++kx) {
                        const int mx = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                        w4[my][mx] += (double)s[ky * 3 + kx];
                    }
                }
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double u = 0.0;
                        for (int t = 0; t < 2; ++t)
                            for (int v = 0; v < 2; ++v) u += G[i][t] * G[j][v] * w4[t][v];
                        const size_t blk = (((size_t)nt * cch + cc) * 9 + (i * 3 + j)) * 4 + ph;
                        dst[(blk * QBN + nl) * BK + kk] = (float)u;
                    }
            }
        }
    }
}

hipError_t patch_wino_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                             float* out, hipStream_t stream, long long* trace) {
    if ((L.C0 % CONV_BK) || (L.C1 % CONV_BK) || (L.Cout & 3) || (H & 1) || (W & 1) || L.w_wino == nullptr)
        return hipErrorInvalidValue;
    PatchWinoArgs a{};
    a.in0 = in0;
    a.in1 = L.C1 ? in1 : nullptr;
    a.C0 = L.C0;
    a.C1 = L.C1;
    const size_t px = (size_t)B * H * W;
    const size_t b0 = px * L.C0 * 4, b1 = px * L.C1 * 4, bw = patch_wino_packed_elems(L.C0 + L.C1, L.Cout) * 4;
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.in0_bytes = (unsigned)b0;
    a.in1_bytes = (unsigned)b1;
    a.w_bytes = (unsigned)bw;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = (W + QT - 1) / QT;
    a.tiles_y = (H + QT - 1) / QT;
    a.ntiles = (L.Cout + QBN - 1) / QBN;
    a.w = L.w_wino;
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.act = act;
    a.out = out;
    static const int yfe = [] { const char* e = getenv("EAMM_PWINO_YOUNG_FIRST"); return e ? atoi(e) : 1; }();
    a.young_first = yfe;
    a.trace = trace;
    constexpr size_t lds_loop = sizeof(float) * 2 * (QPAD * CONV_BK + 4 * QBN * CONV_BK);
    constexpr size_t lds_epi = sizeof(float) * 256 * (QBN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static unsigned long long configured = 0;
    if (hipError_t e = ensure_dynamic_lds(conv_patch_wino_kernel, lds, &configured); e != hipSuccess) return e;
    const int blocks = a.tiles_x * a.tiles_y * B * a.ntiles;
    hipLaunchKernelGGL(conv_patch_wino_kernel, dim3(blocks), dim3(QWAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace eamm
