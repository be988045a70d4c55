// Polyphase minimal-filtering form of the UpBlock2d convolution (reference modules/util.py:883-900: nearest x2 -> 3x3 ->
// BN -> ReLU), products kept in the transform domain for the whole reduction.
//
// In one dimension the two outputs that share a low-resolution pixel are
//     out[2y]   = w0 e[y-1] + (w1 + w2) e[y]          out[2y+1] = (w0 + w1) e[y] + w2 e[y+1]
// -- four multiplies in the collapsed two-phase form.  With the CENTRE pixel as the common term they need three:
//     m0 = e[y] (w0 + w1 + w2),   m1 = (e[y-1] - e[y]) w0,   m2 = (e[y+1] - e[y]) w2;   out[2y] = m0 + m1,  out[2y+1] = m0 + m2.
// In two dimensions nine products per low-resolution pixel and input channel give its four output phases (the collapsed
// phase form needs 16, the reference's convolution of the up-sampled map 36):
//     V = T e T^T,  T = [0 1 0; 1 -1 0; 0 -1 1]      (e: the 3x3 low-resolution neighbourhood; differences with the centre)
//     U = G w G^T,  G = [1 1 1; 1 0 0; 0 0 1]        (w: the BatchNorm-folded 3x3 kernel; computed in double on the host)
//     out(py,px) = sum_ij A[py][i] A[px][j] sum_c (U_ij . V_ij),  A = [1 1 0; 1 0 1]
// but the fold with A is linear and channel-independent, so it is taken out of the channel loop: a wave owns 32 pixels x
// 32 output channels and keeps NINE accumulators (one per transform point, 144 registers) over all input channels; the
// loop is then load - subtract - MFMA only, and the four phases are formed once in the epilogue (3 adds per output).
//   * no product register, no fold adds, nothing of the accumulators is touched by the VALU inside the loop;
//   * the input transform is separable and done per ROW of points: centre row e[1][.] -> points (0, j); e[0][.] - e[1][.]
//     -> points (1, j); e[2][.] - e[1][.] -> points (2, j): 9 LDS reads and 12 subtractions of 16 bytes per K step for
//     36 MFMAs (the per-point form needs 25 reads and 16 subtractions);
//   * all nine points' weights of a 32-channel chunk are 36 KiB: one barrier per chunk (144 MFMAs per wave);
//   * a lane's tile pixel is rotated by two columns on odd rows so that the four 16-lane groups a ds_read_b128 is served
//     in ({0-3,12-15,20-27}, ...) touch sixteen different (pixel & 15) keys of the swizzled patch image: conflict-free.
//   * few tiles (the hourglass decoder's last levels): the channel chunks are split over 2 or 4 workgroups per tile
//     (patch_poly_splits), raw phase sums go to a slab per split and patch_poly_reduce_kernel adds them in a fixed order
//     with bias and activation.
// fp32 throughout; only the summation order differs from the reference (measured at the prediction: DESIGN.md 5.2d).
#include "conv_common.h"

#include <algorithm>
#include <cstdlib>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// a - b on four floats as two v_pk_add_f32 with the second operand negated: hipcc 7.2 turns a vector ADD into packed
// instructions but scalarises a vector SUBTRACT into four v_sub_f32 (172 of them per chunk in this kernel's MFMA stream --
// the f32-input MFMA shares the vector pipe, so every VALU instruction is matrix-pipe time).  Bit-identical to a - b.
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]}, rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return f32x4{rlo[0], rlo[1], rhi[0], rhi[1]};
}

namespace {
constexpr int ZT = 16;                   // tile side in low-resolution pixels
constexpr int ZW = ZT + 2;               // patch side (halo 1)
constexpr int ZPIX = ZW * ZW;            // 324 patch pixels
constexpr int ZPAD = (ZPIX + 7) / 8 * 8; // rounded to whole DMA instructions (8 pixels each)
constexpr int ZBN = 32;                  // output channels per workgroup
constexpr int ZWAVES = 8;
}  // namespace

struct PatchPolyArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] (hourglass skip concatenation) or null
    int C0, C1;
    unsigned in0_bytes, in1_bytes, w_bytes;
    int B, H, W;           // low-resolution input size; output is [B,2H,2W,Cout]
    int tiles_x, tiles_y, ntiles;
    const float* w;        // packed [ntile][cchunk][point 3i+j][32][32], swizzled
    const float* bias;     // [>= ntiles*32]
    int Cout, act;
    float* out;
    int splits;            // > 1: workgroup (.., ks) reduces channel chunks [ks, ks+1) * cchunks / splits and writes raw phase sums
    float* partial;        // [splits][B,2H,2W,Cout]; patch_poly_reduce_kernel adds the slabs, the bias and the activation
    size_t slab;           // floats per slab
};

__global__ __launch_bounds__(ZWAVES * 64) void conv_patch_poly_kernel(const PatchPolyArgs p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = ZPAD * BK;            // floats (41 KiB)
    constexpr int B_STAGE = 9 * ZBN * BK;         // floats (36 KiB)
    constexpr int A_PIECES = ZPAD / 8, B_PIECES = 9 * ZBN / 8;   // 1 KiB DMA pieces per chunk: 41 + 36
    constexpr int A_INSTR = (A_PIECES + ZWAVES - 1) / ZWAVES;    // per wave: 6
    constexpr int B_INSTR = (B_PIECES + ZWAVES - 1) / ZWAVES;    //           5
    static_assert(A_INSTR + B_INSTR <= 12, "one DMA piece per sub-step");
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
    const int ks = L % p.splits;
    L /= p.splits;
    const int tx0 = (L % p.tiles_x) * ZT;
    L /= p.tiles_x;
    const int ty0 = (L % p.tiles_y) * ZT;
    const int b = L / p.tiles_y;
    const int cchunks = (p.C0 + p.C1) / BK;
    const int c_begin = ks * cchunks / p.splits, c_end = (ks + 1) * cchunks / p.splits;

    // ---- loaders.  Patch piece id = wave + 8 j covers patch pixels 8 id .. 8 id + 7, 128 B each (lane>>3 = pixel,
    // lane&7 = 16-byte slot, XOR-swizzled with the pixel on the global side); the pixel's image index is chunk-invariant.
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int pix[A_INSTR];      // (b*H + y)*W + x of this lane's pixel in piece j, or -1
    static_for<A_INSTR>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int i = (wave + ZWAVES * j) * 8 + (lane >> 3);
        const int pyy = i / ZW, pxx = i - pyy * ZW;
        const int y = ty0 + pyy - 1, x = tx0 + pxx - 1;
        const bool ok = i < ZPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        pix[j] = ok ? (b * p.H + y) * p.W + x : -1;
    });
    // ((8 id + (lane>>3)) >> 1) & 7 does not depend on j
    const unsigned slot_b = (unsigned)(((lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7)) << 4);
    auto dma_patch_piece = [&](auto jc, int cc, int st) {
        constexpr int j = decltype(jc)::value;
        if (wave + ZWAVES * j < A_PIECES) {
            const int c0 = cc * BK;
            const bool first = c0 < p.C0;
            const int C = first ? p.C0 : p.C1;
            const unsigned so = (unsigned)(first ? c0 : c0 - p.C0) * 4u;
            const int pj = pix[j];
            const unsigned vo = pj >= 0 ? (unsigned)(pj * C) * 4u + slot_b : OOB;
            float* dst = As + st * A_STAGE + (wave + ZWAVES * j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)dst, 16, vo, so, 0, 0);
        }
    };
    // weights of chunk cc: 36 KiB contiguous; per-lane part fixed, chunk / piece part wave-uniform
    const unsigned w_lane = (unsigned)lane * 16u;
    auto dma_weight_piece = [&](auto jc, int cc, int st) {
        constexpr int j = decltype(jc)::value;
        if (wave + ZWAVES * j < B_PIECES) {
            const unsigned so = (unsigned)((ntile * cchunks + cc) * B_STAGE + (wave + ZWAVES * j) * (8 * BK)) * 4u;
            float* dst = Bs + st * B_STAGE + (wave + ZWAVES * j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)dst, 16, w_lane, so, 0, 0);
        }
    };

    f32x16 acc[9];         // one per transform point (i, j) -> 3 i + j
    static_for<9>([&](auto kc) {
        static_for<16>([&](auto rc) { acc[decltype(kc)::value][decltype(rc)::value] = 0.f; });
    });

    // this lane's pixel inside the tile: wave w owns rows 2w, 2w+1; odd rows rotated by two columns (bank groups, see top).
    // LDS float address of the 16 bytes of patch pixel idx at K step s: (idx*32 + ((half ^ (idx>>1)&7) << 2)) ^ (8 s)
    const int trow = 2 * wave + (l31 >> 4), tcol = ((l31 & 15) - 2 * (l31 >> 4)) & 15;
    const int idx0 = (trow + 1) * ZW + tcol + 1;
    int ta[9];
    static_for<9>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int idx = idx0 + (k / 3 - 1) * ZW + (k % 3 - 1);
        ta[k] = idx * BK + ((half ^ ((idx >> 1) & 7)) << 2);
    });
    const int bw0 = l31 * BK + ((half ^ ((l31 >> 1) & 7)) << 2);   // weight row of this lane inside a point's [32][32] tile

    // Sub-step q = 3 s + i of a chunk: K step s (8 channels), row i of transform points.
    f32x4 x1[3];           // centre row of the neighbourhood at the current K step
    f32x4 xr[3];           // row above / below
    f32x4 v[2][3];         // transformed fragments of a row of points
    f32x4 bb[2][3];        // their weights
    auto fetch = [&](auto qc, const float* a_stage, const float* b_stage) {
        constexpr int q = decltype(qc)::value, s = q / 3, i = q % 3;
        constexpr int row = i == 0 ? 1 : (i == 1 ? 0 : 2);
        static_for<3>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const f32x4 t = *reinterpret_cast<const f32x4*>(a_stage + (ta[3 * row + j] ^ (8 * s)));
            if constexpr (i == 0) x1[j] = t; else xr[j] = t;
        });
        static_for<3>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            bb[q & 1][j] = *reinterpret_cast<const f32x4*>(b_stage + (3 * i + j) * (ZBN * BK) + (bw0 ^ (8 * s)));
        });
    };
    auto transform = [&](auto qc) {
        constexpr int q = decltype(qc)::value, i = q % 3;
        f32x4 r0, r1, r2;
        if constexpr (i == 0) { r0 = x1[0]; r1 = x1[1]; r2 = x1[2]; }
        else { r0 = pk_sub4(xr[0], x1[0]); r1 = pk_sub4(xr[1], x1[1]); r2 = pk_sub4(xr[2], x1[2]); }
        v[q & 1][0] = r1;
        v[q & 1][1] = r0 - r1;
        v[q & 1][2] = r2 - r1;
    };
    auto mfmas = [&](auto qc, auto hc) {   // half hc (0, 1) of the 12 MFMAs of sub-step q: K pairs t = 2 hc, 2 hc + 1
        constexpr int q = decltype(qc)::value, i = q % 3, h = decltype(hc)::value;
        static_for<2>([&](auto tc) {
            constexpr int t = 2 * h + decltype(tc)::value;
            static_for<3>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q & 1][j][t], bb[q & 1][j][t], acc[3 * i + j], 0, 0, 0);
            });
        });
    };

    auto chunk = [&](int cc) {
        const int st = (cc - c_begin) & 1;
        const float* a_stage = As + st * A_STAGE;
        const float* b_stage = Bs + st * B_STAGE;
        const bool more = cc + 1 < c_end;
        fetch(std::integral_constant<int, 0>{}, a_stage, b_stage);
        transform(std::integral_constant<int, 0>{});
        static_for<12>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + 1 < 12) fetch(std::integral_constant<int, q + 1>{}, a_stage, b_stage);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(qc, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q + 1 < 12) transform(std::integral_constant<int, q + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(qc, std::integral_constant<int, 1>{});
            if constexpr (q < B_INSTR) {
                if (more) dma_weight_piece(std::integral_constant<int, q>{}, cc + 1, st ^ 1);
            } else if constexpr (q - B_INSTR < A_INSTR) {
                if (more) dma_patch_piece(std::integral_constant<int, q - B_INSTR>{}, cc + 1, st ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- main loop over channel chunks
    static_for<A_INSTR>([&](auto jc) { dma_patch_piece(jc, c_begin, 0); });
    static_for<B_INSTR>([&](auto jc) { dma_weight_piece(jc, c_begin, 0); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int cc = c_begin; cc < c_end; ++cc) chunk(cc);

    // ---- epilogue: form each phase (3 adds per output), stage its 256 x 32 tile through LDS, store 16-byte pieces
    constexpr int LDO = ZBN + 4, C4 = ZBN / 4, NTHR = ZWAVES * 64, PER = ZT * ZT * C4 / NTHR;
    const int OH = 2 * p.H, OW = 2 * p.W;
    const bool raw = p.splits > 1;
    const float bias = raw ? 0.f : p.bias[ntile * ZBN + l31];
    float* const dst = raw ? p.partial + (size_t)ks * p.slab : p.out;
    static_for<4>([&](auto phc) {
        constexpr int ph = decltype(phc)::value, py = ph >> 1, px = ph & 1;
        if constexpr (ph > 0) __syncthreads();
        static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;                 // MFMA row of this register
            const int pixel = (2 * wave + (m >> 4)) * ZT + (((m & 15) - 2 * (m >> 4)) & 15);
            const float y = (acc[0][r] + acc[1 + px][r]) + (acc[3 * (1 + py)][r] + acc[3 * (1 + py) + 1 + px][r]);
            smem[pixel * LDO + l31] = y + bias;
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / C4, c4 = idx - row * C4;
            const int y = ty0 + (row >> 4), x = tx0 + (row & 15);
            const int n = ntile * ZBN + c4 * 4;
            if (y < p.H && x < p.W && n < p.Cout) {
                float4 o4 = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                if (!raw) {
                    o4.x = apply_act(o4.x, p.act); o4.y = apply_act(o4.y, p.act);
                    o4.z = apply_act(o4.z, p.act); o4.w = apply_act(o4.w, p.act);
                }
                const size_t o = ((size_t)(b * OH + 2 * y + py) * OW + 2 * x + px) * p.Cout + n;
                *reinterpret_cast<float4*>(dst + o) = o4;
            }
        }
    });
}

// out = act(sum of the split-K slabs + bias); 16 bytes per thread and pass, slabs added in a fixed order
__global__ __launch_bounds__(256) void patch_poly_reduce_kernel(const float* __restrict__ partial, int splits, size_t slab,
                                                                const float* __restrict__ bias, int Cout, int act,
                                                                float* __restrict__ out, size_t total4) {
    const int c4n = Cout >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(partial)[i];
        for (int sp = 1; sp < splits; ++sp) {
            const float4 t = reinterpret_cast<const float4*>(partial + (size_t)sp * slab)[i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const float4 bs = reinterpret_cast<const float4*>(bias)[i % c4n];
        v.x = apply_act(v.x + bs.x, act); v.y = apply_act(v.y + bs.y, act);
        v.z = apply_act(v.z + bs.z, act); v.w = apply_act(v.w + bs.w, act);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// Split of the channel reduction over workgroups when the 16x16-pixel x 32-cout tiles alone leave CUs idle (the
// hourglass decoder's last levels): the largest of {4, 2} that keeps one round of the chip and >= 4 chunks per workgroup.
int patch_poly_splits(const PatchLayer& L, int B, int H, int W, int max_splits, int cus) {
    const int blocks = ((W + ZT - 1) / ZT) * ((H + ZT - 1) / ZT) * B * ((L.Cout + ZBN - 1) / ZBN);
    const int cchunks = (L.C0 + L.C1) / CONV_BK;
    // fewest channel chunks per workgroup: 4; calls of one or two frames (128 workgroups per up block on 256 CUs) go down to 2
    // (one frame 1071 -> 1094 frames/s; at 8-frame chains the extra workgroups compete with the other chain: 3848 vs 3837)
    static const int min_env = (int)knob_int("EAMM_PATCH_SPLIT_MIN_CHUNKS", 0);
    const int min_chunks = min_env > 0 ? min_env : (B <= 2 ? 2 : 4);
    for (int sp : {4, 2})
        if (sp <= max_splits && blocks * sp <= cus && cchunks % sp == 0 && cchunks / sp >= min_chunks) return sp;
    return 1;
}

size_t patch_poly_packed_elems(int Cin_packed, int Cout) {
    return (size_t)((Cout + ZBN - 1) / ZBN) * (Cin_packed / CONV_BK) * 9 * ZBN * CONV_BK;
}

// w: 3x3 weights [Cout][Cin][3][3] (BatchNorm folded) -> U = G w G^T, G = [1 1 1; 1 0 0; 0 0 1], laid out
// [ntile][cchunk][point 3i+j][32][32], LDS swizzle applied.
void patch_poly_pack_host(const float* w, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst) {
    static const double G[3][3] = {{1, 1, 1}, {1, 0, 0}, {0, 0, 1}};
    const int BK = CONV_BK, cch = cin_packed / BK;
    const size_t total = patch_poly_packed_elems(cin_packed, Cout);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / ZBN, nl = o % ZBN;
        for (int cp = 0; cp < cin_packed; ++cp) {
            const int c = cin_map ? cin_map[cp] : cp;
            if (c < 0 || c >= Cin) continue;
            const float* s = w + ((size_t)o * Cin + c) * 9;
            const int cc = cp / BK, kl = cp % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int pt = 0; pt < 9; ++pt) {
                const int i = pt / 3, j = pt % 3;
                double u = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) u += G[i][a] * G[j][b2] * (double)s[a * 3 + b2];
                const size_t tile = ((size_t)nt * cch + cc) * 9 + pt;
                dst[(tile * ZBN + nl) * BK + kk] = (float)u;
            }
        }
    }
}

hipError_t patch_poly_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                              float* out, hipStream_t stream, int splits, float* partial, size_t partial_cap) {
    if ((L.C0 % CONV_BK) || (L.C1 % CONV_BK) || (L.Cout & 3) || L.w_poly == nullptr) return hipErrorInvalidValue;
    const size_t slab = (size_t)B * 4 * H * W * L.Cout;
    if (splits < 1 || (splits > 1 && (partial == nullptr || (size_t)splits * slab > partial_cap))) return hipErrorInvalidValue;
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
    a.tiles_x = (W + ZT - 1) / ZT;
    a.tiles_y = (H + ZT - 1) / ZT;
    a.ntiles = (L.Cout + ZBN - 1) / ZBN;
    a.w = L.w_poly;
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.act = act;
    a.out = out;
    a.splits = splits;
    a.partial = partial;
    a.slab = slab;
    constexpr size_t lds_loop = sizeof(float) * 2 * (ZPAD * CONV_BK + 9 * ZBN * CONV_BK);
    constexpr size_t lds_epi = sizeof(float) * (ZT * ZT) * (ZBN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static lds_once_mask configured{0};
    if (hipError_t e = ensure_dynamic_lds(conv_patch_poly_kernel, lds, &configured); e != hipSuccess) return e;
    const int blocks = a.tiles_x * a.tiles_y * B * a.ntiles * splits;
    note_mfma_flops(2.0 * a.tiles_x * a.tiles_y * B * (ZT * ZT) * 9.0 * (a.ntiles * ZBN) * (L.C0 + L.C1));
    hipLaunchKernelGGL(conv_patch_poly_kernel, dim3(blocks), dim3(ZWAVES * 64), lds, stream, a);
    if (splits > 1) {
        const size_t total4 = slab / 4;
        const int rb = (int)std::min<size_t>((total4 + 255) / 256, 4096);
        hipLaunchKernelGGL(patch_poly_reduce_kernel, dim3(rb), dim3(256), 0, stream, partial, splits, slab, L.bias, L.Cout, act, out,
                           total4);
    }
    return hipGetLastError();
}

}  // namespace eamm
