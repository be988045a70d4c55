// Winograd F(4x4, 3x3) form of the 3x3 / pad 1 convolution for the bottleneck ResBlock2d stack
// (reference modules/util.py:858-880, generator.py:89): 36 multiplies per 16 outputs and input channel instead of
// 144 (direct) or 64 (F(2x2,3x3), conv_winograd.hip) -- 4x fewer MFMA passes than the direct form, still plain fp32
// arithmetic (v_mfma_f32_16x16x4_f32 + fp32 VALU), only the summation order / rounding differs (measured at the
// prediction: tools/wino_numerics.py).
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        d: 6x6 input patch (stride 4), Y: 4x4 outputs
//
// Interpolation points {0, +-1, +-2, inf} (Lavin & Gray):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// Two kernels per convolution, as in the F(2x2) form:
//  1. wino4_input_transform_kernel (HBM-bound): V[xi][tile][c], xi = 6i + j, optionally applying the ResBlock
//     pre-activation relu(x*s + t) to in-range pixels first.  V is 2.25x the activation (F(2x2): 4x).
//  2. wino4_gemm_kernel (MFMA-bound): for each xi the GEMM M_xi[tile, o] = sum_c V_xi[tile, c] U_xi[c, o]; when a xi
//     is finished its accumulator is folded along x into Z[q] += A^T[q][j] M_xi, and after the six xi of a row i
//     along y into the sixteen output accumulators Y[p][q] += A^T[p][i] Z[q] -- the transformed products never leave
//     the registers.  The sixteen outputs bound the tile a CU can hold (16 outputs x 64x64 tile x 4 B = half the
//     register file), so a wave owns a 16 x 32 tile of (Winograd tile, output channel) and uses the 16x16x4 MFMA;
//     the LDS image, swizzle and LDS-DMA pipeline are those of conv_mfma_dma.hip / conv_winograd.hip.
//
// U = G g G^T is computed in double on the host from the BatchNorm-folded weights and rounded once.
#include "conv_common.h"

#include <algorithm>
#include <cstdlib>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// input transform
// ---------------------------------------------------------------------------------------------------------
// VEC floats per thread (ext vector: arithmetic is element-wise)
template <int VEC> struct vec_of;
template <> struct vec_of<1> { typedef float type; };
template <> struct vec_of<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct vec_of<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <typename T> __device__ __forceinline__ T vfma(float a, T x, T y);
template <> __device__ __forceinline__ float vfma<float>(float a, float x, float y) { return fmaf(a, x, y); }
template <> __device__ __forceinline__ vec_of<2>::type vfma(float a, vec_of<2>::type x, vec_of<2>::type y) {
    vec_of<2>::type r;
    r[0] = fmaf(a, x[0], y[0]); r[1] = fmaf(a, x[1], y[1]);
    return r;
}
template <> __device__ __forceinline__ vec_of<4>::type vfma(float a, vec_of<4>::type x, vec_of<4>::type y) {
    vec_of<4>::type r;
    r[0] = fmaf(a, x[0], y[0]); r[1] = fmaf(a, x[1], y[1]); r[2] = fmaf(a, x[2], y[2]); r[3] = fmaf(a, x[3], y[3]);
    return r;
}
template <typename T> __device__ __forceinline__ T vrelu_affine(T v, T sc, T sh);
template <> __device__ __forceinline__ float vrelu_affine<float>(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }
template <> __device__ __forceinline__ vec_of<2>::type vrelu_affine(vec_of<2>::type v, vec_of<2>::type sc, vec_of<2>::type sh) {
    vec_of<2>::type r;
    r[0] = fmaxf(fmaf(v[0], sc[0], sh[0]), 0.f); r[1] = fmaxf(fmaf(v[1], sc[1], sh[1]), 0.f);
    return r;
}
template <> __device__ __forceinline__ vec_of<4>::type vrelu_affine(vec_of<4>::type v, vec_of<4>::type sc, vec_of<4>::type sh) {
    vec_of<4>::type r;
    r[0] = fmaxf(fmaf(v[0], sc[0], sh[0]), 0.f); r[1] = fmaxf(fmaf(v[1], sc[1], sh[1]), 0.f);
    r[2] = fmaxf(fmaf(v[2], sc[2], sh[2]), 0.f); r[3] = fmaxf(fmaf(v[3], sc[3], sh[3]), 0.f);
    return r;
}

// (B^T d) for six values in place
template <typename T> __device__ __forceinline__ void bt6(T& d0, T& d1, T& d2, T& d3, T& d4, T& d5) {
    const T s12 = d1 + d2, m12 = d1 - d2;
    const T s34 = d3 + d4, m43 = d4 - d3;
    const T m31 = d3 - d1, m42 = d4 - d2;
    const T t0 = vfma(4.f, d0, vfma(-5.f, d2, d4));
    const T t1 = vfma(-4.f, s12, s34);
    const T t2 = vfma(4.f, m12, m43);
    const T t3 = vfma(2.f, m31, m42);
    const T t4 = vfma(-2.f, m31, m42);
    const T t5 = vfma(4.f, d1, vfma(-5.f, d3, d5));
    d0 = t0; d1 = t1; d2 = t2; d3 = t3; d4 = t4; d5 = t5;
}

// One thread per (tile, VEC channels): a wave covers 64*VEC contiguous channels of one tile.
template <int VEC>
__global__ __launch_bounds__(256) void wino4_input_transform_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ s,
                                                                    const float* __restrict__ t, int B, int H, int W,
                                                                    int C, float* __restrict__ V, int by_xcd) {
    typedef typename vec_of<VEC>::type T;
    const int cvn = C / VEC;
    const int Hq = H >> 2, Wq = W >> 2;
    const size_t Mq = (size_t)B * Hq * Wq;
    const size_t total = Mq * cvn;
    // XCD-aware order (round 5): workgroups are dealt to the eight XCDs round-robin, so consecutive workgroups -- neighbouring
    // tiles, whose 6x6 patches share two of six rows / columns -- land in eight different L2s and every L2 fetches the shared
    // pixels again (PMC: 50.9 MB fetched per 8-frame launch for 16.8 MB of input).  Remapped, XCD x walks the x-th EIGHTH of the
    // tiles: at 8 frames one whole frame per XCD, every overlap inside one L2 (measured in the pipeline: 34.8 -> 31.7 us per launch,
    // 3953 -> 3998 frames/s; with the non-temporal V stores below 26.9 us, 4018 frames/s).  (Placement is a speed hint only.)
    const size_t bid = (by_xcd & 1) ? (size_t)xcd_remap((int)blockIdx.x, (int)gridDim.x) : (size_t)blockIdx.x;
    for (size_t idx = bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % cvn);
        const size_t q = idx / cvn;
        const int qx = (int)(q % Wq);
        const int qy = (int)((q / Wq) % Hq);
        const int b = (int)(q / ((size_t)Wq * Hq));
        T sc = T(1.f), sh = T(0.f);
        if (s != nullptr) {
            sc = reinterpret_cast<const T*>(s)[cv];
            sh = reinterpret_cast<const T*>(t)[cv];
        }
        const T* img = reinterpret_cast<const T*>(x) + (size_t)b * H * W * cvn + cv;
        T d[6][6];
        // all 36 loads are issued before the first use: out-of-range pixels read a clamped (valid) address and are zeroed
        // afterwards -- a bounds branch around each load made the compiler wait for every load before issuing the next
        int yo[6], xo[6];
        bool oky[6], okx[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = 4 * qy - 1 + i, xx = 4 * qx - 1 + i;
            oky[i] = (unsigned)yy < (unsigned)H;
            okx[i] = (unsigned)xx < (unsigned)W;
            yo[i] = min(max(yy, 0), H - 1) * W;
            xo[i] = min(max(xx, 0), W - 1);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) d[i][j] = img[(size_t)(yo[i] + xo[j]) * cvn];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                T v = d[i][j];
                if (s != nullptr) v = vrelu_affine(v, sc, sh);  // zero padding applies to the ACTIVATED tensor
                d[i][j] = (oky[i] && okx[j]) ? v : T(0.f);
            }
#pragma unroll
        for (int j = 0; j < 6; ++j) bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);   // along y
        T* out = reinterpret_cast<T*>(V) + q * cvn + cv;
        const size_t plane = Mq * cvn;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);                              // along x
            if constexpr (VEC == 4) {
                if (by_xcd & 2) {   // V leaves through NON-TEMPORAL stores: it is 2.25 x the input, is not read again by this kernel, and
                    // written through the L2 it evicts the input pixels the neighbouring tiles are about to share (26.9 vs 31.8 us
                    // per launch in the pipeline; the GEMM that reads V next pays 2 us for finding less of it cached: + 0.8 % frames/s)
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (unsigned)(36 * plane * 16), 0x00020000);
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d[i][j]), rs,
                                                               (unsigned)(((size_t)(i * 6 + j) * plane + q * cvn + cv) * 16), 0, 2);
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * plane] = d[i][j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM over the 36 transform points + in-register output transform
// ---------------------------------------------------------------------------------------------------------
struct Wino4Args {
    const float* V;        // [36][Mq][C]
    const float* U;        // packed [ntiles][36 * C/32][BN][32] (swizzled), xi outer / channel chunk inner
    const float* bias;     // [ntiles*BN]
    unsigned v_bytes, u_bytes;
    int Mq, C, Cout;       // tiles (= pixels / 16), input channels, output channels
    int H, W;              // output image size (4*Hq, 4*Wq)
    int mtiles, ntiles;
    int act;
    const float* resid;    // NHWC [B,H,W,Cout]
    float* out;            // NHWC [B,H,W,Cout]
    int groups;            // > 1: the 36 transform points are split over `groups` workgroups per tile (small M): 2 / 3 / 6 = whole rows of six,
                           // 12 = HALF rows (round 6: one frame per call = 16 blocks of 64 x 64 -> 192 workgroups of the eight-wave kernel)
    float* zout;           // groups > 1: [24][Mq][Cout] x-folded products Z[i][q] (groups == 12: [48]: slot 2 i + half holds the partial x fold
                           // of its three points), finished by wino4_output_transform_kernel
    float* epi_scratch;    // DBG 30 (timing experiment, EAMM_WINO4_EPI_V): the epilogue also writes 2.25 x the output here
};

__constant__ float WINO4_AT[4][8] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f},
                                     {0.f, 1.f, -1.f, 2.f, -2.f, 0.f, 0.f, 0.f},
                                     {0.f, 1.f, 1.f, 4.f, 4.f, 0.f, 0.f, 0.f},
                                     {0.f, 1.f, -1.f, 8.f, -8.f, 1.f, 0.f, 0.f}};

// A wave owns 16 Winograd tiles x NT*16 output channels; WM x WN waves per block.  SUB: 32-channel chunks per
// barrier interval; NST: LDS ring depth in intervals; PE: one DMA piece every PE MFMAs from the start of an interval.
template <int NT, int WM, int WN, int SUB, int NST, int PE, int DBG = 0>
__global__ __launch_bounds__(WM* WN * 64) void wino4_gemm_kernel(const Wino4Args p) {
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int BM = WM * 16, BN = WN * NT * 16, BK = CONV_BK;
    constexpr int A_STAGE = SUB * BM * BK, B_STAGE = SUB * BN * BK;  // floats per stage
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;       // DMA instructions per wave per chunk
    constexpr int NPIECE = SUB * (A_INSTR + B_INSTR);
    constexpr bool TRACE = DBG == 8 || DBG == 9;   // DBG: 0 product; 1 no DMA in the loop; 8/9 interval trace (with/without DMA);
    constexpr bool YOUNG_FIRST = DBG != 20;        // 20: every wave interleaves its pieces (the first version)
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    static_assert(BK == 32, "two 16-wide K steps per chunk");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][A_STAGE] [NST][B_STAGE]
    float* const As = smem;
    float* const Bs = smem + NST * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    const int mtile = (L / p.ntiles) % p.mtiles;
    const int grp = L / (p.ntiles * p.mtiles);          // which transform points (0 when groups == 1)
    const int xis_per = 36 / p.groups;                   // 36, 18, 12, 6: whole rows; 3 (groups == 12): half a row
    const int xi0 = grp * xis_per;                       // first transform point of this workgroup
    const int mbase = mtile * BM;
    const int cchunks = p.C / BK;                         // channel chunks per transform point
    const int ci_base = xi0 * cchunks;                    // first chunk of this workgroup's transform points
    const int nsuper = xis_per * cchunks / SUB;           // barrier intervals

    // DMA addressing.  Every piece is ONE buffer_load ... lds whose address splits into a per-lane VGPR part that never
    // changes (the lane's row and 16-byte slot) and a wave-uniform SGPR part that advances by a constant per interval
    // (transform point / channel chunk for V, chunk index for U) -- no per-piece integer division or vector arithmetic
    // in the MFMA stream (the first version recomputed xi = ci / cchunks per piece: ~25 scalar instructions each).
    unsigned arow_off[A_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        const int m = mbase + row;
        arow_off[j] = m < p.Mq ? (unsigned)(m * p.C + slot * 4) * 4u : 0xFFFFF000u;   // M tail: out of range (+ the immediate cannot wrap it)
    }
    const unsigned b_lane = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);
    const unsigned plane_bytes = (unsigned)p.Mq * (unsigned)p.C * 4u;
    const unsigned a_wrap = plane_bytes - (unsigned)(cchunks * BK * 4);   // from the last chunks of xi to the first of xi + 1

    // uniform state of the interval whose pieces are issued next
    unsigned sa_off = (unsigned)xi0 * plane_bytes;                                                 // (xi, cc) in V
    unsigned sb_off = (unsigned)((ntile * 36 * cchunks + ci_base) * BN * BK) * 4u +               // chunk ci in U
                      (unsigned)(wave * B_INSTR * 8 * BK) * 4u;                                    // + this wave's rows
    int dma_cc = 0, n_st = 0;
    auto dma_advance = [&]() {
        dma_cc += SUB;
        sa_off += SUB * BK * 4;
        if (dma_cc == cchunks) {
            dma_cc = 0;
            sa_off += a_wrap;
        }
        sb_off += SUB * BN * BK * 4;
        n_st = n_st + 1 == NST ? 0 : n_st + 1;
    };
    auto dma_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int sub = k / (A_INSTR + B_INSTR), r = k % (A_INSTR + B_INSTR);
        if constexpr (r < A_INSTR) {
            // the instruction's immediate offset moves the LDS address as well as the global one: take it off the LDS side
            float* dst = As + n_st * A_STAGE + sub * (BM * BK) + (wave * A_INSTR + r) * (8 * BK) - sub * BK;
            // (a local copy: passing the captured array element straight to the builtin makes hipcc 7.2's HOST pass drop the
            // kernel's stub without a diagnostic -- the library then fails to load with an undefined kernel symbol)
            const unsigned vo = arow_off[r];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_ptr_t)dst, 16, vo, sa_off, sub * BK * 4, DBG == 40 ? 2 : 0);   // 40 (variant 6): V is streamed once per XCD -- non-temporal
        } else {
            constexpr int j = r - A_INSTR;
            float* dst = Bs + n_st * B_STAGE + sub * (BN * BK) + (wave * B_INSTR + j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)dst, 16, b_lane,
                                                     sb_off + (unsigned)((sub * BN + j * 8) * BK * 4), 0, 0);   // (U non-temporal too: 3955 vs 4048 frames/s -- every XCD's four tile rows share it)
        }
    };

    f32x4_t acc[NT];        // M_xi of the transform point in flight
    f32x4_t Z[4][NT];       // row i folded along x
    f32x4_t Y[16][NT];      // the sixteen outputs of each tile, p = 4*py + px
    static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            acc[j][r] = 0.f;
            static_for<4>([&](auto qc) { Z[decltype(qc)::value][j][r] = 0.f; });
            static_for<16>([&](auto pc) { Y[decltype(pc)::value][j][r] = 0.f; });
        });
    });

    const int sw = (r16 >> 1) & 7;
    constexpr int STEPS = SUB * 2;               // 16-wide K steps per barrier interval
    constexpr int MF = 4 * NT;                   // MFMAs per step
    constexpr int TOTAL_MF = STEPS * MF;
    static_assert(PE >= 1 && NPIECE * PE <= TOTAL_MF, "DMA pieces must fit in the interval");
    // MORE: the interval issues the DMA pieces of the interval D ahead; YOUNG: all of them before the first MFMA.
    // Both are wave-uniform and resolved ONCE per interval (interval() below), so the MFMA stream carries no branches.
    auto compute = [&](int st, auto more_c, auto young_c) {
        constexpr bool MORE = decltype(more_c)::value && DBG != 1 && DBG != 9, YOUNG = decltype(young_c)::value;
        f32x4_t a[2], b[2][NT];
        auto fetch = [&](int step, int buf) {
            const int sub = step >> 1, s = step & 1;
            const int slot = ((4 * s + g) ^ sw) << 2;
            const float* a_base = As + st * A_STAGE + sub * (BM * BK) + (wm * 16 + r16) * BK + slot;
            const float* b_base = Bs + st * B_STAGE + sub * (BN * BK) + (wn * NT * 16 + r16) * BK + slot;
            a[buf] = *reinterpret_cast<const f32x4_t*>(a_base);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[buf][j] = *reinterpret_cast<const f32x4_t*>(b_base + j * 16 * BK);
        };
        fetch(0, 0);
        // The matrix pipe serves the older wave of a SIMD first (waves 0..NW/2-1 finish an interval's MFMAs
        // before waves NW/2.. get going), and a younger wave's interleaved pieces are gated by its own MFMA
        // progress, so they used to issue late, in the tail the older wave no longer covers.  The younger half
        // therefore issues its pieces before its first MFMA, while it would be waiting for the pipe anyway
        // (measured: 2.29 -> 2.25 ms per step; moving the older half's pieces as well, or more pieces to one
        // half, is slower -- profiles/r01_wino4_gemm_investigation.txt).
        if constexpr (MORE && YOUNG) static_for<NPIECE>([&](auto kc) { dma_piece(kc); });
        static_for<STEPS>([&](auto sc) {
            constexpr int step = decltype(sc)::value;
            if constexpr (step + 1 < STEPS) fetch(step + 1, (step + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<MF>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / NT, j = q % NT;
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[step & 1][t], b[step & 1][j][t], acc[j], 0, 0, 0);
                constexpr int gi = step * MF + q;
                if constexpr (MORE && !YOUNG && gi % PE == PE - 1 && gi / PE < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_piece(std::integral_constant<int, gi / PE>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };
    const bool young = YOUNG_FIRST && wave >= NW / 2;
    auto interval = [&](int st, bool more) {
        if (more) {
            if (young) compute(st, std::true_type{}, std::true_type{});
            else compute(st, std::true_type{}, std::false_type{});
            dma_advance();
        } else {
            compute(st, std::false_type{}, std::false_type{});
        }
    };

    // (round 3: computing the column of A^T with scalar selects on j instead of these four s_loads measured SLOWER, 3826 -> 3785
    // frames/s: hipcc issues the loads ahead of the interval's last MFMAs, the select chain sits in the gap)
    auto fold_x = [&](int j) {   // Z[q] += A^T[q][j] * M_ij
        const float c0 = WINO4_AT[0][j], c1 = WINO4_AT[1][j], c2 = WINO4_AT[2][j], c3 = WINO4_AT[3][j];
        static_for<NT>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            static_for<4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float m = acc[jj][r];
                Z[0][jj][r] = fmaf(c0, m, Z[0][jj][r]);
                Z[1][jj][r] = fmaf(c1, m, Z[1][jj][r]);
                Z[2][jj][r] = fmaf(c2, m, Z[2][jj][r]);
                Z[3][jj][r] = fmaf(c3, m, Z[3][jj][r]);
                acc[jj][r] = 0.f;
            });
        });
    };
    auto fold_y = [&](int i) {   // Y[p][q] += A^T[p][i] * Z[q]
        const float c0 = WINO4_AT[0][i], c1 = WINO4_AT[1][i], c2 = WINO4_AT[2][i], c3 = WINO4_AT[3][i];
        static_for<NT>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float z = Z[q][jj][r];
                    Y[0 + q][jj][r] = fmaf(c0, z, Y[0 + q][jj][r]);
                    Y[4 + q][jj][r] = fmaf(c1, z, Y[4 + q][jj][r]);
                    Y[8 + q][jj][r] = fmaf(c2, z, Y[8 + q][jj][r]);
                    Y[12 + q][jj][r] = fmaf(c3, z, Y[12 + q][jj][r]);
                    Z[q][jj][r] = 0.f;
                });
            });
        });
    };

    // split form (few tiles): row i is complete in x -> its four Z[q] go to HBM, the y fold happens in
    // wino4_output_transform_kernel over all six rows
    auto store_z = [&](int i) {
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            static_for<NT>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                const int col = ntile * BN + wn * NT * 16 + jj * 16 + r16;
                static_for<4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int row = mbase + wm * 16 + 4 * g + r;
                    if (row < p.Mq && col < p.Cout) p.zout[((size_t)(i * 4 + q) * p.Mq + row) * p.Cout + col] = Z[q][jj][r];
                    Z[q][jj][r] = 0.f;
                });
            });
        });
    };

    // ---- main loop: ring of NST stages, DMA runs D = NST-1 intervals ahead of the MFMAs
    constexpr int D = NST - 1;
    const int per_xi = cchunks / SUB;  // barrier intervals per transform point
    for (int k = 0; k < D && k < nsuper; ++k) {
        static_for<NPIECE>([&](auto kc) { dma_piece(kc); });
        dma_advance();
    }
    if (D > 1 && nsuper >= D) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NPIECE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int st = 0;
    int xi_left = per_xi, xj = xi0 % 6, xrow = xi0 / 6;
    const bool half_rows = p.groups == 12;   // a workgroup's three points are half of row xrow: its Z is a partial x fold, slot = grp
    for (int sc = 0; sc < nsuper; ++sc) {
        const bool more = sc + D < nsuper;
        long long ts0 = 0, ts1 = 0, ts2 = 0;
        if constexpr (TRACE) ts0 = __builtin_readcyclecounter();
        interval(st, more);
        if constexpr (TRACE) ts1 = __builtin_readcyclecounter();
        if (--xi_left == 0) {
            xi_left = per_xi;
            fold_x(xj);
            if (++xj == 6) {
                xj = 0;
                if (p.zout != nullptr)
                    store_z(half_rows ? grp : xrow++);
                else
                    fold_y(xrow++);
            }
        }
        if constexpr (TRACE) ts2 = __builtin_readcyclecounter();
        if (D > 1 && more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NPIECE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (TRACE) {   // diagnostic (tools/wino4_trace.py): cycle stamps of block 0 go to the `resid` buffer
            const long long ts3 = __builtin_readcyclecounter();
            if (blockIdx.x == 0 && lane == 0 && sc < 72) {
                long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(p.resid)) + (wave * 72 + sc) * 4;
                dbg[0] = ts0; dbg[1] = ts1; dbg[2] = ts2; dbg[3] = ts3;
            }
        }
        __syncthreads();
        st = st + 1 == NST ? 0 : st + 1;
    }

    if (p.zout != nullptr) {
        if (half_rows && xj != 0) store_z(grp);   // the first half of a row ends at j = 2: its partial fold has not left yet
        return;
    }
    // ---- epilogue: one output row py (four pixels) per round, staged through LDS, 16-byte row accesses.
    // A thread keeps the same (channel group, px) and walks BM / PER tiles; their output offsets are decoded once
    // (py only adds a row stride) and out-of-range tiles / channels get an offset the buffer descriptor rejects,
    // so the rounds are branch-free.
    constexpr int LDO = BN + 4, C4 = BN / 4, RPP = NTHR / (4 * C4), PER = BM / RPP;
    static_assert(NTHR % (4 * C4) == 0 && BM % RPP == 0, "tile must split evenly over the threads");
    const int Hq = p.H >> 2, Wq = p.W >> 2;
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = p.bias[ntile * BN + wn * NT * 16 + j * 16 + r16];
    const int e_c4 = tid % C4, e_px = (tid / C4) & 3, e_row0 = tid / (4 * C4);
    const int e_n = ntile * BN + e_c4 * 4;
    unsigned e_off[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int m = mbase + e_row0 + k * RPP;
        const int qx = m % Wq, tq = m / Wq;
        const int qy = tq % Hq, bb = tq / Hq;
        const unsigned o = (unsigned)(((bb * p.H + 4 * qy) * p.W + 4 * qx + e_px) * p.Cout + e_n) * 4u;
        e_off[k] = (m < p.Mq && e_n < p.Cout) ? o : 0xFFFFFFF0u;
    }
    const unsigned out_bytes = (unsigned)p.Mq * 16u * (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, (p.resid != nullptr && !TRACE) ? out_bytes : 0u, 0x00020000);
    const float lo = p.act == ACT_RELU ? 0.f : -INFINITY;
    const unsigned row_bytes = (unsigned)p.W * (unsigned)p.Cout * 4u;
    const float* e_src = smem + (e_px * BM + e_row0) * LDO + e_c4 * 4;
    static_for<4>([&](auto pyc) {
        constexpr int py = decltype(pyc)::value;
        if (py) __syncthreads();
        static_for<4>([&](auto pxc) {
            constexpr int px = decltype(pxc)::value;
            static_for<NT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                float* dst = smem + (px * BM + wm * 16 + 4 * g) * LDO + wn * NT * 16 + j * 16 + r16;
                static_for<4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    dst[r * LDO] = Y[py * 4 + px][j][r] + bias[j];
                });
            });
        });
        __syncthreads();
        const unsigned soff = py * row_bytes;
        u32x4 rr[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsr, e_off[k], soff, 0);  // 0 if no residual
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(e_src + k * RPP * LDO);
            const f32x4_t r4 = __builtin_bit_cast(f32x4_t, rr[k]);
            v = v + r4;
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rso, e_off[k], soff, 0);
            if constexpr (DBG == 30) {
                // UPPER BOUND of "the producing epilogue writes the next convolution's V" (VERDICT r03 route (a)): the bytes of V
                // (36 / 16 = 2.25 x the output) leave from here -- 9 extra 16-byte stores per 4 rounds -- without the transform's
                // arithmetic and without the halo exchange; the caller skips the standalone transform.  Wrong results.
                const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc((void*)p.epi_scratch, 0, 3u * out_bytes, 0x00020000);
                constexpr int extra = py == 3 ? 3 : 2;
                static_for<extra>([&](auto jc) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rse, e_off[k], soff + decltype(jc)::value * out_bytes, 0);
                });
            }
        }
    });
}

// y fold of the split form: Y[p][q] = sum_i A^T[p][i] Z[i][q] (+ bias, residual, activation) -> the 4x4 output pixels;
// POOL: followed by the 2x2 average of DownBlock2d (reference modules/util.py:903-921) -> 2x2 pixels of [B,H/2,W/2,Cout]
// HALVES = 2 (groups == 12): row i arrives as two partial x folds, slots 2 i and 2 i + 1, added first (always in that order).
template <bool POOL, bool RESID, int HALVES = 1>
__global__ __launch_bounds__(256) void wino4_output_transform_kernel(const float* __restrict__ Z, const float* __restrict__ bias,
                                                                     const float* __restrict__ resid, int Mq, int Cout, int H,
                                                                     int W, int act, float* __restrict__ out) {
    const int c4n = Cout >> 2;
    const int Hq = H >> 2, Wq = W >> 2;
    const size_t total = (size_t)Mq * c4n;
    const size_t plane = (size_t)Mq * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const int m = (int)(idx / c4n);
        const int qx = m % Wq, tq = m / Wq;
        const int qy = tq % Hq, b = tq / Hq;
        const f32x4_t* z = reinterpret_cast<const f32x4_t*>(Z) + idx;
        const f32x4_t bs = reinterpret_cast<const f32x4_t*>(bias)[c4];
        const float lo = act == ACT_RELU ? 0.f : -INFINITY;
        f32x4_t colsum[2];   // POOL: the two pooled rows of the current column pair
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4_t zi[6];
            if constexpr (HALVES == 2) {
                f32x4_t za[6], zb[6];      // all twelve loads in flight before the first add
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    za[i] = z[(size_t)((2 * i) * 4 + q) * plane];
                    zb[i] = z[(size_t)((2 * i + 1) * 4 + q) * plane];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) zi[i] = za[i] + zb[i];
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) zi[i] = z[(size_t)(i * 4 + q) * plane];
            }
            const f32x4_t s12 = zi[1] + zi[2], d12 = zi[1] - zi[2], s34 = zi[3] + zi[4], d34 = zi[3] - zi[4];
            f32x4_t y[4];
            y[0] = zi[0] + s12 + s34;
            y[1] = d12 + 2.f * d34;
            y[2] = s12 + 4.f * s34;
            y[3] = d12 + 8.f * d34 + zi[5];
            f32x4_t r4[4];   // residual of this column's four pixels: loaded beside the Z planes, not behind a branch per pixel
            if constexpr (RESID) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
                    r4[pp] = reinterpret_cast<const f32x4_t*>(resid)[(((size_t)(b * H + 4 * qy + pp) * W + 4 * qx + q) * Cout) / 4 + c4];
            }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                f32x4_t v = y[pp] + bs;
                if constexpr (!POOL) {
                    const size_t o = (((size_t)(b * H + 4 * qy + pp) * W + 4 * qx + q) * Cout) / 4 + c4;
                    if constexpr (RESID) v = v + r4[pp];
                    v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
                    reinterpret_cast<f32x4_t*>(out)[o] = v;
                } else {
                    v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
                    y[pp] = v;
                }
            }
            if constexpr (POOL) {
                if ((q & 1) == 0) {
                    colsum[0] = y[0] + y[1];
                    colsum[1] = y[2] + y[3];
                } else {
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const f32x4_t v = 0.25f * (colsum[py] + (y[2 * py] + y[2 * py + 1]));
                        const size_t o = (((size_t)(b * (H >> 1) + 2 * qy + py) * (W >> 1) + 2 * qx + (q >> 1)) * Cout) / 4 + c4;
                        reinterpret_cast<f32x4_t*>(out)[o] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// U[xi][o][c] = (G g G^T)[xi]; w: [Cout][Cin][3][3] (BN folded).  Packed [ntiles][36*Cin/32][BN][32],
// chunk = xi*(Cin/32) + cc, 16-byte slots XOR-swizzled like conv_mfma_dma.
void wino4_pack_host(const float* w, int Cout, int Cin, int BN, float* dst) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int BK = CONV_BK, cch = Cin / BK;
    const int ntiles = (Cout + BN - 1) / BN;
    const size_t total = (size_t)ntiles * 36 * cch * BN * BK;
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / BN, nl = o % BN;
        for (int c = 0; c < Cin; ++c) {
            const float* gw = w + ((size_t)o * Cin + c) * 9;
            double tmp[6][3];
            for (int i = 0; i < 6; ++i)
                for (int k = 0; k < 3; ++k)
                    tmp[i][k] = G[i][0] * gw[0 * 3 + k] + G[i][1] * gw[1 * 3 + k] + G[i][2] * gw[2 * 3 + k];
            const int cc = c / BK, kl = c % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    const int xi = i * 6 + j;
                    dst[(((size_t)nt * 36 * cch + (size_t)xi * cch + cc) * BN + nl) * BK + kk] = (float)u;
                }
        }
    }
}

size_t wino4_packed_elems(int Cout, int Cin, int BN) {
    return (size_t)((Cout + BN - 1) / BN) * 36 * (Cin / CONV_BK) * BN * CONV_BK;
}

hipError_t wino4_transform_launch(const float* x, const float* s, const float* t, int B, int H, int W, int C, float* V,
                                  hipStream_t stream) {
    if ((H & 3) || (W & 3) || (C & 3)) return hipErrorInvalidValue;
    static const int vec_env = [] {   // EAMM_WINO4_TR_VEC = 1 | 2 | 4 forces the channels per thread; unset: by launch size
        const int v = (int)knob_int("EAMM_WINO4_TR_VEC", 0);
        return (v == 1 || v == 2 || v == 4) ? v : 0;
    }();
    // (round 5: a PERSISTENT double-buffered form -- two register images of two channels, the next item's loads in flight while
    //  this one is transformed and stored -- was built, bit-identical, and measured 0 - 10 % SLOWER per launch in the pipeline
    //  at every grid size: profiles/r05_experiments.txt section 3, git 9500c6c)
    // one 256x256 frame is 64 workgroups of four-channel threads on 256 CUs: one channel per thread fills the chip and shortens
    // each thread's 36-load / 36-store chain (12.0 -> 10.6 us per launch, one-frame call 1.069 -> 1.052 ms)
    const int vec = vec_env ? vec_env : ((size_t)B * (H / 4) * (W / 4) * (C / 4) <= 64 * 256 ? 1 : 4);
    const size_t total = (size_t)B * (H / 4) * (W / 4) * (C / vec);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
    static const int xcd = (int)knob_int("EAMM_WINO4_TR_XCD", 3);   // tuning aid: bit 0 = XCD-aware workgroup order, bit 1 = non-temporal V stores
    const int remap = ((size_t)blocks * 256 >= total ? (xcd & 1) : 0) | ((36.0 * total * 16 < 4.0e9 && vec == 4) ? (xcd & 2) : 0);
    if (vec == 1)
        hipLaunchKernelGGL(wino4_input_transform_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, s, t, B, H, W, C, V, remap);
    else if (vec == 2)
        hipLaunchKernelGGL(wino4_input_transform_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, s, t, B, H, W, C, V, remap);
    else
        hipLaunchKernelGGL(wino4_input_transform_kernel<4>, dim3(blocks), dim3(256), 0, stream, x, s, t, B, H, W, C, V, remap);
    return hipGetLastError();
}

template <int SUB, int NST, int PE, int DBG = 0, int WM = 4>
static hipError_t wino4_launch_variant(const Wino4Args& a, hipStream_t stream) {
    constexpr int NT = 2, WN = 2;
    constexpr int BM = WM * 16, BN = WN * NT * 16;
    constexpr size_t lds_loop = sizeof(float) * NST * SUB * (BM + BN) * CONV_BK;
    constexpr size_t lds_epi = sizeof(float) * 4 * BM * (BN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = wino4_gemm_kernel<NT, WM, WN, SUB, NST, PE, DBG>;
    static lds_once_mask configured{0};  // per-device bit mask
    if (hipError_t e = ensure_dynamic_lds(kern, lds, &configured); e != hipSuccess) return e;
    note_mfma_flops(2.0 * 36.0 * a.mtiles * BM * (double)a.ntiles * BN * a.C);
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles * a.groups), dim3(WM * WN * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t wino4_gemm_launch(const WinoLayer& L, const float* V, int B, int H, int W, int act, const float* resid,
                             float* out, hipStream_t stream, int variant, int groups, float* zbuf, int pool) {
    constexpr int BM = 64, BN = 64;
    if (pool && (groups == 1 || resid != nullptr)) return hipErrorInvalidValue;   // the pooled epilogue lives in the output-transform kernel
    if (L.tile != 4 || L.BN != BN || (L.Cout & 3) || L.Cin % (2 * CONV_BK) || (H & 3) || (W & 3))
        return hipErrorInvalidValue;
    Wino4Args a{};
    a.V = V;
    a.U = L.u;
    a.bias = L.bias;
    a.Mq = B * (H / 4) * (W / 4);
    a.C = L.Cin;
    a.Cout = L.Cout;
    a.H = H;
    a.W = W;
    a.mtiles = (a.Mq + BM - 1) / BM;
    a.ntiles = L.ntiles;
    // very few tiles (one 256x256 frame: 4 x 4 x 6 = 96 workgroups): 32-tile blocks of four waves double the grid
    static const int narrow_max = (int)knob_int("EAMM_WINO4_NARROW_MAX_BLOCKS", 128);
    const bool narrow = groups == 6 && a.mtiles * a.ntiles * groups <= narrow_max && L.Cin % (4 * CONV_BK) == 0;
    if (narrow) a.mtiles = (a.Mq + 31) / 32;
    a.act = act;
    a.resid = resid;
    a.out = out;
    if (groups != 1 && groups != 2 && groups != 3 && groups != 6 && groups != 12) return hipErrorInvalidValue;
    if (groups == 12 && (pool || (L.Cin / CONV_BK) * 3 % 4 != 0)) return hipErrorInvalidValue;   // half rows: three points x Cin / 32 chunks, four per interval
    if (groups > 1 && zbuf == nullptr) return hipErrorInvalidValue;
    a.groups = groups;
    a.zout = groups > 1 ? zbuf : nullptr;
#ifdef EAMM_EXPERIMENTS
    a.epi_scratch = (variant == 50 && groups == 1) ? zbuf : nullptr;
    if (variant == 50 && a.epi_scratch == nullptr) variant = 3;
#else
    // variants 10 / 16 / 17 / 50 are timing experiments that compute WRONG results (no DMA in the loop, interval traces, V-sized
    // extra output): they exist only in a -DEAMM_EXPERIMENTS build (make EXPERIMENTS=1); the product library refuses them
    if (variant == 10 || variant == 16 || variant == 17 || variant == 50) return hipErrorInvalidValue;
#endif
    const size_t vb = (size_t)36 * a.Mq * a.C * sizeof(float), ub = wino4_packed_elems(L.Cout, L.Cin, BN) * sizeof(float);
    if (vb >= 0xFFFFF000ull || ub >= 0xFFFFF000ull) return hipErrorInvalidValue;
    a.v_bytes = (unsigned)vb;
    a.u_bytes = (unsigned)ub;
    const bool sub4 = L.Cin % (4 * CONV_BK) == 0;
    hipError_t e = hipErrorInvalidValue;
    // (round 6: a three-stage ring for the narrow launches -- two intervals of operands in flight -- measured 21.8 us per launch against 21.5:
    //  the one-frame launch is not DMA-latency bound; its 12 intervals cost what the steady-state kernel's cost, 1.2 us each, + ~7 us fixed)
    if (narrow) variant = 30;
    switch (variant) {   // (chunks per barrier, ring depth, MFMAs per DMA piece)
        case 30: e = wino4_launch_variant<4, 2, 4, 0, 2>(a, stream); break;
#ifdef EAMM_EXPERIMENTS
        case 50: e = sub4 ? wino4_launch_variant<4, 2, 8, 30>(a, stream) : wino4_launch_variant<2, 2, 4>(a, stream); break;
        case 10: e = wino4_launch_variant<4, 2, 4, 1>(a, stream); break;
        case 16: e = wino4_launch_variant<4, 2, 4, 8>(a, stream); break;
        case 17: e = wino4_launch_variant<4, 2, 4, 9>(a, stream); break;
#endif
        case 0: e = sub4 ? wino4_launch_variant<4, 2, 4>(a, stream) : wino4_launch_variant<2, 2, 4>(a, stream); break;
        case 1: e = wino4_launch_variant<2, 2, 4>(a, stream); break;
        case 2: e = wino4_launch_variant<2, 3, 4>(a, stream); break;
        case 3: e = sub4 ? wino4_launch_variant<4, 2, 8>(a, stream) : wino4_launch_variant<2, 2, 4>(a, stream); break;
        case 4: e = wino4_launch_variant<2, 4, 4>(a, stream); break;
        case 6: e = sub4 ? wino4_launch_variant<4, 2, 8, 40>(a, stream) : wino4_launch_variant<2, 2, 4>(a, stream); break;   // variant 3 with non-temporal V loads (the default since round 5: 4034 -> 4048 frames/s)
        case 5: e = sub4 ? wino4_launch_variant<4, 2, 4, 20>(a, stream) : wino4_launch_variant<2, 2, 4, 20>(a, stream); break;
        default: break;
    }
    if (e != hipSuccess || groups == 1) return e;
    const size_t total = (size_t)a.Mq * (L.Cout / 4);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
    if (groups == 12) {
        if (resid != nullptr)
            hipLaunchKernelGGL((wino4_output_transform_kernel<false, true, 2>), dim3(blocks), dim3(256), 0, stream, zbuf, L.bias, resid,
                               a.Mq, L.Cout, H, W, act, out);
        else
            hipLaunchKernelGGL((wino4_output_transform_kernel<false, false, 2>), dim3(blocks), dim3(256), 0, stream, zbuf, L.bias, resid,
                               a.Mq, L.Cout, H, W, act, out);
    } else if (pool)
        hipLaunchKernelGGL((wino4_output_transform_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, zbuf, L.bias, resid,
                           a.Mq, L.Cout, H, W, act, out);
    else if (resid != nullptr)
        hipLaunchKernelGGL((wino4_output_transform_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, zbuf, L.bias, resid,
                           a.Mq, L.Cout, H, W, act, out);
    else
        hipLaunchKernelGGL((wino4_output_transform_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, zbuf, L.bias, resid,
                           a.Mq, L.Cout, H, W, act, out);
    return hipGetLastError();
}

}  // namespace eamm
