// Winograd F(2x2, 3x3) form of the 3x3 / pad 1 convolution for the bottleneck ResBlock2d stack
// (reference modules/util.py:858-880, generator.py:89): 2.25x fewer multiply-adds than the direct form, still
// plain fp32 arithmetic (v_mfma_f32_32x32x2_f32 + fp32 adds); only the summation order / rounding differs.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch (stride 2), Y: 2x2 outputs
//
// The path's pixel enumeration is already 2x2-quad ordered (m = 4*quad + 2*jy + jx), so Winograd tile t IS
// quad t and its four outputs are the quad's four pixels.  Two kernels per convolution:
//
//  1. wino_input_transform_kernel (HBM-bound): V[xi][tile][c] = (B^T d B)[xi] for the 16 transform points,
//     optionally applying the ResBlock pre-activation relu(x*s + t) (eval BatchNorm folded to scale/shift,
//     util.py:873-874) to in-range pixels first -- so the producer no longer has to emit a pre-activated copy.
//     One thread per (tile, 4 channels): a wave covers 256 contiguous channels, every access is 1 KiB.
//  2. wino_gemm_kernel (MFMA-bound): for xi = 0..15: M_xi[tile, o] = sum_c V_xi[tile, c] U_xi[c, o] as an
//     LDS-DMA GEMM exactly like conv_mfma_dma.hip (same swizzled LDS image, DMA pieces interleaved into the
//     MFMA stream), K walked xi-outer / channel-chunk-inner with two 32-channel chunks per barrier.  When a
//     xi is finished its accumulator is folded into the four output accumulators Y[p] += coef[p][xi] * M_xi
//     (coef in {0,+1,-1}: A^T M A), so the transformed products never leave the registers.  The epilogue
//     stages Y through LDS and writes full 16-byte rows: bias (+ folded BatchNorm), residual, activation.
//
// U = G g G^T is computed in double on the host from the BatchNorm-folded weights and rounded once.
#include "conv_common.h"

#include <algorithm>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---------------------------------------------------------------------------------------------------------
// input transform
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ s,
                                                                   const float* __restrict__ t, int B, int H, int W,
                                                                   int C, float* __restrict__ V) {
    const int c4n = C >> 2;
    const int Hq = H >> 1, Wq = W >> 1;
    const size_t Mq = (size_t)B * Hq * Wq;
    const size_t total = Mq * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t q = idx / c4n;
        const int qx = (int)(q % Wq);
        const int qy = (int)((q / Wq) % Hq);
        const int b = (int)(q / ((size_t)Wq * Hq));
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s != nullptr) {
            sc = reinterpret_cast<const float4*>(s)[c4];
            sh = reinterpret_cast<const float4*>(t)[c4];
        }
        const float4* img = reinterpret_cast<const float4*>(x) + (size_t)b * H * W * c4n + c4;
        float4 d[4][4];
        // all 16 loads issued before the first use (clamped addresses, out-of-range pixels zeroed afterwards): a bounds
        // branch around each load made the compiler wait for every load before issuing the next
        int yo[4], xo[4];
        bool oky[4], okx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = 2 * qy - 1 + i, xx = 2 * qx - 1 + i;
            oky[i] = (unsigned)yy < (unsigned)H;
            okx[i] = (unsigned)xx < (unsigned)W;
            yo[i] = min(max(yy, 0), H - 1) * W;
            xo[i] = min(max(xx, 0), W - 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = img[(size_t)(yo[i] + xo[j]) * c4n];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = d[i][j];
                if (s != nullptr) {  // zero padding applies to the ACTIVATED tensor: only in-range pixels
                    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
                    v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
                    v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                }
                d[i][j] = (oky[i] && okx[j]) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        // B^T d: rows (d0-d2, d1+d2, d2-d1, d1-d3), then the same along columns
        float4 r[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = f4_sub(d[0][j], d[2][j]);
            r[1][j] = f4_add(d[1][j], d[2][j]);
            r[2][j] = f4_sub(d[2][j], d[1][j]);
            r[3][j] = f4_sub(d[1][j], d[3][j]);
        }
        float4* out = reinterpret_cast<float4*>(V) + q * c4n + c4;
        const size_t plane = Mq * c4n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[(size_t)(i * 4 + 0) * plane] = f4_sub(r[i][0], r[i][2]);
            out[(size_t)(i * 4 + 1) * plane] = f4_add(r[i][1], r[i][2]);
            out[(size_t)(i * 4 + 2) * plane] = f4_sub(r[i][2], r[i][1]);
            out[(size_t)(i * 4 + 3) * plane] = f4_sub(r[i][1], r[i][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM over the 16 transform points + in-register output transform
// ---------------------------------------------------------------------------------------------------------
struct WinoArgs {
    const float* V;        // [16][Mq][C]
    const float* U;        // packed [ntiles][16 * C/32][BN][32] (swizzled), xi outer / channel chunk inner
    const float* bias;     // [ntiles*BN]
    unsigned v_bytes, u_bytes;
    int Mq, C, Cout;       // tiles (= pixels / 4), input channels, output channels
    int H, W;              // output image size (2*Hq, 2*Wq)
    int mtiles, ntiles;
    int act;
    const float* resid;    // NHWC [B,H,W,Cout]
    float* out;            // NHWC [B,H,W,Cout]
};

// SUB: 32-channel chunks per barrier interval; NST: LDS ring depth in intervals (prefetch distance NST-1, the
// interval-end wait is a counted vmcnt that leaves the younger intervals' DMA in flight); PE: one DMA piece
// every PE MFMAs from the start of an interval.
template <int MT, int NT, int WM, int WN, int SUB, int NST, int PE>
__global__ __launch_bounds__(WM* WN * 64) void wino_gemm_kernel(const WinoArgs p) {
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, BK = CONV_BK;
    constexpr int A_STAGE = SUB * BM * BK, B_STAGE = SUB * BN * BK;  // floats per stage
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;       // DMA instructions per wave per chunk
    constexpr int NPIECE = SUB * (A_INSTR + B_INSTR);
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NST][A_STAGE] [NST][B_STAGE]
    float* const As = smem;
    float* const Bs = smem + NST * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    const int mtile = L / p.ntiles;
    const int mbase = mtile * BM;
    const int cchunks = p.C / BK;             // channel chunks per transform point
    const int nsuper = 16 * cchunks / SUB;    // barrier intervals

    // A loader: DMA instruction j of this wave fills rows (wave*A_INSTR + j)*8 .. +8 of a chunk
    unsigned arow_off[A_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        const int m = mbase + row;
        arow_off[j] = m < p.Mq ? (unsigned)(m * p.C + slot * 4) * 4u : 0xFFFFFFF0u;
    }
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);
    const unsigned plane_bytes = (unsigned)p.Mq * (unsigned)p.C * 4u;

    // piece k of super-chunk sc (chunks 2sc, 2sc+1) into stage st
    int n_sc = 0, n_st = 0;
    auto dma_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int sub = k / (A_INSTR + B_INSTR), r = k % (A_INSTR + B_INSTR);
        const int ci = n_sc * SUB + sub;              // chunk index: xi * cchunks + cc
        const int xi = ci / cchunks, cc = ci - xi * cchunks;
        if constexpr (r < A_INSTR) {
            const unsigned off = arow_off[r] + (unsigned)xi * plane_bytes + (unsigned)(cc * BK * 4);
            // an out-of-range row keeps a huge offset (0xFFFFFFF0 + small wraps below 2^32 only for xi = cc = 0;
            // clamp explicitly so the M tail always misses the descriptor range)
            const unsigned o2 = arow_off[r] == 0xFFFFFFF0u ? 0xFFFFFFF0u : off;
            float* dst = As + n_st * A_STAGE + sub * (BM * BK) + (wave * A_INSTR + r) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_ptr_t)dst, 16, o2, 0, 0, 0);
        } else {
            constexpr int j = r - A_INSTR;
            const unsigned off =
                (unsigned)(((ntile * 16 * cchunks + ci) * BN + (wave * B_INSTR + j) * 8) * BK + lane * 4) * 4u;
            float* dst = Bs + n_st * B_STAGE + sub * (BN * BK) + (wave * B_INSTR + j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };

    f32x16 acc[MT][NT];       // M_xi of the transform point in flight
    f32x16 Y[4][MT][NT];      // the four outputs of each tile (quad order: p = 2*py + px)
    static_for<MT>([&](auto ic) {
        static_for<NT>([&](auto jc) {
            constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                acc[i][j][r] = 0.f;
                static_for<4>([&](auto pc) { Y[decltype(pc)::value][i][j][r] = 0.f; });
            });
        });
    });

    const int sw = (l31 >> 1) & 7;
    constexpr int MF = 4 * MT * NT;              // MFMAs per 8-wide K sub-step
    constexpr int TOTAL_MF = SUB * (BK / 8) * MF;  // MFMAs per barrier interval
    constexpr int PIECE_EVERY = PE;
    static_assert(PIECE_EVERY >= 1 && NPIECE * PIECE_EVERY <= TOTAL_MF, "DMA pieces must fit in the interval");
    auto compute = [&](int st, bool more) {
        f32x4 a[2][MT], b[2][NT];
        auto fetch = [&](int step, int buf) {  // step = sub * 4 + s
            const int sub = step >> 2, s = step & 3;
            const int slot = ((2 * s + half) ^ sw) << 2;
            const float* a_base = As + st * A_STAGE + sub * (BM * BK) + (wm * MT * 32 + l31) * BK + slot;
            const float* b_base = Bs + st * B_STAGE + sub * (BN * BK) + (wn * NT * 32 + l31) * BK + slot;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[buf][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[buf][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK);
        };
        fetch(0, 0);
        static_for<SUB*(BK / 8)>([&](auto sc) {
            constexpr int step = decltype(sc)::value;
            if constexpr (step + 1 < SUB * (BK / 8)) fetch(step + 1, (step + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<MF>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / (MT * NT), i = (q / NT) % MT, j = q % NT;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][i][t], b[step & 1][j][t], acc[i][j], 0, 0, 0);
                constexpr int g = step * MF + q;
                if constexpr (g % PIECE_EVERY == PIECE_EVERY - 1 && g / PIECE_EVERY < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) dma_piece(std::integral_constant<int, g / PIECE_EVERY>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };

    // A^T = [[1,1,1,0],[0,1,-1,-1]]: Y[py][px] += AT[py][i] * AT[px][j] * M[i][j], xi = 4i + j
    auto fold = [&](int xi) {
        const int i = xi >> 2, j = xi & 3;
        const float r0 = i < 3 ? 1.f : 0.f, r1 = i == 0 ? 0.f : (i == 1 ? 1.f : -1.f);
        const float c0 = j < 3 ? 1.f : 0.f, c1 = j == 0 ? 0.f : (j == 1 ? 1.f : -1.f);
        const float k00 = r0 * c0, k01 = r0 * c1, k10 = r1 * c0, k11 = r1 * c1;
        static_for<MT>([&](auto ic) {
            static_for<NT>([&](auto jc) {
                constexpr int ii = decltype(ic)::value, jj = decltype(jc)::value;
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float m = acc[ii][jj][r];
                    Y[0][ii][jj][r] = fmaf(k00, m, Y[0][ii][jj][r]);
                    Y[1][ii][jj][r] = fmaf(k01, m, Y[1][ii][jj][r]);
                    Y[2][ii][jj][r] = fmaf(k10, m, Y[2][ii][jj][r]);
                    Y[3][ii][jj][r] = fmaf(k11, m, Y[3][ii][jj][r]);
                    acc[ii][jj][r] = 0.f;
                });
            });
        });
    };

    // ---- main loop: ring of NST stages, DMA runs D = NST-1 intervals ahead of the MFMAs
    constexpr int D = NST - 1;
    const int per_xi = cchunks / SUB;  // barrier intervals per transform point
    for (int k = 0; k < D && k < nsuper; ++k) {
        n_sc = k;
        n_st = k;
        static_for<NPIECE>([&](auto kc) { dma_piece(kc); });
    }
    if (D > 1 && nsuper >= D) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NPIECE) : "memory");  // interval 0 has landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int st = 0, st_next = D % NST;
    for (int sc = 0; sc < nsuper; ++sc) {
        const bool more = sc + D < nsuper;
        n_sc = sc + D;
        n_st = st_next;
        compute(st, more);
        if ((sc + 1) % per_xi == 0) fold(sc / per_xi);
        // interval sc+1 must have landed; the D-1 younger intervals may stay in flight (while they exist)
        if (D > 1 && more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NPIECE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        st = st + 1 == NST ? 0 : st + 1;
        st_next = st_next + 1 == NST ? 0 : st_next + 1;
    }

    // ---- epilogue: 4 outputs per tile, staged through LDS, 16-byte row accesses
    constexpr int R = WM * 32, LDO = BN + 4, C4 = BN / 4, PER = R * C4 / NTHR;
    static_assert((R * C4) % NTHR == 0, "tile must split evenly over the threads");
    const int Hq = p.H >> 1, Wq = p.W >> 1;
    static_for<4>([&](auto pc) {
        constexpr int pp = decltype(pc)::value;
        static_for<MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __syncthreads();
            static_for<NT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int col = wn * NT * 32 + j * 32 + l31;
                const float bias = p.bias[ntile * BN + col];
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    smem[row * LDO + col] = Y[pp][i][j][r] + bias;
                });
            });
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int idx = tid + k * NTHR;
                const int row = idx / C4, c4 = idx - row * C4;
                const int m = mbase + (row >> 5) * (MT * 32) + i * 32 + (row & 31);  // tile = quad index
                const int n = ntile * BN + c4 * 4;
                if (m < p.Mq && n < p.Cout) {
                    float4 v = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                    const int qx = m % Wq, tq = m / Wq;
                    const int qy = tq % Hq, b = tq / Hq;
                    const int y = 2 * qy + (pp >> 1), x = 2 * qx + (pp & 1);
                    const size_t o = ((size_t)(b * p.H + y) * p.W + x) * p.Cout + n;
                    if (p.resid != nullptr) {
                        const float4 rr = *reinterpret_cast<const float4*>(p.resid + o);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                    v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                    v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                    *reinterpret_cast<float4*>(p.out + o) = v;
                }
            }
        });
    });
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// U[xi][o][c] = (G g G^T)[xi], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; w: [Cout][Cin][3][3] (BN folded).
// Packed [ntiles][16*Cin/32][BN][32], chunk = xi*(Cin/32) + cc, 16-byte slots XOR-swizzled like conv_mfma_dma.
void wino_pack_host(const float* w, int Cout, int Cin, int BN, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int BK = CONV_BK, cch = Cin / BK;
    const int ntiles = (Cout + BN - 1) / BN;
    const size_t total = (size_t)ntiles * 16 * cch * BN * BK;
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / BN, nl = o % BN;
        for (int c = 0; c < Cin; ++c) {
            const float* g = w + ((size_t)o * Cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k) tmp[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
            const int cc = c / BK, kl = c % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    const int xi = i * 4 + j;
                    dst[(((size_t)nt * 16 * cch + (size_t)xi * cch + cc) * BN + nl) * BK + kk] = (float)u;
                }
        }
    }
}

size_t wino_packed_elems(int Cout, int Cin, int BN) {
    return (size_t)((Cout + BN - 1) / BN) * 16 * (Cin / CONV_BK) * BN * CONV_BK;
}

hipError_t wino_transform_launch(const float* x, const float* s, const float* t, int B, int H, int W, int C, float* V,
                                 hipStream_t stream) {
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
    hipLaunchKernelGGL(wino_input_transform_kernel, dim3(blocks), dim3(256), 0, stream, x, s, t, B, H, W, C, V);
    return hipGetLastError();
}

template <int SUB, int NST, int PE>
static hipError_t wino_launch_variant(const WinoArgs& a, hipStream_t stream) {
    constexpr int MT = 1, NT = 2, WM = 4, WN = 2;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t lds_loop = sizeof(float) * NST * SUB * (BM + BN) * CONV_BK;
    constexpr size_t lds_epi = sizeof(float) * (WM * 32) * (BN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = wino_gemm_kernel<MT, NT, WM, WN, SUB, NST, PE>;
    static lds_once_mask configured{0};  // per-device bit mask
    if (hipError_t e = ensure_dynamic_lds(kern, lds, &configured); e != hipSuccess) return e;
    note_mfma_flops(2.0 * 16.0 * a.mtiles * BM * (double)a.ntiles * BN * a.C);
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(WM * WN * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t wino_gemm_launch(const WinoLayer& L, const float* V, int B, int H, int W, int act, const float* resid,
                            float* out, hipStream_t stream, int variant) {
    constexpr int BM = 128, BN = 128;
    if (L.BN != BN || (L.Cout & 3) || L.Cin % (2 * CONV_BK) || (H & 1) || (W & 1)) return hipErrorInvalidValue;
    WinoArgs a{};
    a.V = V;
    a.U = L.u;
    a.bias = L.bias;
    a.Mq = B * (H / 2) * (W / 2);
    a.C = L.Cin;
    a.Cout = L.Cout;
    a.H = H;
    a.W = W;
    a.mtiles = (a.Mq + BM - 1) / BM;
    a.ntiles = L.ntiles;
    a.act = act;
    a.resid = resid;
    a.out = out;
    const size_t vb = (size_t)16 * a.Mq * a.C * sizeof(float), ub = wino_packed_elems(L.Cout, L.Cin, BN) * sizeof(float);
    if (vb >= 0xFFFFFFF0ull || ub >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.v_bytes = (unsigned)vb;
    a.u_bytes = (unsigned)ub;
    switch (variant) {   // (chunks per barrier, ring depth, MFMAs per DMA piece) -- profiles/r01_wino_variants.txt
        case 0: return wino_launch_variant<2, 2, 4>(a, stream);
        case 1: return wino_launch_variant<1, 4, 4>(a, stream);
        case 2: return wino_launch_variant<1, 5, 4>(a, stream);
        case 3: return wino_launch_variant<2, 2, 2>(a, stream);
        case 4: return wino_launch_variant<2, 2, 1>(a, stream);
        case 5: return wino_launch_variant<1, 3, 4>(a, stream);
        case 6: return wino_launch_variant<1, 4, 2>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace eamm
