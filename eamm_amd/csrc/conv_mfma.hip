// 3x3 / 7x7 convolution on NHWC fp32 activations as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces every F.conv2d + batch_norm + relu (+ avg_pool2d / nearest-upsample / cat / residual add)
// group of the reference's DownBlock2d / UpBlock2d / ResBlock2d / SameBlock2d (modules/util.py:858-938)
// with ONE kernel: eval-mode BatchNorm is folded into the weights at load time, the activation,
// 2x2 average pool, residual add and the next block's pre-activation are epilogues, the nearest
// x2 up-sampling and the channel concatenation of the hourglass decoder are address arithmetic in
// the operand loader (nothing is materialised).
//
// GEMM view:  M = B*H*W output pixels, N = Cout, K = taps * Cin, exact fp32 via
// v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain; gfx950 has no TF32/xf32).
//   * block tile BM x BN = (WM*MT*32) x (WN*NT*32), 4 waves, each wave MT x NT MFMA tiles;
//   * K is walked in chunks of 32 input channels of one filter tap; channel-chunk outer, tap inner,
//     so the 9 (49) shifted re-reads of an activation line hit L2 back to back;
//   * both operands are staged global -> VGPR -> LDS (double buffered, one barrier per chunk) as
//     [row][32+4] so that each lane fetches FOUR consecutive k with one ds_read_b128: lanes 0-31
//     take k = 8s+{0..3}, lanes 32-63 k = 8s+{4..7}; A and B use the same permutation of K, which
//     the sum does not care about;
//   * M is enumerated in 2x2-quad order (m = 4*quad + 2*jy + jx): the four accumulator registers
//     4g..4g+3 of a lane are then exactly one pooling window, so AvgPool2d(2) is in-register;
//   * zero padding = predicated loads; M/N tails are masked;
//   * small-M layers (the deep hourglass levels) use split-K over gridDim with an fp32 slab
//     reduction kernel that applies the epilogue;
//   * blockIdx is remapped so that consecutive logical tiles (same M tile, all N tiles) share an XCD
//     and therefore an L2.
#include "kernels.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace eamm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "block b runs on XCD b%8" -> contiguous chunk per XCD
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

template <int KS, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int BK = CONV_BK, LDK = CONV_LDK;
    constexpr int T = KS * KS, P = KS / 2;
    constexpr int A_PER = BM / 32, B_PER = BN / 32;
    static_assert(WM * WN == 4, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;  // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    L /= p.ntiles;
    const int mtile = L % p.mtiles;
    const int split = L / p.mtiles;
    const int mbase = mtile * BM;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);

    // ---- operand loader state: this thread stages rows arow+32j, 16 bytes at column acol
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int Wq = p.W >> 1, Hq = p.H >> 1;
    int ry[A_PER], rx[A_PER], rb[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int m = mbase + arow + 32 * j;
        if (m < p.M) {
            const int q = m >> 2, jj = m & 3;
            const int qx = q % Wq, t = q / Wq;
            const int qy = t % Hq, b = t / Hq;
            ry[j] = 2 * qy + (jj >> 1);
            rx[j] = 2 * qx + (jj & 1);
            rb[j] = b * p.Hin * p.Win;
        } else {
            ry[j] = -(1 << 20);  // never inside the image
            rx[j] = 0;
            rb[j] = 0;
        }
    }

    // Buffer descriptors (wave-uniform, built from kernel arguments): an out-of-range offset makes the
    // hardware return zeros, so zero padding and the M tail need no branch -- the loads stay in flight
    // under the MFMAs and are first waited for at the LDS store.
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    u32x4 av[A_PER], bv[B_PER];
    auto load_chunk = [&](int ci) {
        const int cc = ci / T, tap = ci - cc * T;
        const int dy = tap / KS - P, dx = tap % KS - P;
        const int c0 = cc * BK;
        const bool first = c0 < p.C0;
        const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
        const int C = first ? p.C0 : p.C1;
        const int coff = (first ? c0 : c0 - p.C0) + acol;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            int yy = ry[j] + dy, xx = rx[j] + dx;
            const bool ok = ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
            if (p.up) {
                yy >>= 1;
                xx >>= 1;
            }
            const unsigned off = ok ? (unsigned)((rb[j] + yy * p.Win + xx) * C + coff) * 4u : OOB;
            av[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
        const unsigned woff = (unsigned)((ntile * p.nchunks + ci) * (BN * BK) + tid * 4) * 4u;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) bv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + j * 4096u, 0, 0);
    };
    auto store_chunk = [&](int st) {
        float* a = As + st * BM * LDK + arow * LDK + acol;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) *reinterpret_cast<u32x4*>(a + j * 32 * LDK) = av[j];
        float* b = Bs + st * BN * LDK + arow * LDK + acol;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) *reinterpret_cast<u32x4*>(b + j * 32 * LDK) = bv[j];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int st) {
        const float* a_base = As + st * BM * LDK + (wm * MT * 32 + l31) * LDK + half * 4;
        const float* b_base = Bs + st * BN * LDK + (wn * NT * 32 + l31) * LDK + half * 4;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDK + s * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDK + s * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: LDS double buffer, one barrier per K chunk
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        int st = 0;
        for (int ci = c_begin; ci < c_end; ++ci) {
            const bool more = ci + 1 < c_end;
            if (more) load_chunk(ci + 1);  // global loads in flight under the MFMAs
            compute(st);
            if (more) store_chunk(st ^ 1);
            __syncthreads();
            st ^= 1;
        }
    }

    // ---- epilogue
    if (p.partial != nullptr) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = ntile * BN + wn * NT * 32 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    p.partial[((size_t)split * p.Mpad + m) * p.Npad + n] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ntile * BN + wn * NT * 32 + j * 32 + l31;
            const bool nok = n < p.Cout;
            const float bias = p.bias[n];
            float s2 = 0.f, t2 = 0.f;
            if (p.out2 != nullptr && nok) {
                s2 = p.s2[n];
                t2 = p.t2[n];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m0 = mbase + wm * MT * 32 + i * 32 + 8 * g + 4 * half;  // first pixel of a quad
                if (m0 >= p.M || !nok) continue;
                const int q = m0 >> 2;
                if (p.pool) {
                    float v = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v += apply_act(acc[i][j][4 * g + e] + bias, p.act);
                    p.out[(size_t)q * p.Cout + n] = 0.25f * v;
                } else {
                    const int qx = q % Wq, t = q / Wq;
                    const int qy = t % Hq, b = t / Hq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int y = 2 * qy + (e >> 1), x = 2 * qx + (e & 1);
                        const size_t pix = (size_t)(b * p.H + y) * p.W + x;
                        float v = acc[i][j][4 * g + e] + bias;
                        if (p.resid != nullptr) v += p.resid[pix * p.Cout + n];
                        v = apply_act(v, p.act);
                        if (p.nchw)
                            p.out[((size_t)(b * p.Cout + n) * p.H + y) * p.W + x] = v;
                        else
                            p.out[pix * p.Cout + n] = v;
                        if (p.out2 != nullptr) p.out2[pix * p.Cout + n] = fmaxf(fmaf(v, s2, t2), 0.f);
                    }
                }
            }
        }
}

// Split-K slab reduction + the same epilogue as the fused path (bias, residual, activation, 2x2
// average pool, NCHW store, next-block pre-activation).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs p, int splits) {
    const int rows = p.pool ? (p.M >> 2) : p.M;
    const size_t total = (size_t)rows * p.Cout;
    const float* __restrict__ partial = p.partial;
    const int Wq = p.W >> 1, Hq = p.H >> 1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % p.Cout);
        const int r = (int)(idx / p.Cout);
        const float b = p.bias[n];
        if (p.pool) {
            float v = 0.f;
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
                for (int sp = 0; sp < splits; ++sp) s += partial[((size_t)sp * p.Mpad + 4 * r + e) * p.Npad + n];
                v += apply_act(s + b, p.act);
            }
            p.out[(size_t)r * p.Cout + n] = 0.25f * v;
        } else {
            float s = 0.f;
            for (int sp = 0; sp < splits; ++sp) s += partial[((size_t)sp * p.Mpad + r) * p.Npad + n];
            const int q = r >> 2, e = r & 3;
            const int qx = q % Wq, t = q / Wq;
            const int qy = t % Hq, bb = t / Hq;
            const int y = 2 * qy + (e >> 1), x = 2 * qx + (e & 1);
            const size_t pix = (size_t)(bb * p.H + y) * p.W + x;
            float v = s + b;
            if (p.resid != nullptr) v += p.resid[pix * p.Cout + n];
            v = apply_act(v, p.act);
            if (p.nchw)
                p.out[((size_t)(bb * p.Cout + n) * p.H + y) * p.W + x] = v;
            else
                p.out[pix * p.Cout + n] = v;
            if (p.out2 != nullptr) p.out2[pix * p.Cout + n] = fmaxf(fmaf(v, p.s2[n], p.t2[n]), 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int conv_tile_n(int Cout) {
    if (Cout > 64) return 128;
    if (Cout > 32) return 64;
    return 32;
}

size_t conv_packed_elems(int ks, int cin_packed, int Cout, int BN) {
    const int ntiles = (Cout + BN - 1) / BN;
    return (size_t)ntiles * BN * ks * ks * cin_packed;
}

void conv_pack_host(const float* w, int Cout, int Cin, int ks, const int* cin_map, int cin_packed, int BN,
                    float* dst) {
    const int T = ks * ks, BK = CONV_BK;
    const int ntiles = (Cout + BN - 1) / BN;
    const int nchunks = T * (cin_packed / BK);
    std::memset(dst, 0, sizeof(float) * conv_packed_elems(ks, cin_packed, Cout, BN));
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ci = 0; ci < nchunks; ++ci) {
            const int cc = ci / T, tap = ci % T;
            float* tile = dst + ((size_t)nt * nchunks + ci) * BN * BK;
            for (int nl = 0; nl < BN; ++nl) {
                const int o = nt * BN + nl;
                if (o >= Cout) continue;
                for (int kl = 0; kl < BK; ++kl) {
                    const int c = cin_map ? cin_map[cc * BK + kl] : cc * BK + kl;
                    if (c < 0 || c >= Cin) continue;
                    tile[nl * BK + kl] = w[((size_t)o * Cin + c) * T + tap];
                }
            }
        }
}

ConvPlan conv_plan(const ConvLayer& L, int M, int force_splits) {
    ConvPlan pl;
    const int BM = 128;
    pl.mtiles = (M + BM - 1) / BM;
    pl.ntiles = L.ntiles;
    const int blocks = pl.mtiles * pl.ntiles;
    int splits = 1;
    if (force_splits > 0) {
        splits = force_splits;
    } else if (blocks < 192) {
        // deep hourglass levels: too few output tiles to fill 256 CUs -> slice K instead
        splits = (256 + blocks - 1) / blocks;
        splits = std::min(splits, std::max(1, L.nchunks / 8));
    }
    splits = std::max(1, std::min(splits, L.nchunks));
    pl.chunks_per_split = (L.nchunks + splits - 1) / splits;
    pl.splits = (L.nchunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
    pl.Mpad = pl.mtiles * BM;
    pl.Npad = pl.ntiles * L.BN;
    pl.partial_elems = pl.splits > 1 ? (size_t)pl.splits * pl.Mpad * pl.Npad : 0;
    return pl;
}

template <int KS, int MT, int NT, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, int blocks, hipStream_t stream) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t lds = sizeof(float) * 2 * (BM + BN) * CONV_LDK;
    auto kern = conv_mfma_kernel<KS, MT, NT, WM, WN>;
    static bool configured = false;  // per instantiation; the attribute is per-device but all devices share the value
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t conv_launch(const ConvLayer& L, const ConvIO& io, hipStream_t stream, int force_splits) {
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in0 = io.in0;
    a.in1 = io.in1;
    a.C0 = L.C0;
    a.C1 = L.C1;
    a.Hin = io.Hin;
    a.Win = io.Win;
    a.up = io.up;
    a.H = io.Hin << io.up;
    a.W = io.Win << io.up;
    a.M = io.B * a.H * a.W;
    a.w = L.w;
    {   // buffer-descriptor ranges (32-bit): every tensor the loader touches must stay below 4 GiB
        const size_t px = (size_t)io.B * io.Hin * io.Win;
        const size_t b0 = px * L.C0 * sizeof(float), b1 = px * L.C1 * sizeof(float);
        const size_t bw = (size_t)L.ntiles * L.nchunks * L.BN * CONV_BK * sizeof(float);
        if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
        a.in0_bytes = (unsigned)b0;
        a.in1_bytes = (unsigned)b1;
        a.w_bytes = (unsigned)bw;
    }
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.nchunks = L.nchunks;
    ConvPlan pl = conv_plan(L, a.M, force_splits);
    while (pl.splits > 1 && pl.partial_elems > io.partial_cap) pl = conv_plan(L, a.M, pl.splits - 1);
    a.mtiles = pl.mtiles;
    a.ntiles = pl.ntiles;
    a.chunks_per_split = pl.chunks_per_split;
    a.Mpad = pl.Mpad;
    a.Npad = pl.Npad;
    a.partial = pl.splits > 1 ? io.partial : nullptr;
    a.act = io.act;
    a.pool = io.pool;
    a.nchw = io.nchw;
    a.resid = io.resid;
    a.out = io.out;
    a.out2 = io.out2;
    a.s2 = io.s2;
    a.t2 = io.t2;
    if ((a.H & 1) || (a.W & 1)) return hipErrorInvalidValue;
    if (pl.splits > 1 && io.partial == nullptr) return hipErrorInvalidValue;
    const int blocks = pl.mtiles * pl.ntiles * pl.splits;
    hipError_t e = hipErrorInvalidValue;
    if (L.ks == 3) {
        if (L.BN == 128) e = launch_cfg<3, 2, 2, 2, 2>(a, blocks, stream);
        else if (L.BN == 64) e = launch_cfg<3, 2, 1, 2, 2>(a, blocks, stream);
        else if (L.BN == 32) e = launch_cfg<3, 1, 1, 4, 1>(a, blocks, stream);
    } else if (L.ks == 7) {
        if (L.BN == 128) e = launch_cfg<7, 2, 2, 2, 2>(a, blocks, stream);
        else if (L.BN == 64) e = launch_cfg<7, 2, 1, 2, 2>(a, blocks, stream);
        else if (L.BN == 32) e = launch_cfg<7, 1, 1, 4, 1>(a, blocks, stream);
    }
    if (e != hipSuccess) return e;
    if (pl.splits > 1) {
        const int rows = io.pool ? (a.M >> 2) : a.M;
        const size_t total = (size_t)rows * L.Cout;
        const int rb = (int)std::min<size_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, stream, a, pl.splits);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace eamm
