// Backward kernels of the path's two operator kinds ("backward of warp/conv kernels"), op level:
//   * warp_features_backward_kernel -- gradient of  out = grid_sample(feat, flow, bilinear, zeros, align_corners=False) * occ
//     (reference modules/generator.py:50-57, 79-84) with respect to the features, the flow and the occlusion map: what
//     autograd derives from F.grid_sample and the multiply.  HBM-bound: dout read once, the four corner lines re-read,
//     d_feat accumulated with float atomics (as ATen's grid_sampler backward does; with one source for all frames the frames'
//     contributions land on the same map).
//   * conv_wgrad_kernel -- weight gradient of a stride-1 "same" KHxKW convolution on NHWC activations
//     (reference modules/util.py:858-938, every Conv2d of the blocks): dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p + (ky,kx)][ci],
//     one fp32-MFMA GEMM per filter tap with M = Cout, N = Cin and K = the pixels of the batch, split over the pixel range;
//     both operands are K-major as they lie in HBM (a pixel's channels are contiguous), so a K step of two pixels is two
//     coalesced rows of each tensor.  conv_bias_grad_kernel: db[co] = sum_p dY[p][co].
//   The DATA gradient of such a convolution is itself a "same" convolution of dY with the transposed, flipped filter, i.e. the
//   forward kernels (eamm_op_conv) on repacked weights; BatchNorm's backward is in batchnorm.hip.
// These are the kernels; eamm_amd/train_graph.py composes them (with batchnorm_nhwc.hip and motion_backward.hip) into the
// generator's end-to-end backward: loss.backward() as in train.py:133.
#include "conv_common.h"

#include <algorithm>
#include <cstdlib>

namespace eamm {

namespace {

struct BilinearB {
    int x0, y0;
    float ax, ay;
    bool valid;
};
// the forward's bilinear_setup (motion.hip), keeping the fractions for the derivative
__device__ __forceinline__ BilinearB bilinear_setup_b(float gx, float gy, int W, int H) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    BilinearB b;
    b.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W);
    b.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);
    b.ax = ix - fx;
    b.ay = iy - fy;
    b.valid = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    return b;
}

}  // namespace

// one thread per (pixel, 4 channels); LPP lanes per pixel (C / 4) reduce the flow / occlusion partials before one atomic
__global__ __launch_bounds__(256) void warp_features_backward_kernel(const float* __restrict__ feat, const float* __restrict__ defo,
                                                                     const float* __restrict__ occ, const float* __restrict__ dout,
                                                                     int n, int ns, int hf, int wf, int C, float* __restrict__ dfeat,
                                                                     float* __restrict__ ddefo, float* __restrict__ docc) {
    const int c4n = C >> 2;
    const size_t total = (size_t)n * hf * wf * c4n;
    const bool pow2 = (c4n & (c4n - 1)) == 0 && c4n <= 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t pix = idx / c4n;
        const int f = (int)(pix / ((size_t)wf * hf));
        const float2 g = reinterpret_cast<const float2*>(defo)[pix];
        const float o = occ ? occ[pix] : 1.f;
        const BilinearB b = bilinear_setup_b(g.x, g.y, wf, hf);
        // a thread owns channels c4 + k * c4n (k = 0..3): the lanes of a pixel touch CONSECUTIVE floats, so each atomic
        // instruction of the wave lands on whole cache lines (4 x fewer L2 atomic operations than a float4 per lane)
        const float* gp = dout + pix * C + c4;
        const float go[4] = {gp[0], gp[c4n], gp[2 * c4n], gp[3 * c4n]};
        const size_t sbase = (size_t)((ns == 1) ? 0 : f) * hf * wf;
        float dgx = 0.f, dgy = 0.f, dsum = 0.f;
        if (b.valid) {
            const float wx[2] = {1.f - b.ax, b.ax}, wy[2] = {1.f - b.ay, b.ay};
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const int yy = b.y0 + cy, xx = b.x0 + cx;
                    if ((unsigned)yy >= (unsigned)hf || (unsigned)xx >= (unsigned)wf) continue;   // zeros padding: no value, no gradient
                    const size_t src = (sbase + (size_t)yy * wf + xx) * C + c4;
                    const float* vp = feat + src;
                    const float dot = go[0] * vp[0] + go[1] * vp[c4n] + go[2] * vp[2 * c4n] + go[3] * vp[3 * c4n];   // sum_c dout * feat
                    const float w = wx[cx] * wy[cy];
                    dsum = fmaf(w, dot, dsum);
                    dgx = fmaf((cx ? 1.f : -1.f) * wy[cy], dot, dgx);                            // d w / d ix
                    dgy = fmaf((cy ? 1.f : -1.f) * wx[cx], dot, dgy);                            // d w / d iy
                    if (dfeat != nullptr) {
                        float* d = dfeat + src;
                        const float s = w * o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) atomicAdd(d + k * c4n, go[k] * s);
                    }
                }
        }
        // ix = ((gx + 1) W - 1) / 2: d ix / d gx = W / 2 (likewise y); the occlusion factor multiplies the sampled value
        float pgx = dgx * o * 0.5f * (float)wf, pgy = dgy * o * 0.5f * (float)hf, po = dsum;
        if (pow2) {   // the c4n lanes of a pixel are a power-of-two aligned group of the wave
            for (int off = c4n >> 1; off > 0; off >>= 1) {
                pgx += __shfl_xor(pgx, off, 64);
                pgy += __shfl_xor(pgy, off, 64);
                po += __shfl_xor(po, off, 64);
            }
            if (c4 == 0) {
                if (ddefo != nullptr) {
                    atomicAdd(ddefo + pix * 2, pgx);
                    atomicAdd(ddefo + pix * 2 + 1, pgy);
                }
                if (docc != nullptr) atomicAdd(docc + pix, po);
            }
        } else {
            if (ddefo != nullptr) {
                atomicAdd(ddefo + pix * 2, pgx);
                atomicAdd(ddefo + pix * 2 + 1, pgy);
            }
            if (docc != nullptr) atomicAdd(docc + pix, po);
        }
    }
}

hipError_t warp_features_backward_launch(const float* feat, const float* defo, const float* occ, const float* dout, int n, int ns,
                                         int hf, int wf, int C, float* dfeat, float* ddefo, float* docc, hipStream_t s) {
    if (C % 4 || n < 1) return hipErrorInvalidValue;
    const size_t total = (size_t)n * hf * wf * (C / 4);
    // the in-wave reduction needs every lane of a pixel's group alive: whole groups per block, and a grid that covers the
    // tensor in one sweep of whole blocks (256 is a multiple of every power-of-two group size)
    const size_t blocks = (total + 255) / 256;
    if (blocks > (size_t)0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_features_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, s, feat, defo, occ, dout, n, ns, hf, wf, C,
                       dfeat, ddefo, docc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// convolution weight gradient
// ---------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;      // [B,H,W,Cin]
    const float* dy;     // [B,H,W,Cout]
    int B, H, W, Cin, Cout, kh, kw;
    long long P;         // B*H*W
    long long per_split; // pixels per split (multiple of 64)
    int splits, mt, nt;  // pixel splits, Cout tiles, Cin tiles (64 each)
    float* partial;      // [splits][planes*kh*kw][mt*64][nt*64]
    int planes;          // independent GEMMs stacked along the grid (the 36 transform points of the Winograd form; else 1)
    long long a_plane, b_plane;   // their strides in dy / x (floats)
};

// block (tile pair, tap, split): D[64 co][64 ci] += sum over the split's pixels; 4 waves of 32 x 32, v_mfma_f32_32x32x2_f32.
// LDS per KC-pixel chunk: dY rows [KC][64] and (shifted, zero-padded) X rows [KC][64]: both as they lie in HBM.  The next
// chunk's rows are in flight in registers while this chunk's MFMAs run; a row's (x, y) advances incrementally -- on gfx950 a
// wave streaming f32 MFMAs starves the VALU instructions of its SIMD's other waves, so the loop carries as few as it can.
template <int KC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int TM = 64, TN = 64, LD = TM + 4, RPT = KC / 16;   // +4: the two K rows of an MFMA step land on different banks
    __shared__ __attribute__((aligned(16))) float As[KC * LD];
    __shared__ __attribute__((aligned(16))) float Bs[KC * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int L = blockIdx.x;
    const int nti = L % p.nt; L /= p.nt;
    const int mti = L % p.mt; L /= p.mt;
    const int taps = p.kh * p.kw;
    const int tap = L % taps; L /= taps;
    const int plane = L % p.planes; L /= p.planes;
    const int split = L;
    const int dyo = tap / p.kw - p.kh / 2, dxo = tap % p.kw - p.kw / 2;
    const int p0 = (int)((long long)split * p.per_split), p1 = (int)std::min<long long>(p.P, (long long)p0 + p.per_split);
    const float* const pdy = p.dy + plane * p.a_plane;
    const float* const px_ = p.x + plane * p.b_plane;
    f32x16 acc;
    static_for<16>([&](auto rc) { acc[decltype(rc)::value] = 0.f; });
    // loader: thread -> (rows lr + 16 j of the chunk, 16-byte group lq of the 64 channels)
    const int lq = tid & 15, lr = tid >> 4;
    const int co = mti * TM + lq * 4, ci = nti * TN + lq * 4;
    const bool co_ok = co < p.Cout, ci_ok = ci < p.Cin;
    int px[RPT], py[RPT];
#pragma unroll
    for (int h2 = 0; h2 < RPT; ++h2) {
        const unsigned pp = (unsigned)(p0 + lr + 16 * h2);
        px[h2] = (int)(pp % (unsigned)p.W);
        py[h2] = (int)((pp / (unsigned)p.W) % (unsigned)p.H);
    }
    const float* ap = pdy + (size_t)(p0 + lr) * p.Cout + co;                                   // row lr of the current chunk
    const float* bp = px_ + ((long long)(p0 + lr) + (long long)dyo * p.W + dxo) * p.Cin + ci;   // the tap's shifted pixel
    const size_t a16 = (size_t)16 * p.Cout, b16 = (size_t)16 * p.Cin;
    f32x4 ra[RPT], rb[RPT];
    auto fetch = [&](int pc) {
#pragma unroll
        for (int h2 = 0; h2 < RPT; ++h2) {
            const bool in = pc + lr + 16 * h2 < p1;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[h2] = (in && co_ok) ? *reinterpret_cast<const f32x4*>(ap + h2 * a16) : z;
            const bool tap_in = (unsigned)(py[h2] + dyo) < (unsigned)p.H && (unsigned)(px[h2] + dxo) < (unsigned)p.W;
            rb[h2] = (in && ci_ok && tap_in) ? *reinterpret_cast<const f32x4*>(bp + h2 * b16) : z;
        }
    };
    auto advance = [&]() {
        ap += RPT * a16;
        bp += RPT * b16;
#pragma unroll
        for (int h2 = 0; h2 < RPT; ++h2) {
            px[h2] += KC;
            if (p.W >= KC) {
                if (px[h2] >= p.W) { px[h2] -= p.W; ++py[h2]; }
            } else {
                py[h2] += px[h2] / p.W;
                px[h2] %= p.W;
            }
            if (py[h2] >= p.H) py[h2] = (p.W >= KC) ? py[h2] - p.H : py[h2] % p.H;
        }
    };
    fetch(p0);
    for (int pc = p0; pc < p1; pc += KC) {
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < RPT; ++h2) {
            *reinterpret_cast<f32x4*>(As + (lr + 16 * h2) * LD + lq * 4) = ra[h2];
            *reinterpret_cast<f32x4*>(Bs + (lr + 16 * h2) * LD + lq * 4) = rb[h2];
        }
        __syncthreads();
        if (pc + KC < p1) {
            advance();
            fetch(pc + KC);
        }
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {   // A[m][k]: lane -> m = lane % 32, k = lane / 32
            const float av = As[(2 * s + (lane >> 5)) * LD + wm * 32 + (lane & 31)];
            const float bv = Bs[(2 * s + (lane >> 5)) * LD + wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // D layout of the 32x32 MFMA: lane -> column n = lane % 32; register r -> row m = (r % 4) + 8 * (r / 4) + 4 * (lane / 32)
    float* out = p.partial + (((size_t)split * taps * p.planes + (size_t)plane * taps + tap) * (p.mt * TM) + mti * TM + wm * 32) * (size_t)(p.nt * TN) + nti * TN + wn * 32;
    static_for<16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(size_t)m * (p.nt * TN) + (lane & 31)] = acc[r];
    });
}

// The same GEMM with the KW taps of one filter ROW in one block, for maps whose width is a multiple of the 32-pixel K chunk
// (every layer of the generator at 64 x 64 and above): a chunk is then a segment of ONE image row, the taps' shifted operands
// are the same X rows offset by one (LDS rows k + t), the zero padding is exact on the B side (the KW - 1 halo rows are
// loaded as zeros at the row ends, the whole chunk as zeros when the tap row leaves the map) and the loop has no masking.
// HBM/L2 bytes per flop drop KW-fold against the per-tap kernel (which moves 16 flop per byte and is L2-bound at ~0.5 of
// the matrix peak): dY 8 KB + X 8.5 KB per 3 x 262144 flop at 3x3.
template <int KW>
__global__ __launch_bounds__(256) void conv_wgrad_row_kernel(const WgradArgs p) {
    constexpr int KC = 32, TM = 64, TN = 64, LD = TM + 4, HALF = KW / 2;
    __shared__ __attribute__((aligned(16))) float As[KC * LD];
    __shared__ __attribute__((aligned(16))) float Bs[(KC + KW - 1) * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int L = blockIdx.x;
    const int nti = L % p.nt; L /= p.nt;
    const int mti = L % p.mt; L /= p.mt;
    const int trow = L % p.kh; L /= p.kh;
    const int split = L;
    const int dyo = trow - p.kh / 2;
    const int p0 = (int)((long long)split * p.per_split), p1 = (int)std::min<long long>(p.P, (long long)p0 + p.per_split);   // multiples of 32
    f32x16 acc[KW];
    static_for<KW>([&](auto tc) { static_for<16>([&](auto rc) { acc[decltype(tc)::value][decltype(rc)::value] = 0.f; }); });
    const int lq = tid & 15, lr = tid >> 4;
    const int co = mti * TM + lq * 4, ci = nti * TN + lq * 4;
    const bool co_ok = co < p.Cout, ci_ok = ci < p.Cin;
    int x0 = (int)((unsigned)p0 % (unsigned)p.W), y = (int)(((unsigned)p0 / (unsigned)p.W) % (unsigned)p.H);
    long long aoff = (long long)(p0 + lr) * p.Cout + co;
    long long boff = ((long long)(p0 + lr) + (long long)dyo * p.W) * p.Cin + ci;
    // halo rows: threads 0 .. 16 (KW - 1) - 1 own one each; pixel offset from the chunk start -HALF..-1 and 32..32 + HALF - 1
    const bool is_halo = tid < 16 * (KW - 1);
    const int hrow = lr, hoff = hrow < HALF ? hrow - HALF : KC + hrow - HALF;
    long long hoffs = ((long long)p0 + hoff + (long long)dyo * p.W) * p.Cin + ci;
    const long long a16 = (long long)16 * p.Cout, b16 = (long long)16 * p.Cin;
    f32x4 ra[2], rb[2], rh;
    auto fetch = [&]() {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const bool yv = (unsigned)(y + dyo) < (unsigned)p.H;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            ra[h2] = co_ok ? *reinterpret_cast<const f32x4*>(p.dy + aoff + h2 * a16) : z;
            rb[h2] = (ci_ok && yv) ? *reinterpret_cast<const f32x4*>(p.x + boff + h2 * b16) : z;
        }
        if (KW > 1) {
            const bool xv = hrow < HALF ? x0 > 0 : x0 + KC < p.W;
            rh = (is_halo && ci_ok && yv && xv) ? *reinterpret_cast<const f32x4*>(p.x + hoffs) : z;
        }
    };
    fetch();
    for (int pc = p0; pc < p1; pc += KC) {
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            *reinterpret_cast<f32x4*>(As + (lr + 16 * h2) * LD + lq * 4) = ra[h2];
            *reinterpret_cast<f32x4*>(Bs + (lr + 16 * h2 + HALF) * LD + lq * 4) = rb[h2];
        }
        if (KW > 1 && is_halo) *reinterpret_cast<f32x4*>(Bs + (hoff + HALF) * LD + lq * 4) = rh;
        __syncthreads();
        if (pc + KC < p1) {
            aoff += 2 * a16;
            boff += 2 * b16;
            hoffs += 2 * b16;
            x0 += KC;
            if (x0 == p.W) {
                x0 = 0;
                if (++y == p.H) y = 0;
            }
            fetch();
        }
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const int k = 2 * s + (lane >> 5);
            const float av = As[k * LD + wm * 32 + (lane & 31)];
            static_for<KW>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const float bv = Bs[(k + t) * LD + wn * 32 + (lane & 31)];   // pixel k + (t - HALF) of the chunk's row
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            });
        }
    }
    const int taps = p.kh * KW;
    static_for<KW>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        float* out = p.partial + (((size_t)split * taps + trow * KW + t) * (p.mt * TM) + mti * TM + wm * 32) * (size_t)(p.nt * TN) +
                     nti * TN + wn * 32;
        static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(size_t)m * (p.nt * TN) + (lane & 31)] = acc[t][r];
        });
    });
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 weight gradient in Winograd form F(3x3, 4x4): the filter gradient of a 4x4 output tile is itself a minimal-filtering
// problem -- dW[k][l] = sum_{a,b<4} dY[a][b] d[a+k][b+l] over the tile's 6x6 input patch d, the same correlation as the forward
// with the roles of filter and output exchanged -- so with the forward's points {0, +-1, +-2, inf}
//     dW = sum_tiles  A'^T [ (G' g G'^T) (.) (B^T d B) ] A',    g = the tile of dY,
//     G' (6x4) rows p^k / N_p: [1/4 0 0 0; -1/6 -1/6 -1/6 -1/6; -1/6 1/6 -1/6 1/6; 1/24 1/12 1/6 1/3; 1/24 -1/12 1/6 -1/3; 0 0 0 1],
//     A'^T (3x6) = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1],
// and B^T d B is exactly the forward's V (wino4_input_transform_kernel).  36 multiplies per (tile, cin, cout) instead of 144:
// the sum over tiles is 36 independent GEMMs  M_xi[co][ci] = sum_q Yhat_xi[q][co] V_xi[q][ci]  (K = tiles, both operands K-major)
// on the per-tap GEMM kernel above with the transform points as its `planes`, then one 36 -> 9 output transform per (co, ci).
// (Identity checked in float64 in tests/test_transform_identities.py.)
template <typename T> __device__ __forceinline__ void gp4(const T g0, const T g1, const T g2, const T g3, T* r) {
    const T e = g0 + g2, o = g1 + g3;
    r[0] = 0.25f * g0;
    r[1] = (-1.f / 6.f) * (e + o);
    r[2] = (-1.f / 6.f) * (e - o);
    const T e2 = (1.f / 24.f) * g0 + (1.f / 6.f) * g2, o2 = (1.f / 12.f) * g1 + (1.f / 3.f) * g3;
    r[3] = e2 + o2;
    r[4] = e2 - o2;
    r[5] = g3;
}

// one thread per (tile, 4 channels): 16 loads, 36 stores; Yhat[xi][q][c], q = (b, qy, qx) as the forward's V
__global__ __launch_bounds__(256) void wino4_dy_transform_kernel(const float* __restrict__ dy, int B, int H, int W, int C,
                                                                 float* __restrict__ Yh) {
    const int c4n = C >> 2, Hq = H >> 2, Wq = W >> 2;
    const size_t Mq = (size_t)B * Hq * Wq, total = Mq * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t q = idx / c4n;
        const int qx = (int)(q % Wq), qy = (int)((q / Wq) % Hq);
        const size_t b = q / ((size_t)Wq * Hq);
        const f32x4* img = reinterpret_cast<const f32x4*>(dy) + ((b * H + 4 * qy) * W + 4 * qx) * c4n + c4;
        f32x4 g[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) g[a][bb] = img[((size_t)a * W + bb) * c4n];
        f32x4 t[6][4];   // G' g (columns)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            f32x4 r[6];
            gp4(g[0][bb], g[1][bb], g[2][bb], g[3][bb], r);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][bb] = r[i];
        }
        f32x4* out = reinterpret_cast<f32x4*>(Yh) + q * c4n + c4;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x4 r[6];
            gp4(t[i][0], t[i][1], t[i][2], t[i][3], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * Mq * c4n] = r[j];
        }
    }
}

// dW[co][ci][3][3] = A'^T (sum over the splits of M[36][co][ci]) A', in a fixed order
__global__ void wino4_wgrad_output_kernel(const float* __restrict__ partial, int splits, int Mpad, int Npad, int Cout, int Cin,
                                          float* __restrict__ dw) {
    const size_t total = (size_t)Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        float m[6][6];   // splits outer, the 36 points inner: 36 independent loads in flight per step, each point's order fixed
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) m[xi / 6][xi % 6] = 0.f;
        for (int k = 0; k < splits; ++k) {
            const float* q = partial + ((size_t)k * 36 * Mpad + co) * Npad + ci;
#pragma unroll
            for (int xi = 0; xi < 36; ++xi) m[xi / 6][xi % 6] += q[(size_t)xi * Mpad * Npad];
        }
        float t[3][6];   // A'^T m
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float s12 = m[1][j] + m[2][j], d12 = m[1][j] - m[2][j], s34 = m[3][j] + m[4][j], d34 = m[3][j] - m[4][j];
            t[0][j] = m[0][j] + s12 + s34;
            t[1][j] = d12 + 2.f * d34;
            t[2][j] = s12 + 4.f * s34 + m[5][j];
        }
        float* o = dw + i * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s12 = t[k][1] + t[k][2], d12 = t[k][1] - t[k][2], s34 = t[k][3] + t[k][4], d34 = t[k][3] - t[k][4];
            o[k * 3 + 0] = t[k][0] + s12 + s34;
            o[k * 3 + 1] = d12 + 2.f * d34;
            o[k * 3 + 2] = s12 + 4.f * s34 + t[k][5];
        }
    }
}

// dW[co][ci][tap] (OIHW) = sum over the splits, in a fixed order
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int taps, int Mpad, int Npad, int Cout, int Cin,
                                         float* __restrict__ dw) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int ci = (int)((i / taps) % Cin);
        const int co = (int)(i / ((size_t)taps * Cin));
        float s = 0.f;
#pragma unroll 8   // independent loads in flight; the additions keep their order
        for (int k = 0; k < splits; ++k) s += partial[(((size_t)k * taps + tap) * Mpad + co) * Npad + ci];
        dw[i] = s;
    }
}

// db[co] = sum_p dy[p][co] in two fixed-order stages: BIAS_PARTS pixel ranges (a thread owns 4 channels, the 256 / (Cout / 4)
// row lanes of a block stride the range), then the parts per channel in double
constexpr int BIAS_PARTS = 256;
__global__ __launch_bounds__(256) void conv_bias_grad_partial_kernel(const float* __restrict__ dy, long long P, int Cout,
                                                                     float* __restrict__ part /*[parts][Cout]*/) {
    __shared__ f32x4 red[256];
    const int c4n = Cout >> 2;                      // <= 256 (Cout <= 1024)
    const int rows = 256 / c4n, c4 = threadIdx.x % c4n, r = threadIdx.x / c4n;
    const long long per = (P + gridDim.x - 1) / gridDim.x, p0 = (long long)blockIdx.x * per, p1 = std::min<long long>(P, p0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
#pragma unroll 8
        for (long long pp = p0 + r; pp < p1; pp += rows) s += reinterpret_cast<const f32x4*>(dy)[pp * c4n + c4];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (r == 0) {
        for (int k = 1; k < rows; ++k) s += red[k * c4n + c4];
        reinterpret_cast<f32x4*>(part)[(size_t)blockIdx.x * c4n + c4] = s;
    }
}
// 64 channels x 4 part lanes per block: each lane adds every 4th part in double, LDS folds the four
__global__ __launch_bounds__(256) void conv_bias_grad_final_kernel(const float* __restrict__ part, int parts, int Cout, float* __restrict__ db) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, r = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < Cout)
        for (int k = r; k < parts; k += 4) s += (double)part[(size_t)k * Cout + c];
    red[r][cl] = s;
    __syncthreads();
    if (r == 0 && c < Cout) db[c] = (float)((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

size_t conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw);

namespace {
// F(3x3,4x4) form: 3x3, map sides multiples of 4, enough tiles for the GEMMs' K (EAMM_WGRAD_WINO4 = 0 turns it off)
bool wgrad_wino4_applies(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    static const int off = knob_int("EAMM_WGRAD_WINO4", 1) == 0;
    static const long long min_tiles = knob_int("EAMM_WGRAD_WINO4_MIN_TILES", 512);
    return !off && kh == 3 && kw == 3 && !(H & 3) && !(W & 3) && !(Cin & 3) && !(Cout & 3) &&
           (long long)B * (H / 4) * (W / 4) >= min_tiles;
}
long long wgrad_splits(long long tiles, long long K, long long min_k) {
    return std::max<long long>(1, std::min<long long>(1024 / std::max<long long>(1, tiles) + 1, K / min_k));
}
size_t round64(size_t n) { return (n + 63) / 64 * 64; }
}  // namespace

bool conv_wgrad_takes_transformed(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    return wgrad_wino4_applies(B, H, W, Cin, Cout, kh, kw);
}

// x_transformed (optional, only in the F(3x3,4x4) form): V = B^T x B as wino4_transform_launch wrote it for the forward convolution
// of the same x ([36][tiles][Cin]) -- the transform is then not run again
hipError_t conv_wgrad_launch(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int kh, int kw, float* dweight,
                             float* dbias, float* workspace, size_t workspace_floats, hipStream_t s, const float* x_transformed) {
    if ((Cin & 3) || (Cout & 3) || Cout > 1024 || !(kh & 1) || !(kw & 1) || B < 1 || (long long)B * H * W >= (1ll << 30)) return hipErrorInvalidValue;
    if (workspace_floats < conv_wgrad_workspace_floats(B, H, W, Cin, Cout, kh, kw)) return hipErrorInvalidValue;
    const size_t bias_floats = (size_t)BIAS_PARTS * Cout;   // the tail of the workspace
    workspace_floats -= bias_floats;
    float* bias_part = workspace + workspace_floats;
    WgradArgs a{};
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw;
    a.P = (long long)B * H * W;
    a.mt = (Cout + 63) / 64;
    a.nt = (Cin + 63) / 64;
    a.planes = 1;
    const int taps = kh * kw;
    if (wgrad_wino4_applies(B, H, W, Cin, Cout, kh, kw)) {
        const long long Mq = (long long)B * (H / 4) * (W / 4);
        float* V = workspace;
        float* Yh = V + round64((size_t)36 * Mq * Cin);
        float* partial = Yh + round64((size_t)36 * Mq * Cout);
        if (x_transformed) {
            V = const_cast<float*>(x_transformed);
        } else {
            hipError_t e = wino4_transform_launch(x, nullptr, nullptr, B, H, W, Cin, V, s);
            if (e != hipSuccess) return e;
        }
        const size_t ty = (size_t)Mq * (Cout / 4);
        hipLaunchKernelGGL(wino4_dy_transform_kernel, dim3((unsigned)std::min<size_t>((ty + 255) / 256, (size_t)1 << 20)), dim3(256), 0, s, dy,
                           B, H, W, Cout, Yh);
        // the 36 GEMMs as planes of the per-tap kernel: a "1x1 filter" over Mq "pixels" in one row
        a.x = V;
        a.dy = Yh;
        a.kh = a.kw = 1;
        a.H = 1;
        a.W = (int)Mq;
        a.P = Mq;
        a.planes = 36;
        a.a_plane = Mq * Cout;
        a.b_plane = Mq * Cin;
        // (a 128 x 128 block tile for these GEMMs -- twice the flop per operand byte -- measured slower: 0.446 vs 0.398 ms at
        //  256 -> 256 @ 64 x 64 x 16; many small blocks hide the f32 MFMA's issue stalls better than few large ones)
        const long long tiles = (long long)a.mt * a.nt * 36;
        const long long splits = wgrad_splits(tiles, Mq, 128);
        a.per_split = ((Mq + splits - 1) / splits + 63) / 64 * 64;
        a.splits = (int)((Mq + a.per_split - 1) / a.per_split);
        a.partial = partial;
        note_mfma_flops(2.0 * 36.0 * (double)Mq * (a.mt * 64.0) * (a.nt * 64.0));      // 36 transform-point GEMMs over the tiles
        hipLaunchKernelGGL(conv_wgrad_kernel<32>, dim3((unsigned)(tiles * a.splits)), dim3(256), 0, s, a);
        const size_t pairs = (size_t)Cout * Cin;
        hipLaunchKernelGGL(wino4_wgrad_output_kernel, dim3((unsigned)std::min<size_t>((pairs + 255) / 256, 65535)), dim3(256), 0, s, partial,
                           a.splits, a.mt * 64, a.nt * 64, Cout, Cin, dweight);
    } else {
        a.x = x;
        a.dy = dy;
        static const int row_off = knob_int("EAMM_WGRAD_ROW", 1) == 0;
        const bool row = !row_off && W % 32 == 0 && (kw == 1 || kw == 3 || kw == 7);   // one block per filter row (else: per tap)
        const long long tiles = (long long)a.mt * a.nt * (row ? kh : taps);
        const long long splits = wgrad_splits(tiles, a.P, 256);
        a.per_split = ((a.P + splits - 1) / splits + 63) / 64 * 64;
        a.splits = (int)((a.P + a.per_split - 1) / a.per_split);
        a.partial = workspace;
        const dim3 grid((unsigned)(tiles * a.splits));
        note_mfma_flops(2.0 * taps * (double)a.P * (a.mt * 64.0) * (a.nt * 64.0));            // one GEMM over the pixels per tap
        if (row && kw == 1) hipLaunchKernelGGL(conv_wgrad_row_kernel<1>, grid, dim3(256), 0, s, a);
        else if (row && kw == 3) hipLaunchKernelGGL(conv_wgrad_row_kernel<3>, grid, dim3(256), 0, s, a);
        else if (row) hipLaunchKernelGGL(conv_wgrad_row_kernel<7>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<32>, grid, dim3(256), 0, s, a);
        const size_t total = (size_t)Cout * Cin * taps;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, workspace,
                           a.splits, taps, a.mt * 64, a.nt * 64, Cout, Cin, dweight);
    }
    if (dbias != nullptr) {
        const long long P = (long long)B * H * W;
        const int parts = (int)std::min<long long>(BIAS_PARTS, (P + 63) / 64);
        hipLaunchKernelGGL(conv_bias_grad_partial_kernel, dim3(parts), dim3(256), 0, s, dy, P, Cout, bias_part);
        hipLaunchKernelGGL(conv_bias_grad_final_kernel, dim3((Cout + 63) / 64), dim3(256), 0, s, bias_part, parts, Cout, dbias);
    }
    return hipGetLastError();
}

size_t conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    const size_t per_tap = (size_t)((Cout + 63) / 64) * 64 * ((Cin + 63) / 64) * 64;
    const long long mn = (long long)((Cout + 63) / 64) * ((Cin + 63) / 64);
    size_t need;
    if (wgrad_wino4_applies(B, H, W, Cin, Cout, kh, kw)) {
        const long long Mq = (long long)B * (H / 4) * (W / 4);
        const size_t part = per_tap * 36 * (size_t)wgrad_splits(mn * 36, Mq, 128);
        need = round64((size_t)36 * Mq * Cin) + round64((size_t)36 * Mq * Cout) + part;
    } else {
        // the row kernel's block count (the per-tap kernel has kw x more tiles, hence fewer splits)
        need = per_tap * kh * kw * (size_t)wgrad_splits(mn * kh, (long long)B * H * W, 256);
    }
    return round64(need) + (size_t)BIAS_PARTS * Cout;
}

}  // namespace eamm
