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
// These are the kernels; composing them into the generator's end-to-end backward is not done.
#include "conv_common.h"

#include <algorithm>

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
        const float4 go = reinterpret_cast<const float4*>(dout)[idx];
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
                    const size_t src = (sbase + (size_t)yy * wf + xx) * c4n + c4;
                    const float4 v = reinterpret_cast<const float4*>(feat)[src];
                    const float dot = go.x * v.x + go.y * v.y + go.z * v.z + go.w * v.w;       // sum_c dout * feat at this corner
                    const float w = wx[cx] * wy[cy];
                    dsum = fmaf(w, dot, dsum);
                    dgx = fmaf((cx ? 1.f : -1.f) * wy[cy], dot, dgx);                            // d w / d ix
                    dgy = fmaf((cy ? 1.f : -1.f) * wx[cx], dot, dgy);                            // d w / d iy
                    if (dfeat != nullptr) {
                        float* d = dfeat + src * 4;
                        const float s = w * o;
                        atomicAdd(d + 0, go.x * s); atomicAdd(d + 1, go.y * s); atomicAdd(d + 2, go.z * s); atomicAdd(d + 3, go.w * s);
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
    long long per_split; // pixels per split (multiple of 32)
    int splits, mt, nt;  // pixel splits, Cout tiles, Cin tiles (64 each)
    float* partial;      // [splits][kh*kw][mt*64][nt*64]
};

// block (tile pair, tap, split): D[64 co][64 ci] += sum over the split's pixels; 4 waves of 32 x 32, v_mfma_f32_32x32x2_f32.
// LDS per 32-pixel chunk: dY rows [32][64] and (shifted, zero-padded) X rows [32][64]: both read as they lie in HBM.
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int KC = 32, TM = 64, TN = 64, LD = TM + 4;   // +4: the two K rows of an MFMA step land on different banks
    __shared__ __attribute__((aligned(16))) float As[KC * LD];
    __shared__ __attribute__((aligned(16))) float Bs[KC * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int L = blockIdx.x;
    const int nti = L % p.nt; L /= p.nt;
    const int mti = L % p.mt; L /= p.mt;
    const int taps = p.kh * p.kw;
    const int tap = L % taps; L /= taps;
    const int split = L;
    const int dyo = tap / p.kw - p.kh / 2, dxo = tap % p.kw - p.kw / 2;
    const long long p0 = (long long)split * p.per_split, p1 = std::min<long long>(p.P, p0 + p.per_split);
    f32x16 acc;
    static_for<16>([&](auto rc) { acc[decltype(rc)::value] = 0.f; });
    // loader: thread -> (row r of the chunk, 16-byte group q of the 64 channels); 256 threads cover 16 rows x 16 groups, twice
    const int lq = tid & 15, lr = tid >> 4;
    for (long long pc = p0; pc < p1; pc += KC) {
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = lr + 16 * h2;
            const long long pp = pc + r;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (pp < p1) {
                const int co = mti * TM + lq * 4, ci = nti * TN + lq * 4;
                if (co < p.Cout) a = *reinterpret_cast<const f32x4*>(p.dy + pp * p.Cout + co);      // (Cout, Cin multiples of 4)
                const int xx = (int)(pp % p.W), yy = (int)((pp / p.W) % p.H);
                const int ys = yy + dyo, xs = xx + dxo;
                if (ci < p.Cin && (unsigned)ys < (unsigned)p.H && (unsigned)xs < (unsigned)p.W)
                    b = *reinterpret_cast<const f32x4*>(p.x + (pp + (long long)dyo * p.W + dxo) * p.Cin + ci);
            }
            *reinterpret_cast<f32x4*>(As + r * LD + lq * 4) = a;
            *reinterpret_cast<f32x4*>(Bs + r * LD + lq * 4) = b;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {   // A[m][k]: lane -> m = lane % 32, k = lane / 32
            const float av = As[(2 * s + (lane >> 5)) * LD + wm * 32 + (lane & 31)];
            const float bv = Bs[(2 * s + (lane >> 5)) * LD + wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // D layout of the 32x32 MFMA: lane -> column n = lane % 32; register r -> row m = (r % 4) + 8 * (r / 4) + 4 * (lane / 32)
    float* out = p.partial + (((size_t)split * taps + tap) * (p.mt * TM) + mti * TM + wm * 32) * (size_t)(p.nt * TN) + nti * TN + wn * 32;
    static_for<16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(size_t)m * (p.nt * TN) + (lane & 31)] = acc[r];
    });
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
        for (int k = 0; k < splits; ++k) s += partial[(((size_t)k * taps + tap) * Mpad + co) * Npad + ci];
        dw[i] = s;
    }
}

// db[co] = sum_p dy[p][co]: one block per 64 channels, double accumulators, fixed order
__global__ __launch_bounds__(256) void conv_bias_grad_kernel(const float* __restrict__ dy, long long P, int Cout, float* __restrict__ db) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
    double s = 0.0;
    if (c < Cout)
        for (long long pp = lane_r; pp < P; pp += 4) s += (double)dy[pp * Cout + c];
    red[lane_r][threadIdx.x & 63] = s;
    __syncthreads();
    if (lane_r == 0 && c < Cout) db[c] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

hipError_t conv_wgrad_launch(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int kh, int kw, float* dweight,
                             float* dbias, float* workspace, size_t workspace_floats, hipStream_t s) {
    if ((Cin & 3) || (Cout & 3) || !(kh & 1) || !(kw & 1) || B < 1) return hipErrorInvalidValue;
    WgradArgs a{};
    a.x = x;
    a.dy = dy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw;
    a.P = (long long)B * H * W;
    a.mt = (Cout + 63) / 64;
    a.nt = (Cin + 63) / 64;
    const int taps = kh * kw;
    const long long tiles = (long long)a.mt * a.nt * taps;
    long long splits = std::max<long long>(1, std::min<long long>(1024 / std::max<long long>(1, tiles) + 1, a.P / 256));
    const size_t per = (size_t)taps * a.mt * 64 * a.nt * 64;
    while (splits > 1 && (size_t)splits * per > workspace_floats) --splits;
    if ((size_t)splits * per > workspace_floats) return hipErrorInvalidValue;
    a.per_split = ((a.P + splits - 1) / splits + 31) / 32 * 32;
    a.splits = (int)((a.P + a.per_split - 1) / a.per_split);
    a.partial = workspace;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)(tiles * a.splits)), dim3(256), 0, s, a);
    const size_t total = (size_t)Cout * Cin * taps;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, workspace,
                       a.splits, taps, a.mt * 64, a.nt * 64, Cout, Cin, dweight);
    if (dbias != nullptr)
        hipLaunchKernelGGL(conv_bias_grad_kernel, dim3((Cout + 63) / 64), dim3(256), 0, s, dy, a.P, Cout, dbias);
    return hipGetLastError();
}

size_t conv_wgrad_workspace_floats(int Cin, int Cout, int kh, int kw) {
    const size_t per = (size_t)kh * kw * ((Cout + 63) / 64) * 64 * ((Cin + 63) / 64) * 64;
    const long long tiles = (long long)((Cout + 63) / 64) * ((Cin + 63) / 64) * kh * kw;
    return per * (size_t)(1024 / std::max<long long>(1, tiles) + 1);
}

}  // namespace eamm
