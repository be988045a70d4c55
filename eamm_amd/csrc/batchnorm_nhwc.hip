// Training-mode BatchNorm on the engine's NHWC activations (second slice: the generator's
// forward in .train() mode).  Same statistics, same packed `sums` layout and the same finalize kernel as batchnorm.hip
// (reference sync_batchnorm/batchnorm.py:46-125); only the tensor layout differs: x is [M, C] with the C channels of a pixel
// contiguous, so a wave reads whole 16-byte channel groups of consecutive pixels and every thread keeps the (sum, sum of
// squares) of FOUR channels in double.
//   bn_nhwc_partial_kernel  x read once; partial[(c, slice)] = (sum, sum of squares)      -> bn_combine_kernel (batchnorm.hip)
//   bn_nhwc_apply_kernel    y = act((x - mean[c]) * scale[c] + bias[c]), optionally followed by the 2x2 average of
//                           DownBlock2d (reference modules/util.py:903-921: conv -> norm -> relu -> pool)
#include "kernels.h"

#include <algorithm>

namespace eamm {

namespace {
constexpr int NB_THREADS = 256;
}

// grid (R): block r covers rows r, r + R, ... in strides; thread t owns channel group t % C4 and row lane t / C4
__global__ __launch_bounds__(NB_THREADS) void bn_nhwc_partial_kernel(const float4* __restrict__ x, long long M, int C4, int R,
                                                                      double* __restrict__ partial) {
    extern __shared__ double red[];   // [row lanes][C4][8]
    const int c4 = threadIdx.x % C4, lane_r = threadIdx.x / C4, lanes = NB_THREADS / C4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (lane_r < lanes) {
#pragma unroll 4
        for (long long m = (long long)blockIdx.x * lanes + lane_r; m < M; m += (long long)R * lanes) {
            const float4 v = x[m * C4 + c4];
            s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
        double* dst = red + ((size_t)lane_r * C4 + c4) * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[2 * k] = s[k];
            dst[2 * k + 1] = q[k];
        }
    }
    __syncthreads();
    // row lanes added in a fixed order by the first C4 * 4 threads (one per channel)
    for (int i = threadIdx.x; i < C4 * 4; i += NB_THREADS) {
        const int g4 = i >> 2, k = i & 3;
        double a = 0.0, b = 0.0;
        for (int l = 0; l < lanes; ++l) {
            a += red[((size_t)l * C4 + g4) * 8 + 2 * k];
            b += red[((size_t)l * C4 + g4) * 8 + 2 * k + 1];
        }
        double* out = partial + ((size_t)(g4 * 4 + k) * R + blockIdx.x) * 2;   // [channel][slice][2]: bn_combine_kernel's layout
        out[0] = a;
        out[1] = b;
    }
}

// one thread per (output pixel, 4 channels); POOL: the output pixel is the average of a 2x2 window of activated inputs
template <bool POOL>
__global__ __launch_bounds__(NB_THREADS) void bn_nhwc_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ mean,
                                                                    const float4* __restrict__ scale, const float4* __restrict__ bias,
                                                                    int B, int H, int W, int C4, int relu, float4* __restrict__ y) {
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const size_t pix = idx / C4;
        const float4 m = mean[c4], s = scale[c4], b = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float lo = relu ? 0.f : -INFINITY;
        auto norm = [&](const float4 v) {
            float4 r;
            r.x = fmaxf(fmaf(v.x - m.x, s.x, b.x), lo); r.y = fmaxf(fmaf(v.y - m.y, s.y, b.y), lo);
            r.z = fmaxf(fmaf(v.z - m.z, s.z, b.z), lo); r.w = fmaxf(fmaf(v.w - m.w, s.w, b.w), lo);
            return r;
        };
        if constexpr (!POOL) {
            y[idx] = norm(x[idx]);
        } else {
            const int xo = (int)(pix % Wo);
            const int yo = (int)((pix / Wo) % Ho);
            const size_t bb = pix / ((size_t)Wo * Ho);
            const size_t base = ((bb * H + 2 * yo) * W + 2 * xo) * C4 + c4;
            const float4 a0 = norm(x[base]), a1 = norm(x[base + C4]), a2 = norm(x[base + (size_t)W * C4]),
                         a3 = norm(x[base + (size_t)W * C4 + C4]);
            float4 r;   // F.avg_pool2d: sum of the window, times 1/4
            r.x = 0.25f * ((a0.x + a1.x) + (a2.x + a3.x)); r.y = 0.25f * ((a0.y + a1.y) + (a2.y + a3.y));
            r.z = 0.25f * ((a0.z + a1.z) + (a2.z + a3.z)); r.w = 0.25f * ((a0.w + a1.w) + (a2.w + a3.w));
            y[idx] = r;
        }
    }
}

// ---- backward of  y = [avgpool2x2](act((x - mean) * scale + bias))  on NHWC (the fused block tail of SameBlock2d / UpBlock2d /
// DownBlock2d / the pre-activations of ResBlock2d, reference modules/util.py:858-938, in training mode) ---------------------
// g = d loss / d (BatchNorm output) = dy (a quarter of the pooled pixel's dy when POOL) where the activation passed, else 0 --
// the ReLU mask is recomputed from x, nothing but x is kept from the forward.  Then exactly batchnorm.hip's backward:
// bn_nhwc_bwd_partial_kernel -> (sum g, sum g * (x - mean)) per (channel, slice) in bn_combine_kernel's layout,
// bn_nhwc_bwd_apply_kernel   -> dx = coef_w * (g - S1/N - (x - mean) * S2 inv_std^2 / N)  with bn_bwd_finalize_kernel's coef.
template <bool POOL>
__device__ __forceinline__ float4 bn_nhwc_upstream(const float4* __restrict__ dy, const float4 v, const float4 m, const float4 s,
                                                   const float4 b, int relu, long long row, int c4, int C4, int H, int W) {
    float4 g;
    if constexpr (POOL) {
        const int x = (int)(row % W), y = (int)((row / W) % H);
        const long long bb = row / ((long long)W * H);
        g = dy[((bb * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C4 + c4];
        g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
    } else {
        g = dy[row * C4 + c4];
    }
    if (relu) {
        if (!(fmaf(v.x - m.x, s.x, b.x) > 0.f)) g.x = 0.f;
        if (!(fmaf(v.y - m.y, s.y, b.y) > 0.f)) g.y = 0.f;
        if (!(fmaf(v.z - m.z, s.z, b.z) > 0.f)) g.z = 0.f;
        if (!(fmaf(v.w - m.w, s.w, b.w) > 0.f)) g.w = 0.f;
    }
    return g;
}

template <bool POOL>
__global__ __launch_bounds__(NB_THREADS) void bn_nhwc_bwd_partial_kernel(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                                          const float4* __restrict__ mean, const float4* __restrict__ scale,
                                                                          const float4* __restrict__ bias, long long M, int C4, int H, int W,
                                                                          int relu, int R, double* __restrict__ partial) {
    extern __shared__ double red[];   // [row lanes][C4][8]
    const int c4 = threadIdx.x % C4, lane_r = threadIdx.x / C4, lanes = NB_THREADS / C4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (lane_r < lanes) {
        const float4 m = mean[c4], sc = scale[c4], b = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (long long r = (long long)blockIdx.x * lanes + lane_r; r < M; r += (long long)R * lanes) {
            const float4 v = x[r * C4 + c4];
            const float4 g = bn_nhwc_upstream<POOL>(dy, v, m, sc, b, relu, r, c4, C4, H, W);
            s[0] += (double)g.x; s[1] += (double)g.y; s[2] += (double)g.z; s[3] += (double)g.w;
            q[0] += (double)g.x * (v.x - m.x); q[1] += (double)g.y * (v.y - m.y);
            q[2] += (double)g.z * (v.z - m.z); q[3] += (double)g.w * (v.w - m.w);
        }
        double* dst = red + ((size_t)lane_r * C4 + c4) * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[2 * k] = s[k];
            dst[2 * k + 1] = q[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C4 * 4; i += NB_THREADS) {
        const int g4 = i >> 2, k = i & 3;
        double a = 0.0, b = 0.0;
        for (int l = 0; l < lanes; ++l) {
            a += red[((size_t)l * C4 + g4) * 8 + 2 * k];
            b += red[((size_t)l * C4 + g4) * 8 + 2 * k + 1];
        }
        double* out = partial + ((size_t)(g4 * 4 + k) * R + blockIdx.x) * 2;
        out[0] = a;
        out[1] = b;
    }
}

template <bool POOL>
__global__ __launch_bounds__(NB_THREADS) void bn_nhwc_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                                        const float4* __restrict__ mean, const float4* __restrict__ scale,
                                                                        const float4* __restrict__ bias, const float* __restrict__ coef,
                                                                        long long M, int C4, int H, int W, int relu, float4* __restrict__ dx) {
    const size_t total = (size_t)M * C4;
    const int C = C4 * 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const long long row = (long long)(idx / C4);
        const float4 m = mean[c4], sc = scale[c4], b = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = reinterpret_cast<const float4*>(coef)[c4], bb = reinterpret_cast<const float4*>(coef + C)[c4],
                     cw = reinterpret_cast<const float4*>(coef + 2 * C)[c4];
        const float4 v = x[idx];
        float4 g = bn_nhwc_upstream<POOL>(dy, v, m, sc, b, relu, row, c4, C4, H, W);
        g.x = cw.x * (g.x - a.x - (v.x - m.x) * bb.x); g.y = cw.y * (g.y - a.y - (v.y - m.y) * bb.y);
        g.z = cw.z * (g.z - a.z - (v.z - m.z) * bb.z); g.w = cw.w * (g.w - a.w - (v.w - m.w) * bb.w);
        dx[idx] = g;
    }
}

// slices of the row range: enough blocks to fill the chip, each with >= ~16 rows per row lane (64 until the combine kernel
// stopped walking the slices serially: 256 channels x 32768 rows were 128 workgroups, 1.6 TB/s)
static int nhwc_slices(long long M, int C4) {
    const int lanes = std::max(1, NB_THREADS / C4);
    const long long want = std::max<long long>(1, M / ((long long)lanes * 16));
    return (int)std::min<long long>(1024, want);
}

size_t bn_nhwc_workspace_floats(long long M, int C) {
    if (C < 4 || (C & 3) || C / 4 > NB_THREADS) return 0;
    return (size_t)C * nhwc_slices(M, C / 4) * 2 * 2;   // doubles as floats
}

hipError_t bn_nhwc_sums_launch(const float* x, long long M, int C, float* sums, float* workspace, hipStream_t s, const BnFuse* fuse) {
    if (C < 4 || (C & 3) || C / 4 > NB_THREADS || M < 1) return hipErrorInvalidValue;
    const int C4 = C / 4, R = nhwc_slices(M, C4), lanes = NB_THREADS / C4;
    double* part = reinterpret_cast<double*>(workspace);
    const size_t lds = sizeof(double) * (size_t)lanes * C4 * 8;
    hipLaunchKernelGGL(bn_nhwc_partial_kernel, dim3(R), dim3(NB_THREADS), lds, s, reinterpret_cast<const float4*>(x), M, C4, R, part);
    return bn_combine_launch(part, C, R, M, sums, s, fuse);
}

hipError_t bn_nhwc_apply_launch(const float* x, const float* mean, const float* scale, const float* bias, int B, int H, int W, int C,
                                int relu, int pool, float* y, hipStream_t s) {
    if (C < 4 || (C & 3) || (pool && ((H | W) & 1))) return hipErrorInvalidValue;
    const size_t total = (size_t)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / 4);
    const int blocks = (int)std::min<size_t>((total + NB_THREADS - 1) / NB_THREADS, (size_t)1 << 20);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NB_THREADS), 0, s, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(scale),
                           reinterpret_cast<const float4*>(bias), B, H, W, C / 4, relu, reinterpret_cast<float4*>(y));
    };
    if (pool) launch(bn_nhwc_apply_kernel<true>);
    else launch(bn_nhwc_apply_kernel<false>);
    return hipGetLastError();
}

hipError_t bn_nhwc_bwd_sums_launch(const float* x, const float* dy, const float* mean, const float* scale, const float* bias, int B, int H,
                                   int W, int C, int relu, int pool, float* sums, float* workspace, hipStream_t s, const BnFuse* fuse) {
    const long long M = (long long)B * H * W;
    if (C < 4 || (C & 3) || C / 4 > NB_THREADS || M < 1 || (pool && ((H | W) & 1))) return hipErrorInvalidValue;
    const int C4 = C / 4, R = nhwc_slices(M, C4), lanes = NB_THREADS / C4;
    double* part = reinterpret_cast<double*>(workspace);
    const size_t lds = sizeof(double) * (size_t)lanes * C4 * 8;
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(R), dim3(NB_THREADS), lds, s, reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(dy),
                           reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(scale),
                           reinterpret_cast<const float4*>(bias), M, C4, H, W, relu, R, part);
    };
    if (pool) launch(bn_nhwc_bwd_partial_kernel<true>);
    else launch(bn_nhwc_bwd_partial_kernel<false>);
    return bn_combine_launch(part, C, R, M, sums, s, fuse);
}

hipError_t bn_nhwc_bwd_apply_launch(const float* x, const float* dy, const float* mean, const float* scale, const float* bias,
                                    const float* coef, int B, int H, int W, int C, int relu, int pool, float* dx, hipStream_t s) {
    const long long M = (long long)B * H * W;
    if (C < 4 || (C & 3) || M < 1 || (pool && ((H | W) & 1))) return hipErrorInvalidValue;
    const size_t total = (size_t)M * (C / 4);
    const int blocks = (int)std::min<size_t>((total + NB_THREADS - 1) / NB_THREADS, (size_t)1 << 20);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NB_THREADS), 0, s, reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(dy),
                           reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(scale),
                           reinterpret_cast<const float4*>(bias), coef, M, C / 4, H, W, relu, reinterpret_cast<float4*>(dx));
    };
    if (pool) launch(bn_nhwc_bwd_apply_kernel<true>);
    else launch(bn_nhwc_bwd_apply_kernel<false>);
    return hipGetLastError();
}

}  // namespace eamm
