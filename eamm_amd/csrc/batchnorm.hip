// Training-mode BatchNorm2d on NCHW fp32 tensors -- SURVEY.md section 8f row N4: forward with running statistics (first
// slice, round 2) and backward (round 3: bn_bwd_* below, the standard synchronised-BatchNorm gradient).
// Reference: sync_batchnorm/batchnorm.py:46-125 (`_SynchronizedBatchNorm.forward`: per-channel sum and sum of squares -> reduce over the replicas -> mean / inverse standard deviation + running
// statistics on the master (`_compute_mean_std`) -> (x - mean) * (inv_std * weight) + bias).
//
// Four HBM-bound kernels; the all-reduce of the 2C (+2) per-channel sums sits between the second and the third and is
// the caller's (torch.distributed / RCCL):
//   bn_partial_sums_kernel  x read once (16-byte loads), one (sum, sum of squares) pair per (channel, slice) block
//   bn_combine_kernel       slices added in a fixed order -> sums[2C], element count in sums[2C], sums[2C+1]
//   bn_finalize_kernel      mean, inverse standard deviation, running-statistics update, per-channel scale
//   bn_apply_kernel         x read once more, y written once
// Algorithmic bytes: 12 B per element (two reads, one write); the statistics are 8 B per channel.
#include "kernels.h"

#include <algorithm>

namespace eamm {

namespace {
constexpr int BN_THREADS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
}  // namespace

// grid (C, S, R): block (c, s, r) covers images n = s, s + S, ... and the r-th of R slices of their HW plane.
// VEC = 4 needs HW % 4 == 0 (planes then start 16-byte aligned whenever x does).
template <int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_sums_kernel(const float* __restrict__ x, int N, int C, int HW, int S,
                                                                      int R, double* __restrict__ partial) {
    const int c = blockIdx.x, s = blockIdx.y, r = blockIdx.z;
    const int per = (HW / VEC + R - 1) / R;                 // VEC-wide elements per slice
    const int lo = r * per, hi = min(HW / VEC, lo + per);
    // double accumulators: the kernel is HBM-bound (the fp64 adds ride along), and sum / sum of squares in float lose
    // the variance of a channel whose mean dwarfs its spread -- ATen's single-replica statistics (mode 1) do not
    double sum = 0.0, ssum = 0.0;
    for (int n = s; n < N; n += S) {
        const float* plane = x + ((size_t)n * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += BN_THREADS) {
            if constexpr (VEC == 4) {
                const float4 v = reinterpret_cast<const float4*>(plane)[i];
                sum += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                ssum += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
            } else {
                const float v = plane[i];
                sum += (double)v;
                ssum += (double)v * v;
            }
        }
    }
    __shared__ double red[2][BN_THREADS / 64];
    sum = wave_sum(sum);
    ssum = wave_sum(ssum);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sum;
        red[1][threadIdx.x >> 6] = ssum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < BN_THREADS / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        double* dst = partial + ((size_t)c * (S * R) + (s * R + r)) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// sums[c] = sum, sums[C + c] = sum of squares as floats (what the replicas exchange, as the reference does);
// sums[2C] + 4096 * sums[2C + 1] = element count per channel (two exact floats, so that the count rides through the same
// float all-reduce as the sums); behind them, at float index 2C + 2, the same 2C totals in double for the
// single-replica statistics
// One wave per channel: lane l adds slices l, l + 64, ... and the lanes are folded in a fixed tree -- a fixed order, and no
// thread walks the (up to 1024) slices of a channel through one dependent chain (23 us average, 108 us at worst, per call
// when a thread did: 1.2 ms of a training step's 324 calls).
__device__ __forceinline__ void bn_finalize_channel(int c, float sum, float ssum, double sum_d, double ssum_d, float size, float eps,
                                                    float momentum, int mode, const float* __restrict__ weight,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ mean_out, float* __restrict__ scale_out,
                                                    float* __restrict__ inv_std_out);
__device__ __forceinline__ void bn_bwd_finalize_channel(int c, int C, double local_s1, double local_s2, double s1, double s2, double size,
                                                        const float* __restrict__ inv_std, const float* __restrict__ weight, float eps,
                                                        int mode, float* __restrict__ dweight, float* __restrict__ dbias,
                                                        float* __restrict__ coef);

__global__ __launch_bounds__(64) void bn_combine_kernel(const double* __restrict__ partial, int C, int P, long long count,
                                                        float* __restrict__ sums, BnFuse fuse) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int p = lane; p < P; p += 64) {
        a += partial[((size_t)c * P + p) * 2];
        b += partial[((size_t)c * P + p) * 2 + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        sums[c] = (float)a;
        sums[C + c] = (float)b;
        double* exact = reinterpret_cast<double*>(sums + 2 * C + 2);
        exact[c] = a;
        exact[C + c] = b;
        const float lo = (float)(count % 4096), hi = (float)(count / 4096);
        if (c == 0) {
            sums[2 * C] = lo;
            sums[2 * C + 1] = hi;
        }
        // One replica: nothing is exchanged between the sums and their finalize step, and that step reads only this channel's
        // totals -- run it here (the values bn_finalize_kernel / bn_bwd_finalize_kernel would read back: bit-identical)
        if (fuse.kind == 1) {
            bn_finalize_channel(c, (float)a, (float)b, a, b, lo + 4096.f * hi, fuse.eps, fuse.momentum, fuse.mode, fuse.weight,
                                fuse.running_mean, fuse.running_var, fuse.mean, fuse.scale, fuse.inv_std_out);
        } else if (fuse.kind == 2) {
            const double size = (double)lo + 4096.0 * (double)hi;
            bn_bwd_finalize_channel(c, C, a, b, fuse.mode == 1 ? a : (double)(float)a, fuse.mode == 1 ? b : (double)(float)b, size,
                                    fuse.inv_std, fuse.weight, fuse.eps, fuse.mode, fuse.dweight, fuse.dbias, fuse.coef);
        }
    }
}

// mode 0: the replicas' path (batchnorm.py:110-125): inv_std = clamp(biased var, eps) ^ -0.5
// mode 1: the single-replica path F.batch_norm(training=True) (batchnorm.py:48-53): inv_std = 1 / sqrt(biased var + eps)
// mode 2: evaluation: running statistics, nothing updated
// one channel of bn_finalize_kernel (modes 0 / 1) from its totals: the float pair the replicas exchange and the double pair behind it
__device__ __forceinline__ void bn_finalize_channel(int c, float sum, float ssum, double sum_d, double ssum_d, float size, float eps,
                                                    float momentum, int mode, const float* __restrict__ weight,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ mean_out, float* __restrict__ scale_out,
                                                    float* __restrict__ inv_std_out) {
    const float w = weight != nullptr ? weight[c] : 1.f;
    float mean, unbias_var, inv_std;
    if (mode == 0) {   // the reference's float32 operations, in its order (batchnorm.py:113-125)
        mean = sum / size;
        const float sumvar = ssum - sum * mean;
        unbias_var = sumvar / (size - 1.f);
        const float bias_var = sumvar / size;
        inv_std = powf(fmaxf(bias_var, eps), -0.5f);
    } else {           // ATen accumulates the batch statistics of a float tensor in double: use the double totals
        const double m = sum_d / (double)size;
        const double sumvar = ssum_d - sum_d * m;
        mean = (float)m;
        unbias_var = (float)(sumvar / ((double)size - 1.0));
        inv_std = (float)(1.0 / sqrt(sumvar / (double)size + (double)eps));
    }
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbias_var;
    mean_out[c] = mean;
    scale_out[c] = inv_std * w;
    if (inv_std_out) inv_std_out[c] = inv_std;
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, float eps, float momentum, int mode,
                                   const float* __restrict__ weight, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean_out, float* __restrict__ scale_out,
                                   float* __restrict__ inv_std_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (mode == 2) {
        const float w = weight != nullptr ? weight[c] : 1.f;
        mean_out[c] = running_mean[c];
        scale_out[c] = w / sqrtf(running_var[c] + eps);
        if (inv_std_out) inv_std_out[c] = 1.f / sqrtf(running_var[c] + eps);
        return;
    }
    const float size = sums[2 * C] + 4096.f * sums[2 * C + 1];
    const double* exact = reinterpret_cast<const double*>(sums + 2 * C + 2);
    bn_finalize_channel(c, sums[c], sums[C + c], exact[c], exact[C + c], size, eps, momentum, mode, weight, running_mean, running_var,
                        mean_out, scale_out, inv_std_out);
}

// y = (x - mean[c]) * scale[c] + bias[c]     (batchnorm.py:74-79); grid (N*C planes, slices)
template <int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ scale, const float* __restrict__ bias,
                                                               int C, int HW, float* __restrict__ y) {
    const int plane = blockIdx.x, c = plane % C;
    const float m = mean[c], s = scale[c], b = bias != nullptr ? bias[c] : 0.f;
    const float* src = x + (size_t)plane * HW;
    float* dst = y + (size_t)plane * HW;
    for (int i = blockIdx.y * BN_THREADS + threadIdx.x; i < HW / VEC; i += gridDim.y * BN_THREADS) {
        if constexpr (VEC == 4) {
            float4 v = reinterpret_cast<const float4*>(src)[i];
            v.x = fmaf(v.x - m, s, b); v.y = fmaf(v.y - m, s, b);
            v.z = fmaf(v.z - m, s, b); v.w = fmaf(v.w - m, s, b);
            reinterpret_cast<float4*>(dst)[i] = v;
        } else {
            dst[i] = fmaf(src[i] - m, s, b);
        }
    }
}

// ---- backward (round 3) ------------------------------------------------------------------------------------
// With xhat = (x - mean) * inv_std and y = xhat * weight + bias over the N_total elements per channel of ALL replicas
// (reference: autograd through sync_batchnorm/batchnorm.py:61-79, 110-125, whose ReduceAddCoalesced / Broadcast carry the
// gradient across the replicas):
//   dbias = sum dy,  dweight = sum dy * xhat                      (this replica's shard; the replicas' sums are the caller's)
//   dx = weight * inv_std * (dy - S1 / N_total - xhat * S2 / N_total),  S1 = sum_all dy,  S2 = sum_all dy * xhat
// -- the same expression for both forms of inv_std ((var + eps)^-1/2 and clamp(var, eps)^-1/2: d inv_std / d var =
// -inv_std^3 / 2 in both), except that a CLAMPED channel (replicas' form, var <= eps) has no gradient through the variance.
// bn_bwd_partial_kernel: (sum dy, sum dy * (x - mean)) per (channel, slice), double accumulators -- the forward's layout, so
// bn_combine_kernel packs them the same way (2C floats + count for the all-reduce, 2C doubles behind).
template <int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     const float* __restrict__ mean, int N, int C, int HW, int S,
                                                                     int R, double* __restrict__ partial) {
    const int c = blockIdx.x, s = blockIdx.y, r = blockIdx.z;
    const int per = (HW / VEC + R - 1) / R;
    const int lo = r * per, hi = min(HW / VEC, lo + per);
    const float m = mean[c];
    double sa = 0.0, sb = 0.0;
    for (int n = s; n < N; n += S) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += BN_THREADS) {
            if constexpr (VEC == 4) {
                const float4 v = reinterpret_cast<const float4*>(x + base)[i];
                const float4 g = reinterpret_cast<const float4*>(dy + base)[i];
                sa += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
                sb += ((double)g.x * (v.x - m) + (double)g.y * (v.y - m)) + ((double)g.z * (v.z - m) + (double)g.w * (v.w - m));
            } else {
                const float g = dy[base + i];
                sa += (double)g;
                sb += (double)g * (x[base + i] - m);
            }
        }
    }
    __shared__ double red[2][BN_THREADS / 64];
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sa;
        red[1][threadIdx.x >> 6] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < BN_THREADS / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        double* dst = partial + ((size_t)c * (S * R) + (s * R + r)) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// local: this replica's packed sums (dweight / dbias); reduced: the same buffer after the caller's all-reduce of its first
// 2C + 2 floats (may be the same pointer on one replica).  coef[c], coef[C + c], coef[2C + c] = S1 / N, S2 * inv_std^2 / N,
// weight * inv_std for bn_bwd_apply_kernel.  mode as in bn_finalize_kernel (2 = evaluation: statistics are constants).
__device__ __forceinline__ void bn_bwd_finalize_channel(int c, int C, double local_s1, double local_s2, double s1, double s2, double size,
                                                        const float* __restrict__ inv_std, const float* __restrict__ weight, float eps,
                                                        int mode, float* __restrict__ dweight, float* __restrict__ dbias,
                                                        float* __restrict__ coef) {
    const double is = (double)inv_std[c];
    if (dbias) dbias[c] = (float)local_s1;
    if (dweight) dweight[c] = (float)(local_s2 * is);
    const float w = weight != nullptr ? weight[c] : 1.f;
    double a = 0.0, b = 0.0;
    if (mode != 2) {
        const bool clamped = mode == 0 && inv_std[c] >= powf(eps, -0.5f);   // clamp(var, eps): no gradient through the variance
        a = s1 / size;
        b = clamped ? 0.0 : s2 * is * is / size;
    }
    coef[c] = (float)a;
    coef[C + c] = (float)b;
    coef[2 * C + c] = (float)(is * (double)w);
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ local, const float* __restrict__ reduced, int C,
                                       const float* __restrict__ inv_std, const float* __restrict__ weight, float eps, int mode,
                                       float* __restrict__ dweight, float* __restrict__ dbias, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* lex = reinterpret_cast<const double*>(local + 2 * C + 2);
    double s1 = 0.0, s2 = 0.0, size = 1.0;
    if (mode != 2) {
        size = (double)reduced[2 * C] + 4096.0 * (double)reduced[2 * C + 1];
        // one replica: the double totals (nothing was exchanged); several: the all-reduced floats, as the forward does
        const double* rex = reinterpret_cast<const double*>(reduced + 2 * C + 2);
        s1 = mode == 1 ? rex[c] : (double)reduced[c];
        s2 = mode == 1 ? rex[C + c] : (double)reduced[C + c];
    }
    bn_bwd_finalize_channel(c, C, lex[c], lex[C + c], s1, s2, size, inv_std, weight, eps, mode, dweight, dbias, coef);
}

template <int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ mean, const float* __restrict__ coef,
                                                                   int C, int HW, float* __restrict__ dx) {
    const int plane = blockIdx.x, c = plane % C;
    const float m = mean[c], a = coef[c], b = coef[C + c], sc = coef[2 * C + c];
    const size_t base = (size_t)plane * HW;
    for (int i = blockIdx.y * BN_THREADS + threadIdx.x; i < HW / VEC; i += gridDim.y * BN_THREADS) {
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x + base)[i];
            float4 g = reinterpret_cast<const float4*>(dy + base)[i];
            g.x = sc * (g.x - a - (v.x - m) * b); g.y = sc * (g.y - a - (v.y - m) * b);
            g.z = sc * (g.z - a - (v.z - m) * b); g.w = sc * (g.w - a - (v.w - m) * b);
            reinterpret_cast<float4*>(dx + base)[i] = g;
        } else {
            dx[base + i] = sc * (dy[base + i] - a - (x[base + i] - m) * b);
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------
void bn_plan(int N, int C, int HW, int* S, int* R) {
    // >= ~2048 blocks when the tensor allows it, each with at least ~4 K elements
    const long long per_channel = (long long)N * HW;
    int want = (int)std::max<long long>(1, std::min<long long>(2048 / std::max(1, C) + 1, per_channel / 4096));
    *S = std::max(1, std::min(N, want));
    *R = std::max(1, std::min((want + *S - 1) / *S, std::max(1, HW / 1024)));
}

size_t bn_workspace_floats(int N, int C, int HW) {
    int S, R;
    bn_plan(N, C, HW, &S, &R);
    return (size_t)C * S * R * 2 * 2;   // (sum, sum of squares) in double per (channel, slice)
}

hipError_t bn_local_sums_launch(const float* x, int N, int C, int HW, float* sums, float* workspace, hipStream_t s) {
    int S, R;
    bn_plan(N, C, HW, &S, &R);
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    double* part = reinterpret_cast<double*>(workspace);   // 8-byte aligned: the caller's allocation is
    if (vec)
        hipLaunchKernelGGL(bn_partial_sums_kernel<4>, dim3(C, S, R), dim3(BN_THREADS), 0, s, x, N, C, HW, S, R, part);
    else
        hipLaunchKernelGGL(bn_partial_sums_kernel<1>, dim3(C, S, R), dim3(BN_THREADS), 0, s, x, N, C, HW, S, R, part);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(C), dim3(64), 0, s, part, C, S * R, (long long)N * HW, sums, BnFuse{});
    return hipGetLastError();
}

hipError_t bn_combine_launch(const double* partial, int C, int P, long long count, float* sums, hipStream_t s, const BnFuse* fuse) {
    hipLaunchKernelGGL(bn_combine_kernel, dim3(C), dim3(64), 0, s, partial, C, P, count, sums, fuse ? *fuse : BnFuse{});
    return hipGetLastError();
}

hipError_t bn_finalize_launch(const float* sums, int C, float eps, float momentum, int mode, const float* weight,
                              float* running_mean, float* running_var, float* mean, float* scale, float* inv_std, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, eps, momentum, mode, weight,
                       running_mean, running_var, mean, scale, inv_std);
    return hipGetLastError();
}

hipError_t bn_apply_launch(const float* x, const float* mean, const float* scale, const float* bias, int N, int C, int HW,
                           float* y, hipStream_t s) {
    const bool vec = (HW % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    const int per = vec ? HW / 4 : HW;
    const int slices = std::max(1, std::min(64, (per + 4 * BN_THREADS - 1) / (4 * BN_THREADS)));
    if (vec)
        hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(N * C, slices), dim3(BN_THREADS), 0, s, x, mean, scale, bias, C, HW, y);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(N * C, slices), dim3(BN_THREADS), 0, s, x, mean, scale, bias, C, HW, y);
    return hipGetLastError();
}

hipError_t bn_bwd_sums_launch(const float* x, const float* dy, const float* mean, int N, int C, int HW, float* sums, float* workspace,
                              hipStream_t s) {
    int S, R;
    bn_plan(N, C, HW, &S, &R);
    const bool vec = (HW % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0);
    double* part = reinterpret_cast<double*>(workspace);
    if (vec)
        hipLaunchKernelGGL(bn_bwd_partial_kernel<4>, dim3(C, S, R), dim3(BN_THREADS), 0, s, x, dy, mean, N, C, HW, S, R, part);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel<1>, dim3(C, S, R), dim3(BN_THREADS), 0, s, x, dy, mean, N, C, HW, S, R, part);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(C), dim3(64), 0, s, part, C, S * R, (long long)N * HW, sums, BnFuse{});
    return hipGetLastError();
}

hipError_t bn_bwd_finalize_launch(const float* local, const float* reduced, int C, const float* inv_std, const float* weight, float eps,
                                  int mode, float* dweight, float* dbias, float* coef, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, local, reduced, C, inv_std, weight, eps, mode,
                       dweight, dbias, coef);
    return hipGetLastError();
}

hipError_t bn_bwd_apply_launch(const float* x, const float* dy, const float* mean, const float* coef, int N, int C, int HW, float* dx,
                               hipStream_t s) {
    const bool vec = (HW % 4 == 0) &&
                     (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
    const int per = vec ? HW / 4 : HW;
    const int slices = std::max(1, std::min(64, (per + 4 * BN_THREADS - 1) / (4 * BN_THREADS)));
    if (vec)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(N * C, slices), dim3(BN_THREADS), 0, s, x, dy, mean, coef, C, HW, dx);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(N * C, slices), dim3(BN_THREADS), 0, s, x, dy, mean, coef, C, HW, dx);
    return hipGetLastError();
}

}  // namespace eamm
