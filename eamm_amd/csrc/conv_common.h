// Device helpers shared by the convolution kernels (register-staged and LDS-DMA variants).
#pragma once
#include "kernels.h"

#include <atomic>
#include <type_traits>

namespace eamm {

// Opt a kernel into > 64 KiB of dynamic LDS once per (kernel instantiation, device): the attribute belongs to the
// device's copy of the function, and one process may drive several GPUs (one handle per device).
// The mask is shared by every host thread that launches the kernel (one thread per device): a real atomic.
typedef std::atomic<unsigned long long> lds_once_mask;
template <typename K>
inline hipError_t ensure_dynamic_lds(K kern, size_t bytes, lds_once_mask* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask->load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done_mask->fetch_or(bit, std::memory_order_release);
    return e;
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  Used wherever an
// accumulator array is indexed, so that no index is ever a run-time value (a failed "#pragma unroll"
// silently demotes the whole array to scratch memory).
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

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

// pixel m (2x2-quad order) -> (b, y, x)
__device__ __forceinline__ void quad_decode(int m, int Hq, int Wq, int& b, int& y, int& x) {
    const int q = m >> 2, jj = m & 3;
    const int qx = q % Wq, t = q / Wq;
    const int qy = t % Hq;
    b = t / Hq;
    y = 2 * qy + (jj >> 1);
    x = 2 * qx + (jj & 1);
}

// pixel m -> (b, y, x) in the launch's enumeration: 2x2 quads (pooling windows / Winograd tiles) or raster
__device__ __forceinline__ void pix_decode(const ConvArgs& p, int m, int& b, int& y, int& x) {
    if (p.linear) {
        x = m % p.W;
        const int t = m / p.W;
        y = t % p.H;
        b = t / p.H;
    } else {
        quad_decode(m, p.H >> 1, p.W >> 1, b, y, x);
    }
}

// one output element through the epilogue (shared by the fused path and the split-K reduction)
__device__ __forceinline__ void epilogue_store(const ConvArgs& p, int phase, int b, int y, int x, int n, float v,
                                               float s2, float t2) {
    int oy = y, ox = x, OH = p.H, OW = p.W;
    if (p.nphase == 4) {
        oy = 2 * y + (phase >> 1);
        ox = 2 * x + (phase & 1);
        OH = 2 * p.H;
        OW = 2 * p.W;
    }
    const size_t pix = (size_t)(b * OH + oy) * OW + ox;
    if (p.split_n > 0) {   // wide channels NHWC [.., split_n] + the last few as one float4 per pixel (no residual / second output here)
        v = apply_act(v, p.act);
        if (n < p.split_n) p.out[pix * p.split_n + n] = v;
        else p.out2[pix * 4 + (n - p.split_n)] = v;
        return;
    }
    if (p.resid != nullptr) v += p.resid[pix * p.Cout + n];
    v = apply_act(v, p.act);
    if (p.nchw)
        p.out[((size_t)(b * p.Cout + n) * OH + oy) * OW + ox] = v;
    else
        p.out[pix * p.Cout + n] = v;
    if (p.out2 != nullptr) p.out2[pix * p.Cout + n] = fmaxf(fmaf(v, s2, t2), 0.f);
}


// Coalesced epilogue for the big-tile kernels: each pass stages the rows {wm*MT*32 + i*32 + 0..31} of every
// wave through LDS ([WM*32][BN+4] floats; the main-loop stages are dead by then), then the whole workgroup
// streams the tile out as 16-byte accesses along the channel dimension -- one wave instruction covers a full
// 1 KiB row (BN = 256) instead of two 128-byte fragments of a 4-byte-per-lane store.  Handles bias, residual
// (float4 loads), activation and the second pre-activated output; pooled / NCHW / split-K outputs and
// channel counts that are not a multiple of 4 use conv_epilogue below.
template <int MT, int NT, int WM, int WN, typename Acc>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& p, Acc& acc, float* smem, int mbase, int ntile,
                                                  int wm, int wn, int l31, int half, int phase, int tid) {
    constexpr int BN = WN * NT * 32, R = WM * 32, LDO = BN + 4, NTHR = WM * WN * 64;
    constexpr int C4 = BN / 4, PER = R * C4 / NTHR;
    static_assert((R * C4) % NTHR == 0, "tile must split evenly over the threads");
    const int OH = p.nphase == 4 ? 2 * p.H : p.H, OW = p.nphase == 4 ? 2 * p.W : p.W;
    static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        __syncthreads();  // previous pass (or the main loop) is done with the LDS
        static_for<NT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int col = wn * NT * 32 + j * 32 + l31;
            const float bias = p.bias[ntile * BN + col];
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                smem[row * LDO + col] = acc[i][j][r] + bias;
            });
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / C4, c4 = idx - row * C4;
            const int m = mbase + (row >> 5) * (MT * 32) + i * 32 + (row & 31);
            const int n = ntile * BN + c4 * 4;
            if (m < p.M && n < p.Cout) {
                float4 v = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                int b, y, x;
                pix_decode(p, m, b, y, x);
                if (p.nphase == 4) {
                    y = 2 * y + (phase >> 1);
                    x = 2 * x + (phase & 1);
                }
                if (p.split_n > 0) {   // split NHWC hand-over (epilogue_store): wide channels with a split_n stride, the last float4 apart
                    const size_t pix = (size_t)(b * OH + y) * OW + x;
                    v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                    v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                    if (n < p.split_n) *reinterpret_cast<float4*>(p.out + pix * p.split_n + n) = v;
                    else *reinterpret_cast<float4*>(p.out2 + pix * 4) = v;
                    continue;
                }
                const size_t o = ((size_t)(b * OH + y) * OW + x) * p.Cout + n;
                if (p.resid != nullptr) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.resid + o);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                *reinterpret_cast<float4*>(p.out + o) = v;
                if (p.out2 != nullptr) {
                    const float4 s2 = *reinterpret_cast<const float4*>(p.s2 + n);
                    const float4 t2 = *reinterpret_cast<const float4*>(p.t2 + n);
                    float4 a;
                    a.x = fmaxf(fmaf(v.x, s2.x, t2.x), 0.f); a.y = fmaxf(fmaf(v.y, s2.y, t2.y), 0.f);
                    a.z = fmaxf(fmaf(v.z, s2.z, t2.z), 0.f); a.w = fmaxf(fmaf(v.w, s2.w, t2.w), 0.f);
                    *reinterpret_cast<float4*>(p.out2 + o) = a;
                }
            }
        }
    });
}

// Epilogue of one block: acc[i][j] is the 32x32 MFMA tile (i, j) of wave (wm, wn).  Either raw split-K
// slabs, or bias + residual + activation (+ in-register 2x2 average pool / second pre-activated output).
template <int MT, int NT, int BN, typename Acc>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, Acc& acc, int mbase, int ntile, int wm, int wn,
                                              int l31, int half, int phase, int split) {
    const int Wq = p.W >> 1, Hq = p.H >> 1;
    if (p.partial != nullptr) {
        float* slab = p.partial + (size_t)(split * p.nphase + phase) * p.Mpad * p.Npad;
        static_for<MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<NT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int n = ntile * BN + wn * NT * 32 + j * 32 + l31;
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int m = mbase + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) slab[(size_t)m * p.Npad + n] = acc[i][j][r];  // rows past M are never reduced
                });
            });
        });
        return;
    }
    static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<NT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int n = ntile * BN + wn * NT * 32 + j * 32 + l31;
            const bool nok = n < p.Cout;
            const float bias = p.bias[n];
            float s2 = 0.f, t2 = 0.f;
            if (p.out2 != nullptr && nok && p.split_n == 0) {
                s2 = p.s2[n];
                t2 = p.t2[n];
            }
            static_for<4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const int m0 = mbase + wm * MT * 32 + i * 32 + 8 * g + 4 * half;  // first pixel of a quad
                if (m0 < p.M && nok) {
                    if (p.pool) {
                        float v = 0.f;
                        static_for<4>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            v += apply_act(acc[i][j][4 * g + e] + bias, p.act);
                        });
                        p.out[(size_t)(m0 >> 2) * p.Cout + n] = 0.25f * v;
                    } else if (p.linear) {
                        static_for<4>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            if (m0 + e < p.M) {
                                int b, y, x;
                                pix_decode(p, m0 + e, b, y, x);
                                epilogue_store(p, phase, b, y, x, n, acc[i][j][4 * g + e] + bias, s2, t2);
                            }
                        });
                    } else {
                        int b, y, x;
                        quad_decode(m0, Hq, Wq, b, y, x);
                        static_for<4>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            epilogue_store(p, phase, b, y + (e >> 1), x + (e & 1), n, acc[i][j][4 * g + e] + bias, s2, t2);
                        });
                    }
                }
            });
        });
    });
}

}  // namespace eamm
