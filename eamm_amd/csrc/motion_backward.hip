// Backward of the dense-motion front end and flow head: the gradients autograd
// derives from the reference's torch ops in modules/dense_motion.py:32-113 and modules/util.py:1005-1052, as HIP kernels next to
// the forward kernels of motion.hip, so that the differentiable generator forward (eamm_amd/train_graph.py) no longer leans on
// rocBLAS / MIOpen / ATen for this part:
//   * antialias_down_backward_kernel  -- AntiAliasInterpolation2d (util.py:1044-1052): the transpose of "zero-pad 6, depth-wise
//     13x13 Gaussian, keep every inv_scale-th row and column": every source pixel gathers the <= 4 x 4 kept outputs it fed.
//   * kp_records_backward_kernel      -- J = J_source inverse(J_driving) (dense_motion.py:55-56) and the key-point values.
//   * motion_front_backward_kernel    -- heat-maps (util.py:815-836, dense_motion.py:32-45), sparse motions T_k (:47-67) and the
//     K+1 bilinear warps of the down-sampled source (:69-79): d hourglass-input / d sparse_deformed -> d records, d source.
//   * motion_head_forward / backward  -- softmax over the K+1 mask logits, deformation = sum_k mask_k T_k, sigmoid occlusion
//     (:98-111) on the two convolutions' NHWC outputs.
// All of it is HBM-bound element-wise / gather work on the 64x64 motion grid (a few MB per batch): one thread per pixel, the
// per-key-point sums reduced per workgroup and then in a FIXED order over the workgroups (the key-point / record gradients are
// deterministic).  NOT deterministic: the gradient of the down-sampled SOURCE image in motion_front_backward_kernel, which is
// accumulated with float atomicAdd over the pixels and the K+1 motions that sampled a source pixel (as ATen's grid_sample
// backward does): it varies in the last bits from run to run; the gradient tests' tolerances cover that.
#include "kernels.h"

#include <algorithm>

namespace eamm {

namespace {

__device__ __forceinline__ float grid_coord_b(int j, int n) { return 2.f * ((float)j / (float)(n - 1)) - 1.f; }

struct BilinearM {
    int x0, y0;
    float ax, ay;
    bool valid;
};
// the forward's bilinear_setup (motion.hip) with the fractions kept for the derivative
__device__ __forceinline__ BilinearM bilinear_setup_m(float gx, float gy, int W, int H) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    BilinearM b;
    b.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W);
    b.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);
    b.ax = ix - fx;
    b.ay = iy - fy;
    b.valid = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    return b;
}

constexpr int MB_THREADS = 256;
constexpr int MB_MAXK = 31;

// sum of v over the workgroup's threads, result valid in thread 0 (fixed tree: deterministic)
__device__ __forceinline__ float block_sum(float v, float* red /*[MB_THREADS / 64]*/) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();                  // `red` may still be read from the previous call
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < MB_THREADS / 64; ++i) s += red[i];
    return s;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// AntiAliasInterpolation2d backward: dsmall [B,h,w,4] (channels 0..2) -> dsrc [B,3,H,W]
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void antialias_down_backward_kernel(const float4* __restrict__ dsmall, const float* __restrict__ aa_w,
                                                                      int B, int H, int W, int inv_scale, float* __restrict__ dsrc) {
    const int h = H / inv_scale, w = W / inv_scale;
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)B * plane;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int X = (int)(idx % W), Y = (int)((idx / W) % H), b = (int)(idx / plane);
    float acc[3] = {0.f, 0.f, 0.f};
    if (inv_scale == 1) {
        const float4 g = dsmall[idx];
        acc[0] = g.x; acc[1] = g.y; acc[2] = g.z;
    } else {
        // forward: small[y][x] = sum_{ky,kx} k[ky][kx] src[s y - 6 + ky][s x - 6 + kx]  =>  ky = Y - s y + 6 in [0, 13)
        const int y_lo = max(0, (Y - 6 + inv_scale - 1) / inv_scale), y_hi = min(h - 1, (Y + 6) / inv_scale);
        const int x_lo = max(0, (X - 6 + inv_scale - 1) / inv_scale), x_hi = min(w - 1, (X + 6) / inv_scale);
        for (int y = y_lo; y <= y_hi; ++y) {
            const int ky = Y - inv_scale * y + 6;
            for (int x = x_lo; x <= x_hi; ++x) {
                const int kx = X - inv_scale * x + 6;
                const float4 g = dsmall[((size_t)b * h + y) * w + x];
                acc[0] = fmaf(g.x, aa_w[0 * 169 + ky * 13 + kx], acc[0]);
                acc[1] = fmaf(g.y, aa_w[1 * 169 + ky * 13 + kx], acc[1]);
                acc[2] = fmaf(g.z, aa_w[2 * 169 + ky * 13 + kx], acc[2]);
            }
        }
    }
    const size_t r = (size_t)Y * W + X;
    dsrc[((size_t)b * 3 + 0) * plane + r] = acc[0];
    dsrc[((size_t)b * 3 + 1) * plane + r] = acc[1];
    dsrc[((size_t)b * 3 + 2) * plane + r] = acc[2];
}

hipError_t antialias_down_backward_launch(const float* dsmall, const float* aa_w, int B, int H, int W, int inv_scale, float* dsrc,
                                          hipStream_t s) {
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(antialias_down_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(dsmall), aa_w, B, H, W, inv_scale, dsrc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// key-point records backward: drec [n,K,8] -> d kd_val, d ks_val [n,K,2], d kd_jac, d ks_jac [n,K,2,2]   (per-pair sources)
//   J = S D^-1:  dS = dJ D^-T,   dD = -D^-T (S^T dJ) D^-T
// ---------------------------------------------------------------------------------------------------------------------
__global__ void kp_records_backward_kernel(const float* __restrict__ kd_jac, const float* __restrict__ ks_jac,
                                           const float* __restrict__ drec, int nk, float* __restrict__ dkd_val,
                                           float* __restrict__ dks_val, float* __restrict__ dkd_jac, float* __restrict__ dks_jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk) return;
    const float* g = drec + (size_t)i * KP_STRIDE;
    if (dkd_val) { dkd_val[i * 2 + 0] = g[0]; dkd_val[i * 2 + 1] = g[1]; }
    if (dks_val) { dks_val[i * 2 + 0] = g[2]; dks_val[i * 2 + 1] = g[3]; }
    if (kd_jac == nullptr || (dkd_jac == nullptr && dks_jac == nullptr)) return;
    const float* d = kd_jac + (size_t)i * 4;
    const float* s = ks_jac + (size_t)i * 4;
    const float det = d[0] * d[3] - d[1] * d[2], inv = 1.f / det;
    const float i00 = d[3] * inv, i01 = -d[1] * inv, i10 = -d[2] * inv, i11 = d[0] * inv;   // D^-1
    const float g00 = g[4], g01 = g[5], g10 = g[6], g11 = g[7];                             // dJ
    if (dks_jac) {   // dS = dJ D^-T : dS[r][c] = sum_m dJ[r][m] Dinv[c][m]
        dks_jac[i * 4 + 0] = g00 * i00 + g01 * i01;
        dks_jac[i * 4 + 1] = g00 * i10 + g01 * i11;
        dks_jac[i * 4 + 2] = g10 * i00 + g11 * i01;
        dks_jac[i * 4 + 3] = g10 * i10 + g11 * i11;
    }
    if (dkd_jac) {   // M = S^T dJ;  dD = -D^-T M D^-T
        const float m00 = s[0] * g00 + s[2] * g10, m01 = s[0] * g01 + s[2] * g11;
        const float m10 = s[1] * g00 + s[3] * g10, m11 = s[1] * g01 + s[3] * g11;
        // P = D^-T M : P[r][c] = sum_m Dinv[m][r] M[m][c]
        const float p00 = i00 * m00 + i10 * m10, p01 = i00 * m01 + i10 * m11;
        const float p10 = i01 * m00 + i11 * m10, p11 = i01 * m01 + i11 * m11;
        // dD = -P D^-T : [r][c] = -sum_m P[r][m] Dinv[c][m]
        dkd_jac[i * 4 + 0] = -(p00 * i00 + p01 * i01);
        dkd_jac[i * 4 + 1] = -(p00 * i10 + p01 * i11);
        dkd_jac[i * 4 + 2] = -(p10 * i00 + p11 * i01);
        dkd_jac[i * 4 + 3] = -(p10 * i10 + p11 * i11);
    }
}

hipError_t kp_records_backward_launch(const float* kd_jac, const float* ks_jac, const float* drec, int n, int K, float* dkd_val,
                                      float* dks_val, float* dkd_jac, float* dks_jac, hipStream_t s) {
    const int nk = n * K;
    hipLaunchKernelGGL(kp_records_backward_kernel, dim3((nk + 127) / 128), dim3(128), 0, s, kd_jac, ks_jac, drec, nk, dkd_val, dks_val,
                       dkd_jac, dks_jac);
    return hipGetLastError();
}

// fixed-order sum of the workgroups' partial record gradients: part [n][blocks][K][8] -> drec [n][K][8] (+= when accumulate)
__global__ void rec_partial_reduce_kernel(const float* __restrict__ part, int n, int blocks, int K, int accumulate,
                                          float* __restrict__ drec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K * KP_STRIDE) return;
    const int f = i / (K * KP_STRIDE), r = i - f * K * KP_STRIDE;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[((size_t)f * blocks + b) * K * KP_STRIDE + r];
    drec[i] = accumulate ? drec[i] + s : s;
}

// ---------------------------------------------------------------------------------------------------------------------
// front end backward.  One thread per motion-grid pixel; for every motion k: the gradient of the warped RGB
// (dhg channels 4k+1..3 + dsd) goes to the four source corners (atomics, as ATen's grid_sampler backward) and, through the
// sampling position, to T_k = J_k (z - kd_k) + ks_k; the heat-map gradient (dhg channel 4k) to kd_k / ks_k.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MB_THREADS) void motion_front_backward_kernel(const float* __restrict__ rec_all,
                                                                           const float4* __restrict__ src_small, int K, int h, int w,
                                                                           float variance, int Cpad, const float* __restrict__ dhg,
                                                                           const float* __restrict__ dsd, float* __restrict__ dsrc_small,
                                                                           float* __restrict__ part) {
    __shared__ float red[MB_THREADS / 64];
    const int f = blockIdx.y;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pi < h * w;
    const int y = live ? pi / w : 0, x = live ? pi - y * w : 0;
    const float gx = grid_coord_b(x, w), gy = grid_coord_b(y, h);
    const float4* src = src_small + (size_t)f * h * w;
    float* dsrc = dsrc_small ? dsrc_small + (size_t)f * h * w * 4 : nullptr;
    const float* rec = rec_all + (size_t)f * K * KP_STRIDE;
    const float* dh = dhg ? dhg + ((size_t)f * h * w + (live ? pi : 0)) * Cpad : nullptr;
    const size_t plane = (size_t)h * w;
    const float* ds = dsd ? dsd + (size_t)f * (K + 1) * 3 * plane + (live ? pi : 0) : nullptr;
    float* out = part + ((size_t)f * gridDim.x + blockIdx.x) * K * KP_STRIDE;
    for (int k = 0; k <= K; ++k) {
        float g_heat = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (live) {
            if (dh) {
                const float4 v = reinterpret_cast<const float4*>(dh)[k];
                g_heat = v.x; g0 = v.y; g1 = v.z; g2 = v.w;
            }
            if (ds) {
                g0 += ds[(size_t)(k * 3 + 0) * plane];
                g1 += ds[(size_t)(k * 3 + 1) * plane];
                g2 += ds[(size_t)(k * 3 + 2) * plane];
            }
        }
        float tx = gx, ty = gy, rx = 0.f, ry = 0.f;
        const float* r = rec + (k > 0 ? k - 1 : 0) * KP_STRIDE;
        if (k > 0) {
            rx = gx - r[0];
            ry = gy - r[1];
            tx = fmaf(r[5], ry, r[4] * rx) + r[2];
            ty = fmaf(r[7], ry, r[6] * rx) + r[3];
        }
        float dtx = 0.f, dty = 0.f;
        if (live) {
            const BilinearM b = bilinear_setup_m(tx, ty, w, h);
            if (b.valid) {
                const float wx[2] = {1.f - b.ax, b.ax}, wy[2] = {1.f - b.ay, b.ay};
                float dix = 0.f, diy = 0.f;
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const int yy = b.y0 + cy, xx = b.x0 + cx;
                        if ((unsigned)yy >= (unsigned)h || (unsigned)xx >= (unsigned)w) continue;   // zeros padding
                        const float4 v = src[yy * w + xx];
                        const float dot = g0 * v.x + g1 * v.y + g2 * v.z;
                        dix = fmaf((cx ? 1.f : -1.f) * wy[cy], dot, dix);
                        diy = fmaf((cy ? 1.f : -1.f) * wx[cx], dot, diy);
                        if (dsrc) {
                            const float wgt = wx[cx] * wy[cy];
                            float* d = dsrc + (size_t)(yy * w + xx) * 4;
                            atomicAdd(d + 0, wgt * g0);
                            atomicAdd(d + 1, wgt * g1);
                            atomicAdd(d + 2, wgt * g2);
                        }
                    }
                dtx = dix * (0.5f * (float)w);     // ix = ((tx + 1) w - 1) / 2
                dty = diy * (0.5f * (float)h);
            }
        }
        if (k == 0) continue;   // the identity grid depends on nothing
        // heat = exp(-|z - kd|^2 / 2v) - exp(-|z - ks|^2 / 2v)
        float dkdx = 0.f, dkdy = 0.f, dksx = 0.f, dksy = 0.f;
        if (live) {
            const float sx = gx - r[2], sy = gy - r[3];
            const float ed = expf(-0.5f * (rx * rx + ry * ry) / variance), es = expf(-0.5f * (sx * sx + sy * sy) / variance);
            dkdx = g_heat * ed * rx / variance;
            dkdy = g_heat * ed * ry / variance;
            dksx = -g_heat * es * sx / variance;
            dksy = -g_heat * es * sy / variance;
            // T = J (z - kd) + ks
            dkdx -= r[4] * dtx + r[6] * dty;
            dkdy -= r[5] * dtx + r[7] * dty;
            dksx += dtx;
            dksy += dty;
        }
        const float vals[8] = {dkdx, dkdy, dksx, dksy, dtx * rx, dtx * ry, dty * rx, dty * ry};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s = block_sum(live ? vals[j] : 0.f, red);
            if (threadIdx.x == 0) out[(k - 1) * KP_STRIDE + j] = s;
        }
    }
}

hipError_t motion_front_backward_launch(const float* rec, const float* src_small, int n, int K, int h, int w, float variance, int Cpad,
                                        const float* dhg, const float* dsd, float* dsrc_small, float* drec, float* workspace,
                                        hipStream_t s) {
    if (K < 1 || K > MB_MAXK || Cpad < 4 * (K + 1) || (Cpad & 3)) return hipErrorInvalidValue;
    const int blocks = (h * w + MB_THREADS - 1) / MB_THREADS;
    if (dsrc_small)
        if (hipError_t e = hipMemsetAsync(dsrc_small, 0, (size_t)n * h * w * 4 * sizeof(float), s); e != hipSuccess) return e;
    hipLaunchKernelGGL(motion_front_backward_kernel, dim3(blocks, n), dim3(MB_THREADS), 0, s, rec, reinterpret_cast<const float4*>(src_small),
                       K, h, w, variance, Cpad, dhg, dsd, dsrc_small, workspace);
    const int tot = n * K * KP_STRIDE;
    hipLaunchKernelGGL(rec_partial_reduce_kernel, dim3((tot + 127) / 128), dim3(128), 0, s, workspace, n, blocks, K, 0, drec);
    return hipGetLastError();
}

size_t motion_backward_workspace_floats(int n, int K, int h, int w) {
    return (size_t)n * ((h * w + MB_THREADS - 1) / MB_THREADS) * K * KP_STRIDE;
}

// ---------------------------------------------------------------------------------------------------------------------
// flow head on the two convolutions' outputs: mask logits [n,h,w,ld] (channels 0..K), occlusion logit [n,h,w,ldo] (channel 0)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void motion_head_forward_kernel(const float* __restrict__ lm, int ld, const float* __restrict__ lo,
                                                                  int ldo, const float* __restrict__ rec_all, int K, int h, int w,
                                                                  float* __restrict__ mask, float* __restrict__ deformation,
                                                                  float* __restrict__ occlusion) {
    const int f = blockIdx.y;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= h * w) return;
    const int y = pi / w, x = pi - y * w;
    const float gx = grid_coord_b(x, w), gy = grid_coord_b(y, h);
    const float* l = lm + ((size_t)f * h * w + pi) * ld;
    const float* rec = rec_all + (size_t)f * K * KP_STRIDE;
    float mx = l[0];
    for (int k = 1; k <= K; ++k) mx = fmaxf(mx, l[k]);
    float sum = 0.f;
    for (int k = 0; k <= K; ++k) sum += expf(l[k] - mx);
    float dx = 0.f, dy = 0.f;
    const size_t plane = (size_t)h * w;
    for (int k = 0; k <= K; ++k) {
        const float mk = expf(l[k] - mx) / sum;
        float tx = gx, ty = gy;
        if (k > 0) {
            const float* r = rec + (k - 1) * KP_STRIDE;
            const float rx = gx - r[0], ry = gy - r[1];
            tx = fmaf(r[5], ry, r[4] * rx) + r[2];
            ty = fmaf(r[7], ry, r[6] * rx) + r[3];
        }
        dx = fmaf(mk, tx, dx);
        dy = fmaf(mk, ty, dy);
        mask[((size_t)f * (K + 1) + k) * plane + pi] = mk;
    }
    reinterpret_cast<float2*>(deformation)[(size_t)f * plane + pi] = make_float2(dx, dy);
    if (lo) occlusion[(size_t)f * plane + pi] = 1.f / (1.f + expf(-lo[((size_t)f * plane + pi) * ldo]));
}

// d mask [n,K+1,h,w], d deformation [n,h,w,2], d occlusion [n,h,w] (any may be null) -> d logits (same layouts as the inputs,
// channels beyond the used ones zeroed), partial d records
__global__ __launch_bounds__(MB_THREADS) void motion_head_backward_kernel(const float* __restrict__ mask, const float* __restrict__ occlusion,
                                                                          const float* __restrict__ rec_all, int K, int h, int w,
                                                                          const float* __restrict__ dmask, const float* __restrict__ ddef,
                                                                          const float* __restrict__ docc, float* __restrict__ dlm, int ld,
                                                                          float* __restrict__ dlo, int ldo, int stacked,
                                                                          float* __restrict__ part) {
    // stacked: the occlusion logit is channel K + 1 of the mask logits' rows (one convolution produced both): its gradient goes
    // to dlm[.., K + 1] and dlo is null
    __shared__ float red[MB_THREADS / 64];
    const int f = blockIdx.y;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pi < h * w;
    const int p = live ? pi : 0;
    const int y = p / w, x = p - y * w;
    const float gx = grid_coord_b(x, w), gy = grid_coord_b(y, h);
    const size_t plane = (size_t)h * w;
    const float* rec = rec_all + (size_t)f * K * KP_STRIDE;
    float2 gd = make_float2(0.f, 0.f);
    if (ddef && live) gd = reinterpret_cast<const float2*>(ddef)[(size_t)f * plane + p];
    // u_k = d mask_k + d def . T_k ;  d logit_k = m_k (u_k - sum_j m_j u_j)
    float dot = 0.f;
    for (int k = 0; k <= K; ++k) {
        const float mk = mask[((size_t)f * (K + 1) + k) * plane + p];
        float tx = gx, ty = gy;
        if (k > 0) {
            const float* r = rec + (k - 1) * KP_STRIDE;
            const float rx = gx - r[0], ry = gy - r[1];
            tx = fmaf(r[5], ry, r[4] * rx) + r[2];
            ty = fmaf(r[7], ry, r[6] * rx) + r[3];
        }
        const float u = (dmask ? dmask[((size_t)f * (K + 1) + k) * plane + p] : 0.f) + gd.x * tx + gd.y * ty;
        dot = fmaf(mk, u, dot);
    }
    float* out = part + ((size_t)f * gridDim.x + blockIdx.x) * K * KP_STRIDE;
    float* dl = dlm + ((size_t)f * plane + p) * ld;
    for (int k = 0; k <= K; ++k) {
        const float mk = mask[((size_t)f * (K + 1) + k) * plane + p];
        float tx = gx, ty = gy, rx = 0.f, ry = 0.f;
        const float* r = rec + (k > 0 ? k - 1 : 0) * KP_STRIDE;
        if (k > 0) {
            rx = gx - r[0];
            ry = gy - r[1];
            tx = fmaf(r[5], ry, r[4] * rx) + r[2];
            ty = fmaf(r[7], ry, r[6] * rx) + r[3];
        }
        const float u = (dmask ? dmask[((size_t)f * (K + 1) + k) * plane + p] : 0.f) + gd.x * tx + gd.y * ty;
        if (live) dl[k] = mk * (u - dot);
        if (k == 0) continue;
        const float dtx = live ? mk * gd.x : 0.f, dty = live ? mk * gd.y : 0.f;   // d T_k = m_k d def
        const float vals[8] = {-(r[4] * dtx + r[6] * dty), -(r[5] * dtx + r[7] * dty), dtx, dty, dtx * rx, dtx * ry, dty * rx, dty * ry};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s = block_sum(vals[j], red);
            if (threadIdx.x == 0) out[(k - 1) * KP_STRIDE + j] = s;
        }
    }
    if (live) {
        float dol = 0.f;
        if ((dlo || stacked) && docc) {
            const float o = occlusion[(size_t)f * plane + p];
            dol = docc[(size_t)f * plane + p] * o * (1.f - o);
        }
        int k0 = K + 1;
        if (stacked) dl[k0++] = dol;
        for (int k = k0; k < ld; ++k) dl[k] = 0.f;
        if (dlo) {
            float* d = dlo + ((size_t)f * plane + p) * ldo;
            d[0] = dol;
            for (int k = 1; k < ldo; ++k) d[k] = 0.f;
        }
    }
}

hipError_t motion_head_forward_launch(const float* lm, int ld, const float* lo, int ldo, const float* rec, int n, int K, int h, int w,
                                      float* mask, float* deformation, float* occlusion, hipStream_t s) {
    if (K < 1 || K > MB_MAXK || ld < K + 1 || (lo && ldo < 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(motion_head_forward_kernel, dim3((h * w + 255) / 256, n), dim3(256), 0, s, lm, ld, lo, ldo, rec, K, h, w, mask,
                       deformation, occlusion);
    return hipGetLastError();
}

hipError_t motion_head_backward_launch(const float* mask, const float* occlusion, const float* rec, int n, int K, int h, int w,
                                       const float* dmask, const float* ddef, const float* docc, float* dlm, int ld, float* dlo, int ldo,
                                       float* drec, float* workspace, hipStream_t s) {
    if (K < 1 || K > MB_MAXK || ld < K + 1 || (dlo && (ldo < 1 || occlusion == nullptr))) return hipErrorInvalidValue;
    const int blocks = (h * w + MB_THREADS - 1) / MB_THREADS;
    // occlusion logits that are channel K + 1 of the mask logits' rows (the caller stacked the two convolutions)
    const int stacked = dlo != nullptr && dlo == dlm + (K + 1) && ldo == ld && ld >= K + 2;
    hipLaunchKernelGGL(motion_head_backward_kernel, dim3(blocks, n), dim3(MB_THREADS), 0, s, mask, occlusion, rec, K, h, w, dmask, ddef, docc,
                       dlm, ld, stacked ? nullptr : dlo, ldo, stacked, workspace);
    const int tot = n * K * KP_STRIDE;
    hipLaunchKernelGGL(rec_partial_reduce_kernel, dim3((tot + 127) / 128), dim3(128), 0, s, workspace, n, blocks, K, 0, drec);
    return hipGetLastError();
}

}  // namespace eamm
