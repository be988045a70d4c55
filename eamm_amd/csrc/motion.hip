// HBM-bound kernels of the dense-motion path: key-point records, heat-map / sparse-motion /
// warped-source front end, softmax-over-motions flow head, flow-gathered feature warp, and the
// once-per-clip source preparation.  One thread per output element group, NHWC / float4 accesses.
//
// Semantics follow SURVEY.md Appendix A; each kernel cites the reference lines it replaces.
#include "kernels.h"

#include <algorithm>

namespace eamm {

// x = 2*(j/(n-1)) - 1, as make_coordinate_grid builds it (reference modules/util.py:844-848).
__device__ __forceinline__ float grid_coord(int j, int n) { return 2.f * ((float)j / (float)(n - 1)) - 1.f; }

// T_k(z) = J_k (z - kp_d) + kp_s (reference modules/dense_motion.py:53-63); rec = kd.xy ks.xy J00 J01 J10 J11.
__device__ __forceinline__ void sparse_motion(const float* __restrict__ rec, float gx, float gy, float& tx,
                                              float& ty) {
    const float rx = gx - rec[0], ry = gy - rec[1];
    tx = fmaf(rec[5], ry, rec[4] * rx) + rec[2];
    ty = fmaf(rec[7], ry, rec[6] * rx) + rec[3];
}

// F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) coordinates and weights.
struct Bilinear {
    int x0, y0;
    float wnw, wne, wsw, wse;
};
__device__ __forceinline__ Bilinear bilinear_setup(float gx, float gy, int W, int H) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Bilinear b;
    // clamp before the int conversion so that wild coordinates stay out of range instead of wrapping
    b.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W);
    b.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);
    const float ax = ix - fx, ay = iy - fy;  // weight of the +1 corner
    b.wnw = (1.f - ax) * (1.f - ay);
    b.wne = ax * (1.f - ay);
    b.wsw = (1.f - ax) * ay;
    b.wse = ax * ay;
    if (!(ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f)) {  // also catches NaN
        b.wnw = b.wne = b.wsw = b.wse = 0.f;
        b.x0 = b.y0 = -2;
    }
    return b;
}

// F.interpolate(mode='bilinear', align_corners=False) source position (reference generator.py:55,83).
struct Lerp {
    int i0, i1;
    float l1;
};
__device__ __forceinline__ Lerp lerp_setup(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = fmaxf(s, 0.f);
    Lerp l;
    l.i0 = (int)s;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.l1 = s - (float)l.i0;
    return l;
}

// ---------------------------------------------------------------------------------------------
// key-point records: J_k = J_source,k * inverse(J_driving,k)  (dense_motion.py:55-56)
// ---------------------------------------------------------------------------------------------
__global__ void kp_prepare_kernel(const float* __restrict__ kd_val, const float* __restrict__ kd_jac,
                                  const float* __restrict__ ks_val, const float* __restrict__ ks_jac, int n, int ns,
                                  int K, float* __restrict__ rec, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int f = i / K, k = i - f * K;
    const int sf = (ns == 1) ? 0 : f;
    float* r = rec + (size_t)i * KP_STRIDE;
    r[0] = kd_val[i * 2 + 0];
    r[1] = kd_val[i * 2 + 1];
    r[2] = ks_val[(sf * K + k) * 2 + 0];
    r[3] = ks_val[(sf * K + k) * 2 + 1];
    float j00 = 1.f, j01 = 0.f, j10 = 0.f, j11 = 1.f;
    if (kd_jac != nullptr) {
        const float* d = kd_jac + (size_t)i * 4;
        const float* s = ks_jac + (size_t)(sf * K + k) * 4;
        const float det = d[0] * d[3] - d[1] * d[2];
        if (!(fabsf(det) > 0.f) || !isfinite(det)) atomicExch(bad, 1);  // torch.inverse raises here
        const float inv = 1.f / det;
        const float i00 = d[3] * inv, i01 = -d[1] * inv, i10 = -d[2] * inv, i11 = d[0] * inv;
        j00 = fmaf(s[1], i10, s[0] * i00);
        j01 = fmaf(s[1], i11, s[0] * i01);
        j10 = fmaf(s[3], i10, s[2] * i00);
        j11 = fmaf(s[3], i11, s[2] * i01);
    }
    r[4] = j00;
    r[5] = j01;
    r[6] = j10;
    r[7] = j11;
}

// ---------------------------------------------------------------------------------------------
// front end: heat-maps (dense_motion.py:32-45, util.py:815-836), sparse motions (:47-67) and the K+1
// bilinear warps of the down-sampled source (:69-79), written straight into the hourglass input
// layout: channel 4k = heat_k, 4k+1..3 = RGB warped by T_k (:93-94), zero-padded to Cpad channels.
// ---------------------------------------------------------------------------------------------
// Image channels beyond three (num_channels 4 .. 6; generator.py:14 accepts any): the channels are handled in GROUPS of three,
// one launch per group g: the per-motion record of the hourglass input grows to G float4 -- group 0 writes (heat, c0, c1, c2) at
// slot k G, group g >= 1 writes (c_3g, c_3g+1, c_3g+2, 0) at slot k G + g -- so that a motion's C + 1 real values stay contiguous
// (heat, c0 .. c_{C-1}) for C <= 6, which is how pad_state_dict lays the filters out.  sparse_deformed planes: [n, K+1, Ctot, h, w],
// this launch writes channels coff .. coff + cn - 1.  One group (G = 1, Ctot = 3, cn = 3): the RGB layout of rounds 1-4.
__global__ __launch_bounds__(256) void motion_front_kernel(const float* __restrict__ rec_all,
                                                           const float4* __restrict__ src_small, int ns, int K, int h,
                                                           int w, float variance, int Cpad,
                                                           float* __restrict__ hg_in,
                                                           float* __restrict__ sparse_deformed, int G, int g, int Ctot, int cn) {
    const int f = blockIdx.y;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= h * w) return;
    const int y = pi / w, x = pi - y * w;
    const float gx = grid_coord(x, w), gy = grid_coord(y, h);
    const float4* src = src_small + (size_t)((ns == 1) ? 0 : f) * h * w;
    const float* rec = rec_all + (size_t)f * K * KP_STRIDE;
    float4* dst = reinterpret_cast<float4*>(hg_in + ((size_t)f * h * w + pi) * Cpad);
    const size_t plane = (size_t)h * w;
    const int coff = 3 * g;
    float* sd = sparse_deformed ? sparse_deformed + ((size_t)f * (K + 1) * Ctot + coff) * plane + pi : nullptr;
    for (int k = 0; k <= K; ++k) {
        float tx = gx, ty = gy, heat = 0.f;
        if (k > 0) {
            const float* r = rec + (k - 1) * KP_STRIDE;
            sparse_motion(r, gx, gy, tx, ty);
            const float dx = gx - r[0], dy = gy - r[1];
            const float sx = gx - r[2], sy = gy - r[3];
            heat = expf(-0.5f * (dx * dx + dy * dy) / variance) - expf(-0.5f * (sx * sx + sy * sy) / variance);
        }
        const Bilinear b = bilinear_setup(tx, ty, w, h);
        float r_ = 0.f, g_ = 0.f, b_ = 0.f;
        const bool x0ok = (unsigned)b.x0 < (unsigned)w, x1ok = (unsigned)(b.x0 + 1) < (unsigned)w;
        const bool y0ok = (unsigned)b.y0 < (unsigned)h, y1ok = (unsigned)(b.y0 + 1) < (unsigned)h;
        // branch-free taps: out-of-range corners read a clamped address and the loaded VALUE is replaced by zero (zeros
        // padding, util.py:69-79; zeroing the weight instead would turn a non-finite border pixel into 0 * Inf = NaN) --
        // a bounds branch around each load made the four loads of a key point, and the key points, wait for one another
        const int xa = min(max(b.x0, 0), w - 1), xb = min(max(b.x0 + 1, 0), w - 1);
        const int ya = min(max(b.y0, 0), h - 1) * w, yb = min(max(b.y0 + 1, 0), h - 1) * w;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 lnw = src[ya + xa], lne = src[ya + xb], lsw = src[yb + xa], lse = src[yb + xb];
        const float4 vnw = (y0ok && x0ok) ? lnw : z4, vne = (y0ok && x1ok) ? lne : z4;
        const float4 vsw = (y1ok && x0ok) ? lsw : z4, vse = (y1ok && x1ok) ? lse : z4;
        const float wnw = b.wnw, wne = b.wne, wsw = b.wsw, wse = b.wse;
        r_ = fmaf(vnw.x, wnw, r_); g_ = fmaf(vnw.y, wnw, g_); b_ = fmaf(vnw.z, wnw, b_);
        r_ = fmaf(vne.x, wne, r_); g_ = fmaf(vne.y, wne, g_); b_ = fmaf(vne.z, wne, b_);
        r_ = fmaf(vsw.x, wsw, r_); g_ = fmaf(vsw.y, wsw, g_); b_ = fmaf(vsw.z, wsw, b_);
        r_ = fmaf(vse.x, wse, r_); g_ = fmaf(vse.y, wse, g_); b_ = fmaf(vse.z, wse, b_);
        dst[k * G + g] = g == 0 ? make_float4(heat, r_, g_, b_) : make_float4(r_, g_, b_, 0.f);
        if (sd) {
            sd[(size_t)(k * Ctot + 0) * plane] = r_;
            if (cn > 1) sd[(size_t)(k * Ctot + 1) * plane] = g_;
            if (cn > 2) sd[(size_t)(k * Ctot + 2) * plane] = b_;
        }
    }
    if (g == 0)
        for (int c4 = (K + 1) * G; c4 < Cpad / 4; ++c4) dst[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// head: mask = softmax over the K+1 motion logits, deformation = sum_k mask_k * T_k, occlusion =
// sigmoid(logit K+1)  (dense_motion.py:98-111).  logits come from the 7x7 MFMA convolution as
// [n,h,w,32] (channels 0..K mask, K+1 occlusion, rest padding).
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_MAXK = 31;

// l[0..K] mask logits, l[K+1] occlusion logit of pixel pi of frame f -> softmax, flow, sigmoid, outputs
__device__ __forceinline__ void head_finish(float (&l)[32], const float* __restrict__ rec, int K, int h, int w, int f,
                                            int pi, int has_occ, float* __restrict__ deformation,
                                            float* __restrict__ occlusion, float* __restrict__ mask_out,
                                            float* __restrict__ occ_out) {
    const int y = pi / w, x = pi - y * w;
    const float gx = grid_coord(x, w), gy = grid_coord(y, h);
    float ov = 0.f;
#pragma unroll
    for (int k = 1; k < 32; ++k)
        if (k == K + 1) ov = l[k];
    float mx = l[0];
#pragma unroll
    for (int k = 1; k < 32; ++k)
        if (k <= K) mx = fmaxf(mx, l[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k)
        if (k <= K) {
            l[k] = expf(l[k] - mx);
            sum += l[k];
        }
    float dx = 0.f, dy = 0.f;
    const size_t plane = (size_t)h * w;
#pragma unroll
    for (int k = 0; k < 32; ++k)
        if (k <= K) {
            const float mk = l[k] / sum;
            float tx = gx, ty = gy;
            if (k > 0) sparse_motion(rec + (k - 1) * KP_STRIDE, gx, gy, tx, ty);
            dx = fmaf(mk, tx, dx);
            dy = fmaf(mk, ty, dy);
            if (mask_out) mask_out[((size_t)f * (K + 1) + k) * plane + pi] = mk;
        }
    reinterpret_cast<float2*>(deformation)[(size_t)f * plane + pi] = make_float2(dx, dy);
    if (has_occ) {
        const float o = 1.f / (1.f + expf(-ov));
        occlusion[(size_t)f * plane + pi] = o;
        if (occ_out) occ_out[(size_t)f * plane + pi] = o;
    }
}

__global__ __launch_bounds__(256) void motion_head_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ rec_all, int K, int h, int w,
                                                          int has_occ, float* __restrict__ deformation,
                                                          float* __restrict__ occlusion, float* __restrict__ mask_out,
                                                          float* __restrict__ occ_out) {
    const int f = blockIdx.y;
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= h * w) return;
    const float4* lg4 = reinterpret_cast<const float4*>(logits + ((size_t)f * h * w + pi) * 32);
    float l[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = lg4[i];
        l[4 * i + 0] = v.x; l[4 * i + 1] = v.y; l[4 * i + 2] = v.z; l[4 * i + 3] = v.w;
    }
    head_finish(l, rec_all + (size_t)f * K * KP_STRIDE, K, h, w, f, pi, has_occ, deformation, occlusion, mask_out,
                occ_out);
}

// Row-split form: the 7x7 head ran as a 7x1 (vertical) MFMA convolution with N = (dx, co), co < NC = K+1(+1):
// part[f,y,x',dx*NC+co] (pixel stride PS floats).  One block = 64 pixels of one row; the 70 x 7*NC partial products
// are staged in LDS (odd row stride: conflict-free column walks), then each thread gathers its 7 horizontal taps,
// adds the bias and finishes as above.
constexpr int HR_TILE = 64;
__global__ __launch_bounds__(HR_TILE) void motion_head_rowsplit_kernel(const float* __restrict__ part, int PS, int NC,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ rec_all, int K, int h,
                                                                       int w, int has_occ,
                                                                       float* __restrict__ deformation,
                                                                       float* __restrict__ occlusion,
                                                                       float* __restrict__ mask_out,
                                                                       float* __restrict__ occ_out) {
    extern __shared__ float tile[];   // [(HR_TILE + 6)][7*NC | 1]
    const int NV = 7 * NC, LD = NV | 1;
    const int x0 = blockIdx.x * HR_TILE, y = blockIdx.y, f = blockIdx.z;
    const float* row = part + ((size_t)(f * h + y) * w) * PS;
    if (((NV | PS) & 3) == 0) {   // 16-byte loads (the pixel stride and the used width are multiples of 4 floats)
        const int NV4 = NV >> 2;
        for (int i = threadIdx.x; i < (HR_TILE + 6) * NV4; i += blockDim.x) {
            const int px = i / NV4, c4 = i - px * NV4;
            const int x = x0 + px - 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)x < (unsigned)w) v = *reinterpret_cast<const float4*>(row + (size_t)x * PS + 4 * c4);
            float* d = tile + px * LD + 4 * c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < (HR_TILE + 6) * NV; i += blockDim.x) {
            const int px = i / NV, c = i - px * NV;
            const int x = x0 + px - 3;
            tile[px * LD + c] = (unsigned)x < (unsigned)w ? row[(size_t)x * PS + c] : 0.f;
        }
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    float l[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        float v = 0.f;
        if (k < NC) {
            v = bias[k];
            for (int dx = 0; dx < 7; ++dx) v += tile[(threadIdx.x + dx) * LD + dx * NC + k];
        }
        l[k] = v;
    }
    head_finish(l, rec_all + (size_t)f * K * KP_STRIDE, K, h, w, f, y * w + x, has_occ, deformation, occlusion,
                mask_out, occ_out);
}

// flow / occlusion at feature-map pixel (y,x): direct read when the motion grid matches the feature
// map, else the reference's bilinear resize (generator.py:52-56, 82-83).
__device__ __forceinline__ void flow_at(const float2* __restrict__ defo, const float* __restrict__ occ, int h,
                                        int w, int H, int W, int y, int x, float& gx, float& gy, float& o) {
    if (h == H && w == W) {
        const float2 d = defo[y * w + x];
        gx = d.x;
        gy = d.y;
        o = occ ? occ[y * w + x] : 1.f;
        return;
    }
    const Lerp ly = lerp_setup(y, h, H), lx = lerp_setup(x, w, W);
    const float2 d00 = defo[ly.i0 * w + lx.i0], d01 = defo[ly.i0 * w + lx.i1];
    const float2 d10 = defo[ly.i1 * w + lx.i0], d11 = defo[ly.i1 * w + lx.i1];
    const float wy1 = ly.l1, wy0 = 1.f - wy1, wx1 = lx.l1, wx0 = 1.f - wx1;
    gx = wy0 * (wx0 * d00.x + wx1 * d01.x) + wy1 * (wx0 * d10.x + wx1 * d11.x);
    gy = wy0 * (wx0 * d00.y + wx1 * d01.y) + wy1 * (wx0 * d10.y + wx1 * d11.y);
    o = 1.f;
    if (occ) {
        o = wy0 * (wx0 * occ[ly.i0 * w + lx.i0] + wx1 * occ[ly.i0 * w + lx.i1]) +
            wy1 * (wx0 * occ[ly.i1 * w + lx.i0] + wx1 * occ[ly.i1 * w + lx.i1]);
    }
}

// ---------------------------------------------------------------------------------------------
// feature warp: out = grid_sample(feat, deformation) * occlusion  (generator.py:50-57, 79-84), NHWC,
// one thread per (pixel, 4 channels): a wave covers 256 contiguous channels of one pixel, so every
// corner gather is a 1 KiB coalesced read.  Also emits the first res-block's pre-activation
// relu(bn1(out)) (util.py:873-874) so that the bottleneck convolutions read ready operands.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_features_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ deformation,
                                                            const float* __restrict__ occlusion, int n, int ns,
                                                            int hf, int wf, int C, int h, int w,
                                                            float* __restrict__ out, float* __restrict__ out2,
                                                            const float* __restrict__ s2,
                                                            const float* __restrict__ t2) {
    // one thread = one pixel x TWO 4-channel groups (c4 and c4 + c4n/2): the flow / bilinear setup is shared and
    // eight 16-byte gathers are in flight per thread; a wave still covers contiguous 1 KiB runs of each half
    const int c4n = C >> 2, half_n = c4n >> 1;
    const size_t total = (size_t)n * hf * wf * half_n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % half_n);
        const size_t pix = idx / half_n;
        const int x = (int)(pix % wf);
        const int y = (int)((pix / wf) % hf);
        const int f = (int)(pix / ((size_t)wf * hf));
        float gx, gy, o;
        flow_at(reinterpret_cast<const float2*>(deformation) + (size_t)f * h * w,
                occlusion ? occlusion + (size_t)f * h * w : nullptr, h, w, hf, wf, y, x, gx, gy, o);
        const Bilinear b = bilinear_setup(gx, gy, wf, hf);
        const float4* src = reinterpret_cast<const float4*>(feat + (size_t)((ns == 1) ? 0 : f) * hf * wf * C) + c4;
        const bool x0ok = (unsigned)b.x0 < (unsigned)wf, x1ok = (unsigned)(b.x0 + 1) < (unsigned)wf;
        const bool y0ok = (unsigned)b.y0 < (unsigned)hf, y1ok = (unsigned)(b.y0 + 1) < (unsigned)hf;
        const float wgt[4] = {b.wnw, b.wne, b.wsw, b.wse};
        const bool cok[4] = {y0ok && x0ok, y0ok && x1ok, y1ok && x0ok, y1ok && x1ok};
        // clamp the corner coordinates so every load is in range; the VALUE of an out-of-range corner is replaced by zero
        // (grid_sample's zeros padding; a zero weight would let a non-finite border value through as 0 * Inf = NaN)
        const int xa = min(max(b.x0, 0), wf - 1), xb = min(max(b.x0 + 1, 0), wf - 1);
        const int ya = min(max(b.y0, 0), hf - 1), yb = min(max(b.y0 + 1, 0), hf - 1);
        const size_t off[4] = {(size_t)(ya * wf + xa) * c4n, (size_t)(ya * wf + xb) * c4n, (size_t)(yb * wf + xa) * c4n,
                               (size_t)(yb * wf + xb) * c4n};
        float4 v[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[g][k] = src[off[k] + g * half_n];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!cok[k]) v[g][k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // same accumulation order as before: nw, ne, sw, se
                acc.x = fmaf(v[g][k].x, wgt[k], acc.x); acc.y = fmaf(v[g][k].y, wgt[k], acc.y);
                acc.z = fmaf(v[g][k].z, wgt[k], acc.z); acc.w = fmaf(v[g][k].w, wgt[k], acc.w);
            }
            acc.x *= o; acc.y *= o; acc.z *= o; acc.w *= o;
            const size_t oidx = pix * c4n + c4 + g * half_n;
            reinterpret_cast<float4*>(out)[oidx] = acc;
            if (out2) {
                const float4 s = reinterpret_cast<const float4*>(s2)[c4 + g * half_n];
                const float4 t = reinterpret_cast<const float4*>(t2)[c4 + g * half_n];
                float4 a;
                a.x = fmaxf(fmaf(acc.x, s.x, t.x), 0.f); a.y = fmaxf(fmaf(acc.y, s.y, t.y), 0.f);
                a.z = fmaxf(fmaf(acc.z, s.z, t.z), 0.f); a.w = fmaxf(fmaf(acc.w, s.w, t.w), 0.f);
                reinterpret_cast<float4*>(out2)[oidx] = a;
            }
        }
    }
}

// 'deformed' side output: flow resized to the frame, grid_sample of the RGB source (generator.py:86).
__global__ __launch_bounds__(256) void warp_image_kernel(const float* __restrict__ src,
                                                         const float* __restrict__ deformation, int n, int ns, int H,
                                                         int W, int h, int w, float* __restrict__ out, int Cplanes,
                                                         int Cout) {
    const size_t total = (size_t)n * H * W;
    const size_t plane = (size_t)H * W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int f = (int)(idx / plane);
        float gx, gy, o;
        flow_at(reinterpret_cast<const float2*>(deformation) + (size_t)f * h * w, nullptr, h, w, H, W, y, x, gx, gy,
                o);
        const Bilinear b = bilinear_setup(gx, gy, W, H);
        const float* s = src + (size_t)((ns == 1) ? 0 : f) * Cplanes * plane;
        const bool x0ok = (unsigned)b.x0 < (unsigned)W, x1ok = (unsigned)(b.x0 + 1) < (unsigned)W;
        const bool y0ok = (unsigned)b.y0 < (unsigned)H, y1ok = (unsigned)(b.y0 + 1) < (unsigned)H;
        for (int c = 0; c < Cout; ++c) {
            const float* pc = s + c * plane;
            float v = 0.f;
            if (y0ok && x0ok) v = fmaf(pc[b.y0 * W + b.x0], b.wnw, v);
            if (y0ok && x1ok) v = fmaf(pc[b.y0 * W + b.x0 + 1], b.wne, v);
            if (y1ok && x0ok) v = fmaf(pc[(b.y0 + 1) * W + b.x0], b.wsw, v);
            if (y1ok && x1ok) v = fmaf(pc[(b.y0 + 1) * W + b.x0 + 1], b.wse, v);
            out[((size_t)f * Cout + c) * plane + (size_t)y * W + x] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// once per clip: NCHW RGB source -> (a) NHWC zero-padded to Cpad channels for the 7x7 MFMA encoder
// convolution, (b) anti-aliased, down-sampled [h,w,4] copy for the motion front end
// (AntiAliasInterpolation2d, util.py:1044-1052: zero pad 6, depthwise 13x13, keep every inv_scale-th
// row/column -- only the kept outputs are computed).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void source_to_nhwc_kernel(const float* __restrict__ src, int ns, int H, int W,
                                                             int Cpad, float* __restrict__ dst) {
    const size_t plane = (size_t)H * W;
    const int c4n = Cpad >> 2;
    const size_t total = (size_t)ns * plane * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t pix = idx / c4n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 == 0) {
            const size_t b = pix / plane, r = pix % plane;
            const float* p = src + b * 3 * plane + r;
            v = make_float4(p[0], p[plane], p[2 * plane], 0.f);
        }
        reinterpret_cast<float4*>(dst)[idx] = v;
    }
}

// cn (1 .. 4) channels coff .. coff + cn - 1 of a source with Cplanes planes per image -> one float4 per kept pixel, written at
// float4 slot `slot` of the pixel's c4pad slots; `fill`: the pixel's other slots are zeroed (the first launch of a pixel line).
__global__ __launch_bounds__(256) void antialias_down_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ aa_w, int ns, int H, int W,
                                                             int inv_scale, int c4pad, float4* __restrict__ dst, int Cplanes,
                                                             int coff, int cn, int slot, int fill) {
    const int h = H / inv_scale, w = W / inv_scale;
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)ns * h * w;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h), b = (int)(idx / ((size_t)w * h));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (inv_scale == 1) {
        for (int c = 0; c < cn; ++c) acc[c] = src[((size_t)b * Cplanes + coff + c) * plane + (size_t)y * W + x];
    } else {
        const int cy = y * inv_scale - 6, cx = x * inv_scale - 6;
        for (int c = 0; c < cn; ++c) {
            const float* p = src + ((size_t)b * Cplanes + coff + c) * plane;
            const float* k = aa_w + (coff + c) * 169;
            float s = 0.f;
            for (int ky = 0; ky < 13; ++ky) {
                const int yy = cy + ky;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int kx = 0; kx < 13; ++kx) {
                    const int xx = cx + kx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    s = fmaf(p[(size_t)yy * W + xx], k[ky * 13 + kx], s);
                }
            }
            acc[c] = s;
        }
    }
    dst[idx * c4pad + slot] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (fill)
        for (int k = 0; k < c4pad; ++k)
            if (k != slot) dst[idx * c4pad + k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Final 7x7 convolution, second half (generator.py:92-93).  The MFMA kernel ran it as a 7x1 (vertical)
// convolution with N = (dx, co): part[b,y,x',dx*3+co] = sum_{dy,c} in[b,y+dy-3,x',c] * w[co,c,dy,dx].
// Here the seven horizontal taps are gathered: out[b,co,y,x] = sigmoid(bias[co] + sum_dx part[b,y,x+dx-3,
// dx*3+co]) (zero outside the row), written NCHW.  One block = one row segment of 128 pixels staged in
// LDS (row stride 33 dwords: conflict-free for the per-pixel column walk).
constexpr int FS_TILE = 128, FS_LD = 33;
__global__ __launch_bounds__(128) void final_shift_sum_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ bias, int H, int W,
                                                              float* __restrict__ out) {
    __shared__ float tile[(FS_TILE + 6) * FS_LD];
    const int x0 = blockIdx.x * FS_TILE, y = blockIdx.y, b = blockIdx.z;
    const float4* row = reinterpret_cast<const float4*>(part + ((size_t)(b * H + y) * W) * 32);
    for (int i = threadIdx.x; i < (FS_TILE + 6) * 8; i += blockDim.x) {
        const int px = i >> 3, c4 = i & 7;
        const int x = x0 + px - 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)x < (unsigned)W) v = row[(size_t)x * 8 + c4];
        float* d = tile + px * FS_LD + c4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= W) return;
    float acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
    for (int dx = 0; dx < 7; ++dx) {
        const float* s = tile + (threadIdx.x + dx) * FS_LD + dx * 3;
        acc[0] += s[0];
        acc[1] += s[1];
        acc[2] += s[2];
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int co = 0; co < 3; ++co)
        out[((size_t)b * 3 + co) * plane + (size_t)y * W + x] = 1.f / (1.f + expf(-acc[co]));
}

// prediction [n,3,H,W] float -> [n,H,W,3] uint8 (img_as_ubyte rounding), the frame format demo.py:507 saves.
__global__ __launch_bounds__(256) void to_u8_kernel(const float* __restrict__ pred, int n, int H, int W,
                                                    uint8_t* __restrict__ out) {
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)n * plane;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t f = idx / plane, r = idx % plane;
        const float* p = pred + f * 3 * plane + r;
        for (int c = 0; c < 3; ++c) {
            const float v = fminf(fmaxf(rintf(p[c * plane] * 255.f), 0.f), 255.f);
            out[idx * 3 + c] = (uint8_t)v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int grid_for(size_t total, int cap = 16384) {
    return (int)std::min<size_t>((total + 255) / 256, (size_t)cap);
}

hipError_t kp_prepare_launch(const float* kd_val, const float* kd_jac, const float* ks_val, const float* ks_jac, int n,
                             int ns, int K, float* kp_rec, int* bad_flag, hipStream_t s) {
    hipLaunchKernelGGL(kp_prepare_kernel, dim3((n * K + 255) / 256), dim3(256), 0, s, kd_val, kd_jac, ks_val, ks_jac,
                       n, ns, K, kp_rec, bad_flag);
    return hipGetLastError();
}

hipError_t motion_front_launch(const float* kp_rec, const float* src_small, int n, int ns, int K, int h, int w,
                               float variance, int Cpad, float* hg_in, float* sparse_deformed, hipStream_t s, int groups,
                               int channels, size_t group_stride) {
    // one launch per group of three image channels (groups = 1, channels = 3: the RGB layout); src_small of group g starts
    // group_stride floats behind group g - 1's
    if (groups < 1 || groups > 2 || channels < 1 || channels > 3 * groups || (K + 1) * 4 * groups > Cpad) return hipErrorInvalidValue;
    for (int g = 0; g < groups; ++g)
        hipLaunchKernelGGL(motion_front_kernel, dim3((h * w + 255) / 256, n), dim3(256), 0, s, kp_rec,
                           reinterpret_cast<const float4*>(src_small + (size_t)g * group_stride), ns, K, h, w, variance, Cpad, hg_in,
                           sparse_deformed, groups, g, groups == 1 ? 3 : channels, std::min(3, channels - 3 * g));
    return hipGetLastError();
}

hipError_t motion_head_launch(const float* logits, const float* kp_rec, int n, int K, int h, int w, int has_occ,
                              float* deformation, float* occlusion, float* mask_out, float* occ_out, hipStream_t s) {
    if (K > HEAD_MAXK - 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(motion_head_kernel, dim3((h * w + 255) / 256, n), dim3(256), 0, s, logits, kp_rec, K, h, w,
                       has_occ, deformation, occlusion, mask_out, occ_out);
    return hipGetLastError();
}

hipError_t motion_head_rowsplit_launch(const float* part, int PS, int NC, const float* bias, const float* kp_rec, int n,
                                       int K, int h, int w, int has_occ, float* deformation, float* occlusion,
                                       float* mask_out, float* occ_out, hipStream_t s) {
    if (K > HEAD_MAXK - 1 || NC > 32) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (HR_TILE + 6) * ((7 * NC) | 1);
    hipLaunchKernelGGL(motion_head_rowsplit_kernel, dim3((w + HR_TILE - 1) / HR_TILE, h, n), dim3(HR_TILE), lds, s, part,
                       PS, NC, bias, kp_rec, K, h, w, has_occ, deformation, occlusion, mask_out, occ_out);
    return hipGetLastError();
}

hipError_t warp_features_launch(const float* feat, const float* deformation, const float* occlusion, int n, int ns,
                                int hf, int wf, int C, int h, int w, float* out, float* out2, const float* s2,
                                const float* t2, hipStream_t s) {
    if (C % 8) return hipErrorInvalidValue;
    const size_t total = (size_t)n * hf * wf * (C / 8);
    hipLaunchKernelGGL(warp_features_kernel, dim3(grid_for(total, 1 << 20)), dim3(256), 0, s, feat, deformation,
                       occlusion, n, ns, hf, wf, C, h, w, out, out2, s2, t2);
    return hipGetLastError();
}

hipError_t warp_image_launch(const float* src, const float* deformation, int n, int ns, int H, int W, int h, int w,
                             float* out, hipStream_t s, int src_planes, int channels) {
    const size_t total = (size_t)n * H * W;
    hipLaunchKernelGGL(warp_image_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, deformation, n, ns, H, W, h, w,
                       out, src_planes, channels);
    return hipGetLastError();
}

// A generator built without a motion network (dense_motion_params=None, generator.py:22-23, 64): the encoder features go
// to the bottleneck unwarped -- frame i gets source i (or the one cached source); second output relu(x*s + t) is the first
// res-block's pre-activation for the direct convolution form.
__global__ void broadcast_features_kernel(const float4* __restrict__ feat, int ns, size_t per_frame4, int C4, size_t total4,
                                          float4* __restrict__ out, float4* __restrict__ out2, const float4* __restrict__ s2,
                                          const float4* __restrict__ t2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / per_frame4, r = i - f * per_frame4;
        const float4 v = feat[(ns == 1 ? 0 : f) * per_frame4 + r];
        out[i] = v;
        if (out2 != nullptr) {
            const float4 s = s2[r % C4], t = t2[r % C4];
            float4 a;
            a.x = fmaxf(fmaf(v.x, s.x, t.x), 0.f); a.y = fmaxf(fmaf(v.y, s.y, t.y), 0.f);
            a.z = fmaxf(fmaf(v.z, s.z, t.z), 0.f); a.w = fmaxf(fmaf(v.w, s.w, t.w), 0.f);
            out2[i] = a;
        }
    }
}

hipError_t broadcast_features_launch(const float* feat, int n, int ns, int hf, int wf, int C, float* out, float* out2,
                                     const float* s2, const float* t2, hipStream_t s) {
    if (C & 3) return hipErrorInvalidValue;
    const size_t per_frame4 = (size_t)hf * wf * C / 4, total4 = per_frame4 * n;
    const int blocks = (int)std::min<size_t>((total4 + 255) / 256, (size_t)1 << 16);
    hipLaunchKernelGGL(broadcast_features_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(feat), ns,
                       per_frame4, C / 4, total4, reinterpret_cast<float4*>(out), reinterpret_cast<float4*>(out2),
                       reinterpret_cast<const float4*>(s2), reinterpret_cast<const float4*>(t2));
    return hipGetLastError();
}

hipError_t source_prepare_launch(const float* src, const float* aa_w, int ns, int H, int W, int inv_scale, int Cpad,
                                 float* src_nhwc, float* src_small, hipStream_t s) {
    const size_t t1 = (size_t)ns * H * W * (Cpad / 4);
    hipLaunchKernelGGL(source_to_nhwc_kernel, dim3(grid_for(t1)), dim3(256), 0, s, src, ns, H, W, Cpad, src_nhwc);
    const size_t t2 = (size_t)ns * (H / inv_scale) * (W / inv_scale);
    hipLaunchKernelGGL(antialias_down_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, src, aa_w, ns, H, W,
                       inv_scale, 1, reinterpret_cast<float4*>(src_small), 3, 0, 3, 0, 1);
    return hipGetLastError();
}

hipError_t antialias_down_launch(const float* src, const float* aa_w, int ns, int H, int W, int inv_scale, int Cpad,
                                 float* dst, hipStream_t s, int src_planes, int first_channel, int channels, int slot) {
    // `channels` (1 .. 4) channels from first_channel on of a source with src_planes planes per image (aa_w: one 13x13 filter per
    // plane) into float4 slot `slot` of every pixel's Cpad / 4 slots; slot 0 also zeroes the pixel's other slots
    if (channels < 1 || channels > 4 || slot < 0 || slot >= Cpad / 4 || first_channel + channels > src_planes) return hipErrorInvalidValue;
    const size_t t2 = (size_t)ns * (H / inv_scale) * (W / inv_scale);
    hipLaunchKernelGGL(antialias_down_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, src, aa_w, ns, H, W,
                       inv_scale, Cpad / 4, reinterpret_cast<float4*>(dst), src_planes, first_channel, channels, slot, slot == 0 ? 1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// key-point detector head (reference modules/keypoint_detector.py:77-105 / 180-205)
// ---------------------------------------------------------------------------------------------
// [B,C,h,w] feature map -> [B,h,w,Cpad] (zero padded): the layout the MFMA convolutions read.
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, int B, int C, int HW,
                                                               int Cpad, float* __restrict__ dst, int planes) {
    const size_t total = (size_t)B * HW * Cpad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cpad);
        const size_t pix = idx / Cpad;
        const size_t b = pix / HW, r = pix % HW;
        dst[idx] = c < C ? src[(b * planes + c) * HW + r] : 0.f;     // (the first C of `planes` planes per image)
    }
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    // 256 threads: wave shuffles, then one value per wave through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : v + u;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// One block per (key point k, image b).  logits: [B,h,w,Cs] from the 7x7 "same" MFMA convolution; channel k is
// the key-point logit, channels K + 4*m + c the jacobian maps.  The reference convolves with padding `pad` (0),
// i.e. only the interior [off, off+oh) x [off, off+ow), off = 3 - pad, exists: heatmap = softmax(logit /
// temperature) over that window, value = sum heatmap * grid(oh, ow), jacobian = sum heatmap * jacobian_map.
__global__ __launch_bounds__(256) void kp_head_kernel(const float* __restrict__ logits, int K, int njm, int h, int w,
                                                      int Cs, int off, int oh, int ow, float temperature,
                                                      float* __restrict__ value, float* __restrict__ jacobian,
                                                      float* __restrict__ heatmap) {
    __shared__ float red[8];
    const int k = blockIdx.x, b = blockIdx.y;
    const int n = oh * ow;
    const float* base = logits + (size_t)b * h * w * Cs;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / ow + off, x = i % ow + off;
        m = fmaxf(m, base[(size_t)(y * w + x) * Cs + k] / temperature);
    }
    m = block_reduce(m, red, true);
    float s = 0.f, vx = 0.f, vy = 0.f, j0 = 0.f, j1 = 0.f, j2 = 0.f, j3 = 0.f;
    const int jm = njm > 0 ? K + 4 * (njm == 1 ? 0 : k) : 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int yy = i / ow, xx = i % ow;
        const float* px = base + (size_t)((yy + off) * w + xx + off) * Cs;
        const float e = expf(px[k] / temperature - m);
        s += e;
        vx = fmaf(e, grid_coord(xx, ow), vx);
        vy = fmaf(e, grid_coord(yy, oh), vy);
        if (njm > 0) {
            j0 = fmaf(e, px[jm + 0], j0);
            j1 = fmaf(e, px[jm + 1], j1);
            j2 = fmaf(e, px[jm + 2], j2);
            j3 = fmaf(e, px[jm + 3], j3);
        }
    }
    s = block_reduce(s, red, false);
    vx = block_reduce(vx, red, false);
    vy = block_reduce(vy, red, false);
    if (njm > 0) {
        j0 = block_reduce(j0, red, false);
        j1 = block_reduce(j1, red, false);
        j2 = block_reduce(j2, red, false);
        j3 = block_reduce(j3, red, false);
    }
    if (threadIdx.x == 0) {
        value[((size_t)b * K + k) * 2 + 0] = vx / s;
        value[((size_t)b * K + k) * 2 + 1] = vy / s;
        if (njm > 0 && jacobian) {
            float* jo = jacobian + ((size_t)b * K + k) * 4;
            jo[0] = j0 / s; jo[1] = j1 / s; jo[2] = j2 / s; jo[3] = j3 / s;
        }
    }
    if (heatmap) {
        float* ho = heatmap + ((size_t)b * K + k) * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int y = i / ow + off, x = i % ow + off;
            ho[i] = expf(base[(size_t)(y * w + x) * Cs + k] / temperature - m) / s;
        }
    }
}

// Round 6, batched calls: the reductions split over PIXEL SLICES so that the whole chip streams the logits, each line read by one block.
// The per-(image, key point) kernel above runs K x B blocks that each read one float of every 256-byte logit line three times over;
// with one block per image (a first version of this round, 109 us per 64-frame batch against 116) the 67 MB of a batch hang on 64 CUs
// at ~30 GB/s each.  Here block (b, slice) owns n / S pixels and ALL key points of a group of KG: pass 1 its slice's maxima, pass 2 the
// sums relative to them (the slice's lines come out of the L2 the second time) -> `part` [B][K][S][8] = (max, s, vx, vy, j0..j3).
// kp_head_combine_kernel folds the slices in slice order: M = max m_s, total = sum_s exp(m_s - M) * part_s (deterministic; the
// rescaling rounds once more than the one-block form: <= 2e-7 of the value), writes value / jacobian and (M, S) for the optional
// heat-map pass kp_head_heatmap_kernel.
template <int KG>
__global__ __launch_bounds__(256) void kp_head_slice_kernel(const float* __restrict__ logits, int K, int njm, int h, int w, int Cs,
                                                            int off, int oh, int ow, float temperature, int S,
                                                            float* __restrict__ part) {
    constexpr int NV = 7;                       // s, vx, vy, j0..j3
    __shared__ float red[4][KG * NV];           // one row per wave
    __shared__ float fin[KG];
    const int b = blockIdx.x, sl = blockIdx.y, k0 = blockIdx.z * KG;
    const int kn = min(KG, K - k0);
    const int n = oh * ow;
    const int i0 = (int)((long long)n * sl / S), i1 = (int)((long long)n * (sl + 1) / S);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* base = logits + (size_t)b * h * w * Cs;
    float m[KG];
#pragma unroll
    for (int k = 0; k < KG; ++k) m[k] = -INFINITY;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float* px = base + (size_t)((i / ow + off) * w + i % ow + off) * Cs + k0;
#pragma unroll
        for (int k = 0; k < KG; ++k)
            if (k < kn) m[k] = fmaxf(m[k], px[k] / temperature);
    }
#pragma unroll
    for (int k = 0; k < KG; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], o));
        if (lane == 0) red[wave][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < KG) fin[threadIdx.x] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KG; ++k) m[k] = fin[k];
    __syncthreads();
    float acc[KG][NV];
#pragma unroll
    for (int k = 0; k < KG; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[k][v] = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const int yy = i / ow, xx = i % ow;
        const float* px = base + (size_t)((yy + off) * w + xx + off) * Cs;
        const float gx = grid_coord(xx, ow), gy = grid_coord(yy, oh);
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            if (k < kn) {
                const float e = expf(px[k0 + k] / temperature - m[k]);
                acc[k][0] += e;
                acc[k][1] = fmaf(e, gx, acc[k][1]);
                acc[k][2] = fmaf(e, gy, acc[k][2]);
                if (njm > 0) {
                    const float* pj = px + K + 4 * (njm == 1 ? 0 : k0 + k);
                    acc[k][3] = fmaf(e, pj[0], acc[k][3]);
                    acc[k][4] = fmaf(e, pj[1], acc[k][4]);
                    acc[k][5] = fmaf(e, pj[2], acc[k][5]);
                    acc[k][6] = fmaf(e, pj[3], acc[k][6]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KG; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float x = acc[k][v];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if (lane == 0) red[wave][k * NV + v] = x;
        }
    __syncthreads();
    if (threadIdx.x < kn * 8) {
        const int k = threadIdx.x >> 3, v = threadIdx.x & 7;
        float r = v == 0 ? fin[k] : ((red[0][k * NV + v - 1] + red[1][k * NV + v - 1]) + red[2][k * NV + v - 1]) + red[3][k * NV + v - 1];
        part[(((size_t)b * K + k0 + k) * S + sl) * 8 + v] = r;
    }
}

__global__ __launch_bounds__(64) void kp_head_combine_kernel(const float* __restrict__ part, int K, int njm, int S,
                                                            float* __restrict__ value, float* __restrict__ jacobian,
                                                            float* __restrict__ stats) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* p = part + ((size_t)b * K + k) * S * 8;
        float M = -INFINITY;
        for (int sl = 0; sl < S; ++sl) M = fmaxf(M, p[sl * 8]);
        float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < S; ++sl) {
            const float sc = expf(p[sl * 8] - M);        // (an empty slice has max -inf: scale 0, sums 0)
#pragma unroll
            for (int v = 0; v < 7; ++v) t[v] = fmaf(sc, p[sl * 8 + 1 + v], t[v]);
        }
        value[((size_t)b * K + k) * 2 + 0] = t[1] / t[0];
        value[((size_t)b * K + k) * 2 + 1] = t[2] / t[0];
        if (njm > 0 && jacobian) {
            float* jo = jacobian + ((size_t)b * K + k) * 4;
            jo[0] = t[3] / t[0]; jo[1] = t[4] / t[0]; jo[2] = t[5] / t[0]; jo[3] = t[6] / t[0];
        }
        stats[((size_t)b * K + k) * 2 + 0] = M;
        stats[((size_t)b * K + k) * 2 + 1] = t[0];
    }
}

__global__ __launch_bounds__(256) void kp_head_heatmap_kernel(const float* __restrict__ logits, const float* __restrict__ stats, int K, int h,
                                                             int w, int Cs, int off, int oh, int ow, float temperature, int S,
                                                             float* __restrict__ heatmap) {
    const int b = blockIdx.x, sl = blockIdx.y;
    const int n = oh * ow;
    const int i0 = (int)((long long)n * sl / S), i1 = (int)((long long)n * (sl + 1) / S);
    const float* base = logits + (size_t)b * h * w * Cs;
    const float* st = stats + (size_t)b * K * 2;
    for (int k = 0; k < K; ++k) {      // (k outer: a wave's stores of one key point are contiguous)
        const float M = st[2 * k], Sd = st[2 * k + 1];
        float* ho = heatmap + ((size_t)b * K + k) * n;
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x)
            ho[i] = expf(base[(size_t)((i / ow + off) * w + i % ow + off) * Cs + k] / temperature - M) / Sd;
    }
}

hipError_t nchw_to_nhwc_pad_launch(const float* src, int B, int C, int H, int W, int Cpad, float* dst, hipStream_t s, int planes) {
    const size_t total = (size_t)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, B, C, H * W, Cpad, dst, planes > 0 ? planes : C);
    return hipGetLastError();
}

size_t kp_head_workspace_floats(int B, int K) { return (size_t)B * K * (KP_HEAD_SLICES * 8 + 2); }

hipError_t kp_head_launch(const float* logits, int B, int K, int njm, int h, int w, int Cs, int pad, float temperature,
                          float* value, float* jacobian, float* heatmap, hipStream_t s, float* ws) {
    const int off = 3 - pad, oh = h - 2 * off, ow = w - 2 * off;
    if (off < 0 || oh < 1 || ow < 1) return hipErrorInvalidValue;
    // batched calls with a workspace: pixel slices over the whole chip + a combine step (round 6, kp_head_slice_kernel above).  Few images
    // (or no workspace): one block per (image, key point).  EAMM_KP_HEAD_SLICE_MIN_B (tuning aid): < 0 never
    static const int slice_min_b = (int)knob_int("EAMM_KP_HEAD_SLICE_MIN_B", 8);
    if (ws != nullptr && slice_min_b >= 0 && B >= slice_min_b) {
        constexpr int KG = 10;
        const int S = KP_HEAD_SLICES;
        float* part = ws;
        float* stats = ws + (size_t)B * K * S * 8;
        hipLaunchKernelGGL(kp_head_slice_kernel<KG>, dim3(B, S, (K + KG - 1) / KG), dim3(256), 0, s, logits, K, njm, h, w, Cs, off, oh, ow,
                           temperature, S, part);
        hipLaunchKernelGGL(kp_head_combine_kernel, dim3(B), dim3(64), 0, s, part, K, njm, S, value, jacobian, stats);
        if (heatmap)
            hipLaunchKernelGGL(kp_head_heatmap_kernel, dim3(B, S), dim3(256), 0, s, logits, stats, K, h, w, Cs, off, oh, ow, temperature, S, heatmap);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kp_head_kernel, dim3(K, B), dim3(256), 0, s, logits, K, njm, h, w, Cs, off, oh, ow, temperature,
                       value, jacobian, heatmap);
    return hipGetLastError();
}

hipError_t final_shift_sum_launch(const float* part, const float* bias, int n, int H, int W, float* out,
                                  hipStream_t s) {
    hipLaunchKernelGGL(final_shift_sum_kernel, dim3((W + FS_TILE - 1) / FS_TILE, H, n), dim3(FS_TILE), 0, s, part,
                       bias, H, W, out);
    return hipGetLastError();
}

hipError_t to_u8_launch(const float* pred, int n, int H, int W, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(to_u8_kernel, dim3(grid_for((size_t)n * H * W)), dim3(256), 0, s, pred, n, H, W, out);
    return hipGetLastError();
}

}  // namespace eamm
