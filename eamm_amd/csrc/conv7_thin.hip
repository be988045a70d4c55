// The two 7x7 layers of the generator with THREE channels on one side -- `first` (3 -> block_expansion) and `final`
// (block_expansion -> 3), reference modules/generator.py:26, 48 -- for the training path.
// Padding the thin side to the implicit-GEMM kernels' 32-channel granule wastes 10x (measured: 7 ms of a 26 ms step), so the
// thin side is kept as ONE float4 per pixel ([B,H,W,4], fourth channel zero) and becomes part of the GEMM's K or N index
// together with the 49 taps: k = (ky * 7 + kx) * 4 + ci, 196 values.
//   conv7_thin_in_kernel     y[p][n] = sum_k patch[p + tap(k)][ci(k)] * Wk[k][n] (+ bias):  the forward of `first`, and the DATA
//                            gradient of `final` (x = d(pre-sigmoid) [.,4], Wk = the filter transposed and flipped)
//   conv7_thin_wgrad_kernel  D[c][k] = sum_p wide[p][c] * patch[p + tap(k)][ci(k)]:  the weight gradient of `first` (wide = dY)
//                            and of `final` (wide = x, thin = d(pre-sigmoid), taps negated by the caller's unpacking)
// Both on v_mfma_f32_32x32x2_f32 with the thin operand gathered from an LDS patch (a 4 x 16 pixel strip + halo 3: 10 x 22
// float4), the wide operand / the packed filter read as they lie.  fp32 throughout.
#include "conv_common.h"

#include <algorithm>

namespace eamm {

namespace {
constexpr int T7_K = 196;                 // 49 taps x 4 channels
constexpr int T7_SH = 4, T7_SW = 16;      // strip of output pixels per step
constexpr int T7_PH = T7_SH + 6, T7_PW = T7_SW + 6;   // patch with halo 3
}  // namespace

struct Thin7Args {
    const float* thin;     // [B,H,W,4]
    const float* wide;     // wgrad: [B,H,W,N]
    const float* wk;       // forward: packed [196][N]
    const float* bias;     // forward: [N] or null
    float* out;            // forward: [B,H,W,N]; wgrad: partial [blocks][N][224]
    int B, H, W, N;        // N = 32 | 64 wide channels
    int strips_x, strips_y, nstrips;
};

// grid: persistent blocks over the strips; a strip is 4 x 16 = 64 pixels = two 32-row MFMA blocks; the four waves split
// (M block, N block): wm = wave >> 1 (strip rows 2 wm, 2 wm + 1), wn = wave & 1 (32 of the N channels; idle when N = 32)
template <int NB>   // N / 32
__global__ __launch_bounds__(256) void conv7_thin_in_kernel(const Thin7Args p) {
    constexpr int N = NB * 32, LDW = N + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ws = smem;                               // [196][LDW]
    float* const Ps = smem + T7_K * LDW;                  // [PH][PW][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < T7_K * (N / 4); i += 256) {     // the packed filter, once per block
        const int k = i / (N / 4), q = i % (N / 4);
        *reinterpret_cast<f32x4*>(Ws + k * LDW + q * 4) = reinterpret_cast<const f32x4*>(p.wk)[i];
    }
    const int m = lane & 31, kk = lane >> 5;
    const int prow = 2 * wm + (m >> 4), pcol = m & 15;    // this lane's pixel in the strip
    for (int sidx = blockIdx.x; sidx < p.nstrips; sidx += gridDim.x) {
        const int sx = sidx % p.strips_x, sy = (sidx / p.strips_x) % p.strips_y, b = sidx / (p.strips_x * p.strips_y);
        const int y0 = sy * T7_SH, x0 = sx * T7_SW;
        __syncthreads();                                  // the previous strip's patch is no longer read (and Ws is written)
        if (tid < T7_PH * T7_PW) {
            const int py = tid / T7_PW, px = tid % T7_PW;
            const int y = y0 + py - 3, x = x0 + px - 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                v = reinterpret_cast<const f32x4*>(p.thin)[((size_t)b * p.H + y) * p.W + x];
            *reinterpret_cast<f32x4*>(Ps + tid * 4) = v;
        }
        __syncthreads();
        f32x16 acc;
        static_for<16>([&](auto rc) { acc[decltype(rc)::value] = 0.f; });
        // waves with wn beyond the N blocks (N = 32: wn = 1) idle in the MFMA loop
        if (wn < NB) {
#pragma unroll 7
            for (int tap = 0; tap < 49; ++tap) {
                const int ky = tap / 7, kx = tap % 7;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(Ps + ((prow + ky) * T7_PW + pcol + kx) * 4);
                const float* wrow = Ws + (tap * 4 + kk) * LDW + wn * 32 + m;
                // k = 4 tap + {0,1} then {2,3}: lane half kk takes channel kk, then 2 + kk
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk ? a4[1] : a4[0], wrow[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk ? a4[3] : a4[2], wrow[2 * LDW], acc, 0, 0, 0);
            }
            // D: lane -> column n = lane % 32; register r -> row (pixel) (r % 4) + 8 (r / 4) + 4 (lane / 32)
            const int n = wn * 32 + m;
            const float bs = p.bias ? p.bias[n] : 0.f;
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int mm = (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int y = y0 + 2 * wm + (mm >> 4), x = x0 + (mm & 15);
                if (y < p.H && x < p.W) p.out[(((size_t)b * p.H + y) * p.W + x) * N + n] = acc[r] + bs;
            });
        }
    }
}

// D[c][k] over the block's strips; waves split the 7 column blocks of k (196 -> 224): wave w owns blocks w and w + 4 (< 7)
template <int NB>
__global__ __launch_bounds__(256) void conv7_thin_wgrad_kernel(const Thin7Args p) {
    constexpr int N = NB * 32, LDA = N + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                               // [64 pixels][LDA]  wide rows of the strip
    float* const Ps = smem + 64 * LDA;                    // [PH][PW][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kk = lane >> 5;
    f32x16 acc[2][NB];                                    // [own column block 0/1][row block]
    static_for<2>([&](auto jc) {
        static_for<NB>([&](auto ic) { static_for<16>([&](auto rc) { acc[decltype(jc)::value][decltype(ic)::value][decltype(rc)::value] = 0.f; }); });
    });
    // column n of block nb: k = nb * 32 + n -> (tap, ci); beyond 196: zero (address 0 of a zeroed pad is not needed: masked)
    int poff[2];                                          // patch offset of this lane's column, per own block
    bool pok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = (wave + 4 * j) * 32 + m;
        pok[j] = (wave + 4 * j) < 7 && k < T7_K;
        const int tap = k >> 2, ci = k & 3;
        poff[j] = ((tap / 7) * T7_PW + tap % 7) * 4 + ci;
    }
    for (int sidx = blockIdx.x; sidx < p.nstrips; sidx += gridDim.x) {
        const int sx = sidx % p.strips_x, sy = (sidx / p.strips_x) % p.strips_y, b = sidx / (p.strips_x * p.strips_y);
        const int y0 = sy * T7_SH, x0 = sx * T7_SW;
        __syncthreads();
        if (tid < T7_PH * T7_PW) {
            const int py = tid / T7_PW, px = tid % T7_PW;
            const int y = y0 + py - 3, x = x0 + px - 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                v = reinterpret_cast<const f32x4*>(p.thin)[((size_t)b * p.H + y) * p.W + x];
            *reinterpret_cast<f32x4*>(Ps + tid * 4) = v;
        }
        for (int i = tid; i < 64 * (N / 4); i += 256) {   // wide rows of the 64 pixels (zero outside the map)
            const int px = i / (N / 4), q = i % (N / 4);
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < p.H && x < p.W) v = reinterpret_cast<const f32x4*>(p.wide)[(((size_t)b * p.H + y) * p.W + x) * (N / 4) + q];
            *reinterpret_cast<f32x4*>(As + px * LDA + q * 4) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {                    // K steps of two pixels
            const int px = 2 * s + kk;
            const int pbase = ((px >> 4) * T7_PW + (px & 15)) * 4;
            float av[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) av[i] = As[px * LDA + i * 32 + m];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (wave + 4 * j < 7) {                   // wave-uniform
                    const float bv = pok[j] ? Ps[pbase + poff[j]] : 0.f;
                    static_for<NB>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[j][i], 0, 0, 0);
                    });
                }
            }
        }
    }
    float* out = p.out + (size_t)blockIdx.x * N * 224;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (wave + 4 * j < 7) {
            static_for<NB>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;          // wide channel c
                    out[(size_t)row * 224 + (wave + 4 * j) * 32 + m] = acc[j][i][r];
                });
            });
        }
    }
}

// packed filter of conv7_thin_in_kernel from OIHW: forward  Wk[(tap, ci)][n] = w[n][ci][tap]      (w: [N,3,7,7]),
// data gradient (transposed = 1)                              Wk[(tap, co)][n] = w[co][n][48 - tap] (w: [3,N,7,7])
__global__ void conv7_thin_pack_kernel(const float* __restrict__ w, int N, int transposed, float* __restrict__ wk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T7_K * N) return;
    const int n = i % N, k = i / N, tap = k >> 2, c = k & 3;
    float v = 0.f;
    if (c < 3) v = transposed ? w[((size_t)c * N + n) * 49 + (48 - tap)] : w[((size_t)n * 3 + c) * 49 + tap];
    wk[i] = v;
}

// sum of the blocks' partials -> OIHW.  thin_is_input = 1: dW[n][ci][tap] (first: [N,3,7,7]); 0: the thin side is the OUTPUT
// gradient and the patch offset runs against the filter offset: dW[co][n][48 - tap] (final: [3,N,7,7])
__global__ void conv7_thin_wgrad_reduce_kernel(const float* __restrict__ partial, int blocks, int N, int thin_is_input,
                                               float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 3 * 49) return;
    const int tap = i % 49, c = (i / 49) % 3, n = i / (49 * 3);
    const int k = tap * 4 + c;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < blocks; ++b) s += partial[((size_t)b * N + n) * 224 + k];
    if (thin_is_input) dw[((size_t)n * 3 + c) * 49 + tap] = s;
    else dw[((size_t)c * N + n) * 49 + (48 - tap)] = s;
}

namespace {
int thin7_blocks(int nstrips) { return std::min(nstrips, 1024); }
}  // namespace

size_t conv7_thin_workspace_floats(int B, int H, int W, int N) {
    const int nstrips = B * ((H + T7_SH - 1) / T7_SH) * ((W + T7_SW - 1) / T7_SW);
    return std::max<size_t>((size_t)T7_K * N, (size_t)thin7_blocks(nstrips) * N * 224);
}

// y = conv7x7_same(thin [B,H,W,4], w) (+ bias) -> [B,H,W,N]; transposed: w is the [3,N,7,7] filter of the N -> 3 layer whose
// data gradient this computes.  workspace >= conv7_thin_workspace_floats.
hipError_t conv7_thin_in_launch(const float* thin, const float* w, const float* bias, int B, int H, int W, int N, int transposed,
                                float* out, float* workspace, hipStream_t s) {
    if (N != 32 && N != 64) return hipErrorInvalidValue;
    Thin7Args a{};
    a.thin = thin;
    a.wk = workspace;
    a.bias = bias;
    a.out = out;
    a.B = B; a.H = H; a.W = W; a.N = N;
    a.strips_x = (W + T7_SW - 1) / T7_SW;
    a.strips_y = (H + T7_SH - 1) / T7_SH;
    a.nstrips = B * a.strips_x * a.strips_y;
    hipLaunchKernelGGL(conv7_thin_pack_kernel, dim3((T7_K * N + 255) / 256), dim3(256), 0, s, w, N, transposed, workspace);
    const size_t lds = sizeof(float) * ((size_t)T7_K * (N + 4) + T7_PH * T7_PW * 4);
    const int blocks = std::min(a.nstrips, 2048);
    note_mfma_flops(2.0 * T7_K * (double)N * B * H * W);        // K = (tap, channel) = 196 per output pixel and wide channel
    hipError_t e;
    if (N == 64) {
        static lds_once_mask configured{0};
        if ((e = ensure_dynamic_lds(conv7_thin_in_kernel<2>, lds, &configured)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv7_thin_in_kernel<2>, dim3(blocks), dim3(256), lds, s, a);
    } else {
        static lds_once_mask configured{0};
        if ((e = ensure_dynamic_lds(conv7_thin_in_kernel<1>, lds, &configured)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv7_thin_in_kernel<1>, dim3(blocks), dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

// weight gradient: thin_is_input = 1: dW [N,3,7,7] of y = conv(thin) from wide = dY; 0: dW [3,N,7,7] of y3 = conv(wide) from
// thin = d y3 (fourth channel zero)
hipError_t conv7_thin_wgrad_launch(const float* thin, const float* wide, int B, int H, int W, int N, int thin_is_input, float* dw,
                                   float* workspace, hipStream_t s) {
    if (N != 32 && N != 64) return hipErrorInvalidValue;
    Thin7Args a{};
    a.thin = thin;
    a.wide = wide;
    a.out = workspace;
    a.B = B; a.H = H; a.W = W; a.N = N;
    a.strips_x = (W + T7_SW - 1) / T7_SW;
    a.strips_y = (H + T7_SH - 1) / T7_SH;
    a.nstrips = B * a.strips_x * a.strips_y;
    const int blocks = thin7_blocks(a.nstrips);
    const size_t lds = sizeof(float) * ((size_t)64 * (N + 4) + T7_PH * T7_PW * 4);
    note_mfma_flops(2.0 * T7_K * (double)N * B * H * W);
    if (N == 64) hipLaunchKernelGGL(conv7_thin_wgrad_kernel<2>, dim3(blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv7_thin_wgrad_kernel<1>, dim3(blocks), dim3(256), lds, s, a);
    hipLaunchKernelGGL(conv7_thin_wgrad_reduce_kernel, dim3((N * 147 + 255) / 256), dim3(256), 0, s, workspace, blocks, N, thin_is_input, dw);
    return hipGetLastError();
}

}  // namespace eamm
