// The generator's first block: SameBlock2d 7x7, 3 -> Cout, BatchNorm folded, ReLU (reference modules/generator.py:61,
// modules/util.py:923-938) straight from the NCHW RGB source.
//
// With three input channels the reduction is 7 * 7 * 3 = 147 long.  The generic 7x7 kernel pads the channels to a 32-wide
// chunk per tap (K = 1568, ten times the work); here K is enumerated as k = 4 * tap + c (c = 3: a zero weight row), 196
// deep, and the implicit-GEMM A operand is gathered from a (16 + 6)^2-pixel patch of the three colour planes held in LDS:
//   * workgroup = 8 waves = a 16 x 16 pixel tile x all Cout channels; wave w owns pixel rows 2w, 2w + 1 (32 pixels);
//   * K step s of v_mfma_f32_32x32x2_f32 covers k = 2 s + half: tap = s >> 1, c = 2 (s & 1) + half -- the lane's half picks
//     the colour plane by a constant added once to its base address, the tap is a compile-time displacement: every A value
//     is one ds_read_b32 with an immediate offset, no address arithmetic in the loop;
//   * the patch planes use a row pitch of 48 floats: the 16 pixels of the wave's second row then sit 16 banks from the
//     first row's, so the 32 lanes of a ds_read_b32 group touch 32 different banks;
//   * the folded weights ([196][Cout], 49 KiB at Cout = 64) are copied to LDS once per workgroup.
// fp32 throughout; only the summation order differs from the reference.
#include "kernels.h"
#include "conv_common.h"

namespace eamm {

namespace {
constexpr int FT = 16;                  // tile side
constexpr int FP = FT + 6;              // patch side (halo 3)
constexpr int FPITCH = 48;              // floats per patch row (see above)
constexpr int FPLANE = FP * FPITCH;     // floats per colour plane
constexpr int FK = 196;                 // 49 taps x 4 (three colours + one zero row)
constexpr int FWAVES = 8;
}  // namespace

struct First7Args {
    const float* src;      // [ns,3,H,W]
    const float* w;        // [196][Cout] folded weights, row k = 4 * (dy * 7 + dx) + c
    const float* bias;     // [Cout]
    int H, W, Cout;
    int tiles_x, tiles_y;
    float* out;            // [ns,H,W,Cout]
};

template <int NT>
__global__ __launch_bounds__(FWAVES * 64) void conv_first7_kernel(const First7Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][FP][FPITCH] patch planes, [196][Cout] weights
    float* const patch = smem;
    float* const wl = smem + 4 * FPLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int L = blockIdx.x;
    const int tx0 = (L % p.tiles_x) * FT;
    L /= p.tiles_x;
    const int ty0 = (L % p.tiles_y) * FT;
    const int b = L / p.tiles_y;
    const int Cout = NT * 32;

    // ---- stage the patch (zero padding outside the image; the fourth plane only has to be finite) and the weights
    const float* img = p.src + (size_t)b * 3 * p.H * p.W;
    for (int i = tid; i < 4 * FP * FP; i += FWAVES * 64) {
        const int c = i / (FP * FP), r = i - c * (FP * FP);
        const int py = r / FP, px = r - py * FP;
        const int y = ty0 + py - 3, x = tx0 + px - 3;
        float v = 0.f;
        if (c < 3 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) v = img[((size_t)c * p.H + y) * p.W + x];
        patch[c * FPLANE + py * FPITCH + px] = v;
    }
    for (int i = tid; i < FK * Cout / 4; i += FWAVES * 64)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();

    f32x16 acc[NT];
    static_for<NT>([&](auto jc) {
        static_for<16>([&](auto rc) { acc[decltype(jc)::value][decltype(rc)::value] = 0.f; });
    });
    // lane bases: pixel (2 wave + (l31 >> 4), l31 & 15) of the tile, colour plane `half` (+ 2 on odd K steps);
    // weight row `half` (+ 2 s), column l31 (+ 32 j)
    const float* const a0 = patch + half * FPLANE + (2 * wave + (l31 >> 4)) * FPITCH + (l31 & 15);
    const float* const b0 = wl + half * Cout + l31;
    static_for<FK / 2>([&](auto sc) {
        constexpr int s = decltype(sc)::value, tap = s >> 1, dy = tap / 7, dx = tap % 7;
        const float a = a0[2 * (s & 1) * FPLANE + dy * FPITCH + dx];
        static_for<NT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[2 * s * Cout + j * 32], acc[j], 0, 0, 0);
        });
    });

    // ---- epilogue: bias + ReLU, 128-byte runs per (pixel, 32-channel group)
    static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n = j * 32 + l31;
        const float bias = p.bias[n];
        static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;      // MFMA row of this register
            const int y = ty0 + 2 * wave + (m >> 4), x = tx0 + (m & 15);
            if (y < p.H && x < p.W) p.out[(((size_t)b * p.H + y) * p.W + x) * p.Cout + n] = fmaxf(acc[j][r] + bias, 0.f);
        });
    });
}

// w: [Cout][3][7][7] with the BatchNorm scale folded in -> [196][Cout], row k = 4 * tap + c (c = 3: zeros)
void first7_pack_host(const float* w, int Cout, float* dst) {
    for (size_t i = 0; i < (size_t)FK * Cout; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < 3; ++c)
            for (int t = 0; t < 49; ++t) dst[(size_t)(4 * t + c) * Cout + o] = w[((size_t)o * 3 + c) * 49 + t];
}

bool first7_supported(int Cout) { return Cout == 32 || Cout == 64 || Cout == 96 || Cout == 128; }

hipError_t first7_launch(const float* src, const float* w_packed, const float* bias, int ns, int H, int W, int Cout, float* out,
                         hipStream_t stream) {
    if (!first7_supported(Cout)) return hipErrorInvalidValue;
    First7Args a{};
    a.src = src;
    a.w = w_packed;
    a.bias = bias;
    a.H = H;
    a.W = W;
    a.Cout = Cout;
    a.tiles_x = (W + FT - 1) / FT;
    a.tiles_y = (H + FT - 1) / FT;
    a.out = out;
    const size_t lds = sizeof(float) * (4 * FPLANE + (size_t)FK * Cout);
    const int blocks = a.tiles_x * a.tiles_y * ns;
    auto go = [&](auto kern, lds_once_mask* configured) -> hipError_t {
        if (hipError_t e = ensure_dynamic_lds(kern, lds, configured); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FWAVES * 64), lds, stream, a);
        return hipGetLastError();
    };
    static lds_once_mask cfg[4];
    switch (Cout / 32) {
        case 1: return go(conv_first7_kernel<1>, &cfg[0]);
        case 2: return go(conv_first7_kernel<2>, &cfg[1]);
        case 3: return go(conv_first7_kernel<3>, &cfg[2]);
        default: return go(conv_first7_kernel<4>, &cfg[3]);
    }
}

}  // namespace eamm
