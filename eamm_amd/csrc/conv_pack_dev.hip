// Device-side filter packing for the op-level convolution entry that takes DEVICE weights (eamm_op_conv_dev): the training path
// (eamm_amd/autograd_ops.py) calls a convolution with parameters that change every optimiser step, so the host-side repack of
// eamm_op_conv (device -> host copy, scalar loop, upload) would dominate the step.  Same layouts as conv_pack_host
// (conv_mfma.hip, register-staged kernel: [ntiles][taps * Cin/32][BN][32]) and wino4_pack_host (conv_winograd4.hip:
// U = G g G^T, [ntiles][36 * Cin/32][BN][32] with XOR-swizzled 16-byte slots), from an OIHW tensor in HBM -- or, with
// `transposed`, from the FORWARD filter [Cin][Cout][kh][kw] read transposed over (out, in) and flipped over (y, x): the filter
// of the data gradient (reference: autograd of nn.Conv2d, modules/util.py:858-938), with no intermediate tensor.
#include "conv_common.h"

#include <algorithm>

namespace eamm {

__global__ __launch_bounds__(256) void conv_pack_dev_kernel(const float* __restrict__ w, int Cout, int Cin, int T, int BN, int cin_packed,
                                                            int transposed, float* __restrict__ dst, size_t total) {
    const int nchunks = T * (cin_packed / CONV_BK);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int kl = (int)(idx % CONV_BK);
        const int nl = (int)((idx / CONV_BK) % BN);
        const int ci = (int)((idx / ((size_t)CONV_BK * BN)) % nchunks);
        const int nt = (int)(idx / ((size_t)CONV_BK * BN * nchunks));
        const int o = nt * BN + nl, c = (ci / T) * CONV_BK + kl, tap = ci % T;
        float v = 0.f;
        if (o < Cout && c < Cin)
            v = transposed ? w[((size_t)c * Cout + o) * T + (T - 1 - tap)] : w[((size_t)o * Cin + c) * T + tap];
        dst[idx] = v;
    }
}

// one thread per (padded output channel, input channel): the 36 transform points of its 3x3 filter, in double like the host pack
__global__ __launch_bounds__(256) void wino4_pack_dev_kernel(const float* __restrict__ w, int Cout, int Cin, int BN, int ntiles,
                                                             int transposed, float* __restrict__ dst) {
    const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int cch = Cin / CONV_BK;
    const size_t total = (size_t)ntiles * BN * Cin;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cin);
        const int o = (int)(idx / Cin);
        const int nt = o / BN, nl = o % BN;
        double g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
                if (o < Cout) v = transposed ? w[((size_t)c * Cout + o) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((size_t)o * Cin + c) * 9 + ky * 3 + kx];
                g[ky][kx] = (double)v;
            }
        double tmp[6][3];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) tmp[i][k] = G[i][0] * g[0][k] + G[i][1] * g[1][k] + G[i][2] * g[2][k];
        const int cc = c / CONV_BK, kl = c % CONV_BK;
        const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                dst[(((size_t)nt * 36 * cch + (size_t)(i * 6 + j) * cch + cc) * BN + nl) * CONV_BK + kk] = (float)u;
            }
    }
}

// the final layer's column-patch filter (conv_col7.hip; eamm_op_conv tile 4002 packs the same on the host): the 7x7, Cout = 3
// filter as a 7x1 filter with N = (dx, co) = 21 of 32 columns, w'[dx*3+co][c][dy] = w[co][c][dy][dx], LDS-DMA (swizzled) layout
__global__ __launch_bounds__(256) void col7_pack_dev_kernel(const float* __restrict__ w, int C, float* __restrict__ dst) {
    const int total = 7 * (C / CONV_BK) * 32 * CONV_BK;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kl = idx % CONV_BK, nl = (idx / CONV_BK) % 32, ci = idx / (CONV_BK * 32);
        const int cc = ci / 7, dy = ci % 7, c = cc * CONV_BK + kl;
        float v = 0.f;
        if (nl < 21) v = w[(((size_t)(nl % 3) * C + c) * 7 + dy) * 7 + nl / 3];
        const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
        dst[((size_t)ci * 32 + nl) * CONV_BK + kk] = v;
    }
}

__global__ void bias_pad_dev_kernel(const float* __restrict__ b, int Cout, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (b != nullptr && i < Cout) ? b[i] : 0.f;
}

hipError_t conv_pack_dev_launch(const float* w, int Cout, int Cin, int T, int BN, int transposed, float* dst, hipStream_t s) {
    const int cin_packed = (Cin + CONV_BK - 1) / CONV_BK * CONV_BK;
    const size_t total = conv_packed_elems(T, cin_packed, Cout, BN, 1);
    hipLaunchKernelGGL(conv_pack_dev_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16)), dim3(256), 0, s, w, Cout, Cin, T,
                       BN, cin_packed, transposed, dst, total);
    return hipGetLastError();
}

hipError_t wino4_pack_dev_launch(const float* w, int Cout, int Cin, int BN, int transposed, float* dst, hipStream_t s) {
    if (Cin % CONV_BK) return hipErrorInvalidValue;
    const int ntiles = (Cout + BN - 1) / BN;
    const size_t total = (size_t)ntiles * BN * Cin;
    hipLaunchKernelGGL(wino4_pack_dev_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16)), dim3(256), 0, s, w, Cout, Cin, BN,
                       ntiles, transposed, dst);
    return hipGetLastError();
}

hipError_t col7_pack_dev_launch(const float* w, int C, float* dst, hipStream_t s) {
    if (C % CONV_BK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(col7_pack_dev_kernel, dim3((7 * C * 32 + 255) / 256), dim3(256), 0, s, w, C, dst);
    return hipGetLastError();
}

hipError_t bias_pad_dev_launch(const float* b, int Cout, int n, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(bias_pad_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, s, b, Cout, n, dst);
    return hipGetLastError();
}

}  // namespace eamm
