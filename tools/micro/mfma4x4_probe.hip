// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950 (round 4: the final 7x7 layer's N = 21 fits a 4-wide N granule; conv_col7.hip).
//  1. lane layout with the A-block broadcast (cbsz = 4, abid = g): D[i][lane p] ?= A[lane 4 g + i] * B[lane p]
//  2. issue rate: cycles per instruction with 3 / 6 rotating accumulators, against v_mfma_f32_32x32x2_f32
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma4x4_probe tools/micro/mfma4x4_probe.hip && tools/micro/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(float* out) {   // out[abid 0..15][i 0..3][lane 0..63]
    const int lane = threadIdx.x;
    const float a = 100.f + lane;       // A operand: lane l holds A[block l / 4][row l % 4]
    const float b = 1.f + 0.001f * lane; // B operand: lane l holds B[block l / 4][col l % 4]
#define ONE(G)                                                                        \
    {                                                                                 \
        f32x4 c = {0.f, 0.f, 0.f, 0.f};                                               \
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, G, 0);                     \
        for (int i = 0; i < 4; ++i) out[(G * 4 + i) * 64 + lane] = c[i];              \
    }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
#undef ONE
    {   // no broadcast for comparison: out[16*4*64 + i*64 + lane]
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        for (int i = 0; i < 4; ++i) out[(64 + i) * 64 + lane] = c[i];
    }
}

template <int NACC>
__global__ void rate4_kernel(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (j % 6 == 0) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 0, 0);
                if (j % 6 == 1) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 1, 0);
                if (j % 6 == 2) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 2, 0);
                if (j % 6 == 3) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 3, 0);
                if (j % 6 == 4) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 4, 0);
                if (j % 6 == 5) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 4, 5, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void rate32_kernel(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* d;
    long long* dc;
    hipMalloc(&d, 1 << 22);
    hipMalloc(&dc, 64);
    layout_kernel<<<1, 64>>>(d);
    std::vector<float> h(68 * 64);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // expectation: out[g][i][p] = A[4g + i] * B[p] = (100 + 4g + i) * (1 + 0.001 p)
    int bad = 0;
    for (int g = 0; g < 16; ++g)
        for (int i = 0; i < 4; ++i)
            for (int p = 0; p < 64; ++p) {
                const float want = (100.f + 4 * g + i) * (1.f + 0.001f * p);
                if (fabsf(h[(g * 4 + i) * 64 + p] - want) > 1e-3f) ++bad;
            }
    printf("layout with cbsz=4: D[vgpr i][lane p] == A[lane 4*abid+i] * B[lane p] : %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) {
        for (int g = 0; g < 2; ++g)
            for (int i = 0; i < 4; ++i) {
                printf("abid %d vgpr %d:", g, i);
                for (int p = 0; p < 12; ++p) printf(" %.3f", h[(g * 4 + i) * 64 + p]);
                printf("\n");
            }
    }
    int bad0 = 0;   // no broadcast: D[i][lane p] = A[lane 4*(p/4) + i] * B[lane p]
    for (int i = 0; i < 4; ++i)
        for (int p = 0; p < 64; ++p)
            if (fabsf(h[(64 + i) * 64 + p] - (100.f + 4 * (p / 4) + i) * (1.f + 0.001f * p)) > 1e-3f) ++bad0;
    printf("layout without broadcast: D[i][p] == A[4*(p/4)+i] * B[p] : %s (%d mismatches)\n", bad0 ? "NO" : "yes", bad0);
    const int iters = 20000;
    auto report = [&](const char* name, int per_iter, double flops_per_instr) {
        long long c = 0;
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.2f cycles / instruction (%5.1f FLOP / cycle / SIMD)\n", name, (double)c / iters / per_iter,
               flops_per_instr / ((double)c / iters / per_iter));
    };
    rate4_kernel<3><<<1, 64>>>(d, dc, iters);  hipDeviceSynchronize(); report("4x4x1_16b, 3 rotating accumulators, 1 wave", 12, 512.0);
    rate4_kernel<6><<<1, 64>>>(d, dc, iters);  hipDeviceSynchronize(); report("4x4x1_16b, 6 rotating accumulators, 1 wave", 24, 512.0);
    rate4_kernel<3><<<1, 512>>>(d, dc, iters); hipDeviceSynchronize(); report("4x4x1_16b, 3 accumulators, 2 waves / SIMD", 12, 512.0);
    rate4_kernel<6><<<1, 512>>>(d, dc, iters); hipDeviceSynchronize(); report("4x4x1_16b, 6 accumulators, 2 waves / SIMD", 24, 512.0);
    rate32_kernel<<<1, 64>>>(d, dc, iters);    hipDeviceSynchronize(); report("32x32x2, 2 accumulators, 1 wave", 8, 4096.0);
    rate32_kernel<<<1, 512>>>(d, dc, iters);   hipDeviceSynchronize(); report("32x32x2, 2 accumulators, 2 waves / SIMD", 8, 4096.0);
    return 0;
}
