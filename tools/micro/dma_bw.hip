// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/dma_bw.hip -o tools/micro/dma_bw
// Microbenchmark: LDS-DMA (buffer_load ... lds) streaming rate per CU when every CU streams at once.
//   dma_bw <group> <region_KB> <inflight> <iters> [mode]
// group   consecutive blocks (same XCD after the remap) that read the SAME stream (1 = private streams)
// region  bytes of one stream before it wraps (small -> L2 resident)
// inflight  1 KiB pieces each wave keeps outstanding (vmcnt depth)
// mode 0: LDS-DMA, 1: buffer_load to VGPR + ds_write_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int INFLIGHT, int MODE>
__global__ __launch_bounds__(512) void dma_kernel(const float* src, unsigned total_bytes, int group, unsigned region,
                                                  int pieces_per_wave, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned stream = (unsigned)(L / group) * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, total_bytes, 0x00020000);
    float* dst = smem + wave * (INFLIGHT * 256);   // INFLIGHT KiB per wave
    unsigned pos = (unsigned)wave * 1024u;           // the 8 waves interleave 1 KiB pieces of the stream
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces_per_wave; i += INFLIGHT) {
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            const unsigned off = stream + (pos % region) + lane * 16u;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + k * 256), 16, off, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc += v;
            }
            pos += 8 * 1024u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1) *reinterpret_cast<u32x4*>(dst + lane * 4) = acc;
    __syncthreads();
    if (sink != nullptr && smem[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

template <int MODE>
void run(int inflight, dim3 g, size_t lds, const float* src, unsigned tb, int group, unsigned region, int ppw) {
    switch (inflight) {
        case 1: hipLaunchKernelGGL((dma_kernel<1, MODE>), g, dim3(512), lds, 0, src, tb, group, region, ppw, nullptr); break;
        case 2: hipLaunchKernelGGL((dma_kernel<2, MODE>), g, dim3(512), lds, 0, src, tb, group, region, ppw, nullptr); break;
        case 4: hipLaunchKernelGGL((dma_kernel<4, MODE>), g, dim3(512), lds, 0, src, tb, group, region, ppw, nullptr); break;
        case 8: hipLaunchKernelGGL((dma_kernel<8, MODE>), g, dim3(512), lds, 0, src, tb, group, region, ppw, nullptr); break;
        default: hipLaunchKernelGGL((dma_kernel<16, MODE>), g, dim3(512), lds, 0, src, tb, group, region, ppw, nullptr); break;
    }
}

int main(int argc, char** argv) {
    const int group = argc > 1 ? atoi(argv[1]) : 1;
    const unsigned region = (argc > 2 ? atoi(argv[2]) : 2048) * 1024u;
    const int inflight = argc > 3 ? atoi(argv[3]) : 8;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    const int blocks = 256;
    const int ppw = 2048;   // 2 MiB per wave, 16 MiB per block
    const int nstreams = (blocks + group - 1) / group;
    const size_t total = (size_t)nstreams * region;
    if (total >= 0xFFFFFFF0ull) { printf("too large\n"); return 1; }
    float* src;
    hipMalloc((void**)&src, total);
    hipMemset(src, 0, total);
    const size_t lds = 128 * 1024;
    hipFuncSetAttribute((const void*)dma_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)dma_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto go = [&]() {
        if (mode == 0) run<0>(inflight, dim3(blocks), lds, src, (unsigned)total, group, region, ppw);
        else run<1>(inflight, dim3(blocks), lds, src, (unsigned)total, group, region, ppw);
    };
    go();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double bytes = (double)blocks * 8 * ppw * 1024.0;
    printf("mode=%d group=%3d region=%6u KB inflight=%2d/wave: %8.1f us  %7.2f TB/s chip  %6.2f B/clk/CU @2.4GHz  (%s)\n", mode,
           group, region / 1024, inflight, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3) / 2.4e9,
           hipGetErrorString(hipGetLastError()));
    return 0;
}
