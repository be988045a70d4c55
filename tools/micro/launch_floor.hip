// Round 6: what ONE dependent kernel boundary costs on this box -- the one-frame path (BASELINE configs[1]) is ~75 dependent launches
// per frame, and the smallest kernels of its trace (kp_prepare, warp, split-K reduce) all last 4.5 - 5.5 us whatever they do.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/launch_floor.hip -o tools/micro/launch_floor && tools/micro/launch_floor
// Chains of N launches on one stream, wall clock per launch: (a) an empty kernel, (b) one block touching one float, (c) 256 blocks
// writing `MB` MiB that the next launch reads (so the end-of-kernel write-back and the next kernel's first misses are in the figure),
// each also replayed from a HIP graph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[0] += 1.f; }
__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        v.x += 1.f;
        out[i] = v;
    }
}

template <typename F> static double chain(hipStream_t s, int n, F launch) {
    for (int i = 0; i < 50; ++i) launch(i);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
template <typename F> static double graphed(hipStream_t s, int n, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (double(n) * reps);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return us;
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    const int N = 2000;
    float* p; hipMalloc(&p, 4096);
    hipMemset(p, 0, 4096);
    printf("empty kernel, 1 block x 64:            eager %.2f us / launch, graph %.2f\n", chain(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }),
           graphed(s, 200, 10, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }));
    printf("empty kernel, 256 blocks x 256:        eager %.2f us / launch, graph %.2f\n", chain(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }),
           graphed(s, 200, 10, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }));
    printf("one float read-modify-write, 1 block:  eager %.2f us / launch, graph %.2f\n", chain(s, N, [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, p); }),
           graphed(s, 200, 10, [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, p); }));
    for (size_t mb : {1, 4, 16, 64}) {
        const size_t n4 = mb * (1u << 20) / 16;
        float4 *a, *b; hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16);
        hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
        auto l = [&](int i) { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n4); };
        const double e = chain(s, N, l), g = graphed(s, 200, 10, l);
        printf("ping-pong copy of %2zu MiB, 1024 blocks:  eager %.2f us / launch (%.2f TB/s), graph %.2f\n", mb, e, 2.0 * mb * 1.048576 / e, g);
        hipFree(a); hipFree(b);
    }
    return 0;
}
