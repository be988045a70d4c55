// Temporal smoothing of a clip's driving key points on the device.
//
// Reference: make_animation_smooth filters every frame's key points through filter1.OneEuroFilter between its two loops
// (demo.py:231-250: `process(kp.cpu() * 10) / 10`, one filter for the values, one for the jacobians; filter1.py:13-47) --
// a host loop of tiny tensor operations per frame with two device round trips each.  The recurrence is sequential in t and
// independent per element, so here ONE thread owns one element (K*2 + K*4 = 60 per clip at K = 10) and walks the T frames:
//     dx_t   = (x_t - x_{t-1}) * freq                       (0 for t = 0)
//     edx_t  = a_d * dx_t + (1 - a_d) * edx_{t-1}           (dx_0 for t = 0),  a_d = alpha(dcutoff)
//     s_t    = a_t * x_t + (1 - a_t) * s_{t-1}              (x_0 for t = 0),   a_t = alpha(mincutoff + beta * |edx_t|)
//     alpha(c) = 1 / (1 + (1 / (2 pi c)) / te),  te = 1 / freq
// with x = input * scale and output s / scale, every step in float32 in the reference's operation order (separate multiplies
// and adds, true divisions where ATen's CPU kernels divide: no fused multiply-add, so the result is the host filter's bit for bit).  A frame's 60 values are loaded
// sixteen frames ahead of the dependent chain (the loads do not depend on it), so the walk costs the chain's ~40 dependent
// float operations per frame (two correctly rounded reciprocals among them: ~380 cycles), not a memory round trip per frame:
// 2048 frames in 0.32 ms (rocprofv3, profiles/r05_e2e_kernel_trace_stats.txt) where the host loop takes 600 ms.
#include "kernels.h"

namespace eamm {

constexpr int EURO_AHEAD = 16;

// hipcc contracts a * b + c into a fused multiply-add by default, and HIP's __fmul_rn / __fadd_rn are plain operators to it: left
// alone, WHERE it fuses depends on the position in the unrolled walk (steps are paired for v_pk_mul_f32; one of a pair gets the fused
// form) -- a clip filtered in chunks then differs from the clip filtered whole in the last bit (measured: cuts at odd frames), and
// from the reference's separately rounded operations.  THIS FILE IS COMPILED WITH -ffp-contract=off (csrc/Makefile).

static __device__ __forceinline__ float euro_alpha(float cutoff, float te) {
    // The reference filters CPU tensors (demo.py:245 `.cpu() * 10`): ATen evaluates 1.0 / (2 * np.pi * cutoff) as
    // reciprocal(cutoff * float(2 pi)) and `tau / te` as a TRUE float32 division by float(te) (its CPU kernel divides; only the CUDA
    // kernel multiplies by the reciprocal -- round 5 followed that one and was 1-2 ulp off the reference filter)
    const float tau = __frcp_rn(__fmul_rn(cutoff, 6.283185307179586f));
    return __frcp_rn(__fadd_rn(1.0f, __fdiv_rn(tau, te)));
}

// (x and out carry no __restrict__: include/eamm_hip.h allows out == x; a thread reads its element sixteen frames ahead into
//  registers before it writes those frames, so the in-place form is well defined)
__global__ __launch_bounds__(64) void one_euro_kernel(const float* x, int T, int E, float mincutoff, float beta,
                                                     float a_d, float one_m_ad, float freq, float te, float scale,
                                                     float* out, float* __restrict__ state, int resume) {
    // state (optional, [3][E]): the filter's memory -- previous scaled input, previous filtered value, previous filtered
    // derivative -- written at the end; with `resume` it is read first and frame 0 of this call continues the sequence (a clip
    // filtered in chunks gives the bits of the clip filtered whole)
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    float prev_x = 0.f, prev_s = 0.f, prev_edx = 0.f;
    if (resume) {
        prev_x = state[e];
        prev_s = state[E + e];
        prev_edx = state[2 * E + e];
    }
    for (int t0 = 0; t0 < T; t0 += EURO_AHEAD) {
        float buf[EURO_AHEAD];
#pragma unroll
        for (int i = 0; i < EURO_AHEAD; ++i) buf[i] = (t0 + i < T) ? x[(size_t)(t0 + i) * E + e] : 0.f;
#pragma unroll
        for (int i = 0; i < EURO_AHEAD; ++i) {
            const int t = t0 + i;
            if (t >= T) break;
            const float xv = __fmul_rn(buf[i], scale);
            float s, edx;
            if (t == 0 && !resume) {          // first sample: dx = 0 and both low-pass filters pass their input through (filter1.py:19-21, 41-42)
                edx = 0.f;
                s = xv;
            } else {
                const float dx = __fmul_rn(__fsub_rn(xv, prev_x), freq);
                edx = __fadd_rn(__fmul_rn(a_d, dx), __fmul_rn(one_m_ad, prev_edx));
                const float a = euro_alpha(__fadd_rn(mincutoff, __fmul_rn(beta, fabsf(edx))), te);
                s = __fadd_rn(__fmul_rn(a, xv), __fmul_rn(__fsub_rn(1.0f, a), prev_s));
            }
            prev_x = xv;
            prev_s = s;
            prev_edx = edx;
            out[(size_t)t * E + e] = __fdiv_rn(s, scale);   // `/ scale` on a CPU tensor: a true float32 division by float(scale)
        }
    }
    if (state != nullptr) {
        state[e] = prev_x;
        state[E + e] = prev_s;
        state[2 * E + e] = prev_edx;
    }
}

hipError_t one_euro_launch(const float* x, int T, int E, float mincutoff, float beta, float dcutoff, float freq, float scale,
                           float* out, hipStream_t stream, float* state, int resume) {
    if (!x || !out || T < 0 || E < 1 || !(freq > 0.f) || !(dcutoff > 0.f) || !(scale != 0.f) || (resume && !state)) return hipErrorInvalidValue;
    if (T == 0) return hipSuccess;
    // alpha(dcutoff) is a Python-float (double) computation in the reference (filter1.py:35-38 on scalars), rounded when it meets the tensor
    const double te = 1.0 / (double)freq;
    const double tau_d = 1.0 / (2.0 * 3.14159265358979323846 * (double)dcutoff);
    const double a_dd = 1.0 / (1.0 + tau_d / te);
    const float a_d = (float)a_dd, one_m_ad = (float)(1.0 - a_dd);   // `(1.0 - a_d)` is a double subtraction there too
    // tensor / Python-scalar on the CPU: a division by the scalar cast to float
    hipLaunchKernelGGL(one_euro_kernel, dim3((E + 63) / 64), dim3(64), 0, stream, x, T, E, mincutoff, beta, a_d, one_m_ad, freq, (float)te, scale,
                       out, state, resume);
    return hipGetLastError();
}

}  // namespace eamm
