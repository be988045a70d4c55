// C ABI of the training-mode BatchNorm forward and backward (include/eamm_hip.h, row N4; kernels in batchnorm.hip).
// Stateless: the caller owns every buffer and selects the device; errors are reported per thread.
#include "../../include/eamm_hip.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdio>
#include <string>

using namespace eamm;

namespace {
thread_local std::string g_bn_error;

int bn_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_bn_error = buf;
    return code;
}

int bn_check(hipError_t e, const char* what) {
    return e == hipSuccess ? EAMM_OK : bn_fail(EAMM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
}
}  // namespace

extern "C" {

const char* eamm_bn_last_error(void) { return g_bn_error.c_str(); }

size_t eamm_bn_workspace_floats(int N, int C, int HW) {
    if (N < 1 || C < 1 || HW < 1) return 0;
    return bn_workspace_floats(N, C, HW);
}

int eamm_bn_local_sums(const float* x, int N, int C, int HW, float* sums, float* workspace, void* stream) {
    if (!x || !sums || !workspace) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (N < 1 || C < 1 || HW < 1) return bn_fail(EAMM_ERR_ARG, "empty tensor [%d,%d,%d]", N, C, HW);
    if ((long long)N * HW >= (1ll << 36)) return bn_fail(EAMM_ERR_ARG, "more than 2^36 elements per channel");
    return bn_check(bn_local_sums_launch(x, N, C, HW, sums, workspace, reinterpret_cast<hipStream_t>(stream)), "bn_local_sums");
}

int eamm_bn_finalize(const float* sums, int C, float eps, float momentum, int mode, const float* weight, float* running_mean,
                     float* running_var, float* mean, float* scale, float* inv_std, void* stream) {
    if (!running_mean || !running_var || !mean || !scale || (mode != EAMM_BN_EVAL && !sums))
        return bn_fail(EAMM_ERR_ARG, "null argument");
    if (C < 1 || mode < 0 || mode > 2) return bn_fail(EAMM_ERR_ARG, "bad channel count or mode");
    return bn_check(bn_finalize_launch(sums, C, eps, momentum, mode, weight, running_mean, running_var, mean, scale, inv_std,
                                       reinterpret_cast<hipStream_t>(stream)), "bn_finalize");
}

int eamm_bn_apply(const float* x, const float* mean, const float* scale, const float* bias, int N, int C, int HW, float* y,
                  void* stream) {
    if (!x || !mean || !scale || !y) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (N < 1 || C < 1 || HW < 1) return bn_fail(EAMM_ERR_ARG, "empty tensor [%d,%d,%d]", N, C, HW);
    return bn_check(bn_apply_launch(x, mean, scale, bias, N, C, HW, y, reinterpret_cast<hipStream_t>(stream)), "bn_apply");
}

int eamm_bn_backward_sums(const float* x, const float* dy, const float* mean, int N, int C, int HW, float* sums, float* workspace,
                          void* stream) {
    if (!x || !dy || !mean || !sums || !workspace) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (N < 1 || C < 1 || HW < 1) return bn_fail(EAMM_ERR_ARG, "empty tensor [%d,%d,%d]", N, C, HW);
    if ((long long)N * HW >= (1ll << 36)) return bn_fail(EAMM_ERR_ARG, "more than 2^36 elements per channel");
    return bn_check(bn_bwd_sums_launch(x, dy, mean, N, C, HW, sums, workspace, reinterpret_cast<hipStream_t>(stream)), "bn_backward_sums");
}

int eamm_bn_backward_finalize(const float* local_sums, const float* reduced_sums, int C, const float* inv_std, const float* weight,
                              float eps, int mode, float* dweight, float* dbias, float* coef, void* stream) {
    if (!local_sums || !reduced_sums || !inv_std || !coef) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (C < 1 || mode < 0 || mode > 2) return bn_fail(EAMM_ERR_ARG, "bad channel count or mode");
    return bn_check(bn_bwd_finalize_launch(local_sums, reduced_sums, C, inv_std, weight, eps, mode, dweight, dbias, coef,
                                           reinterpret_cast<hipStream_t>(stream)), "bn_backward_finalize");
}

int eamm_bn_backward_apply(const float* x, const float* dy, const float* mean, const float* coef, int N, int C, int HW, float* dx,
                           void* stream) {
    if (!x || !dy || !mean || !coef || !dx) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (N < 1 || C < 1 || HW < 1) return bn_fail(EAMM_ERR_ARG, "empty tensor [%d,%d,%d]", N, C, HW);
    return bn_check(bn_bwd_apply_launch(x, dy, mean, coef, N, C, HW, dx, reinterpret_cast<hipStream_t>(stream)), "bn_backward_apply");
}

// ---- the same module on NHWC activations, fused with the block's ReLU and DownBlock2d's 2x2 average (batchnorm_nhwc.hip) ----
size_t eamm_bn_nhwc_workspace_floats(long long M, int C) {
    if (M < 1 || C < 4 || (C & 3)) return 0;
    return bn_nhwc_workspace_floats(M, C);
}

int eamm_bn_nhwc_local_sums(const float* x, long long M, int C, float* sums, float* workspace, void* stream) {
    if (!x || !sums || !workspace) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (M < 1 || C < 4 || (C & 3) || C > 1024 || M >= (1ll << 36)) return bn_fail(EAMM_ERR_ARG, "bad shape [%lld,%d] (C a multiple of 4, at most 1024)", M, C);
    return bn_check(bn_nhwc_sums_launch(x, M, C, sums, workspace, reinterpret_cast<hipStream_t>(stream)), "bn_nhwc_local_sums");
}

int eamm_bn_nhwc_local_stats(const float* x, long long M, int C, float eps, float momentum, int mode, const float* weight,
                             float* running_mean, float* running_var, float* sums, float* mean, float* scale, float* inv_std,
                             float* workspace, void* stream) {
    if (!x || !sums || !workspace || !running_mean || !running_var || !mean || !scale) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (M < 1 || C < 4 || (C & 3) || C > 1024 || M >= (1ll << 36)) return bn_fail(EAMM_ERR_ARG, "bad shape [%lld,%d] (C a multiple of 4, at most 1024)", M, C);
    if (mode != 0 && mode != 1) return bn_fail(EAMM_ERR_ARG, "mode must be 0 (replicas' formula) or 1 (single replica)");
    BnFuse f;
    f.kind = 1;
    f.mode = mode;
    f.eps = eps;
    f.momentum = momentum;
    f.weight = weight;
    f.running_mean = running_mean;
    f.running_var = running_var;
    f.mean = mean;
    f.scale = scale;
    f.inv_std_out = inv_std;
    return bn_check(bn_nhwc_sums_launch(x, M, C, sums, workspace, reinterpret_cast<hipStream_t>(stream), &f), "bn_nhwc_local_stats");
}

int eamm_bn_nhwc_backward_local(const float* x, const float* dy, const float* mean, const float* scale, const float* bias, int B, int H,
                                int W, int C, int relu, int pool, const float* inv_std, const float* weight, float eps, int mode,
                                float* sums, float* grad_weight, float* grad_bias, float* coef, float* workspace, void* stream) {
    if (!x || !dy || !mean || !scale || !sums || !workspace || !inv_std || !coef) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || C > 1024 || (pool && ((H | W) & 1)))
        return bn_fail(EAMM_ERR_ARG, "bad shape [%d,%d,%d,%d]", B, H, W, C);
    if (mode != 0 && mode != 1) return bn_fail(EAMM_ERR_ARG, "mode must be 0 (replicas' formula) or 1 (single replica)");
    BnFuse f;
    f.kind = 2;
    f.mode = mode;
    f.eps = eps;
    f.weight = weight;
    f.inv_std = inv_std;
    f.dweight = grad_weight;
    f.dbias = grad_bias;
    f.coef = coef;
    return bn_check(bn_nhwc_bwd_sums_launch(x, dy, mean, scale, bias, B, H, W, C, relu, pool, sums, workspace,
                                            reinterpret_cast<hipStream_t>(stream), &f), "bn_nhwc_backward_local");
}

int eamm_bn_nhwc_apply(const float* x, const float* mean, const float* scale, const float* bias, int B, int H, int W, int C, int relu,
                       int pool, float* y, void* stream) {
    if (!x || !mean || !scale || !y) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || (pool && ((H | W) & 1))) return bn_fail(EAMM_ERR_ARG, "bad shape [%d,%d,%d,%d]", B, H, W, C);
    return bn_check(bn_nhwc_apply_launch(x, mean, scale, bias, B, H, W, C, relu, pool, y, reinterpret_cast<hipStream_t>(stream)), "bn_nhwc_apply");
}

int eamm_bn_nhwc_backward_sums(const float* x, const float* dy, const float* mean, const float* scale, const float* bias, int B, int H,
                               int W, int C, int relu, int pool, float* sums, float* workspace, void* stream) {
    if (!x || !dy || !mean || !scale || !sums || !workspace) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || C > 1024 || (pool && ((H | W) & 1)))
        return bn_fail(EAMM_ERR_ARG, "bad shape [%d,%d,%d,%d]", B, H, W, C);
    return bn_check(bn_nhwc_bwd_sums_launch(x, dy, mean, scale, bias, B, H, W, C, relu, pool, sums, workspace,
                                            reinterpret_cast<hipStream_t>(stream)), "bn_nhwc_backward_sums");
}

int eamm_bn_nhwc_backward_apply(const float* x, const float* dy, const float* mean, const float* scale, const float* bias,
                                const float* coef, int B, int H, int W, int C, int relu, int pool, float* dx, void* stream) {
    if (!x || !dy || !mean || !scale || !coef || !dx) return bn_fail(EAMM_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || (pool && ((H | W) & 1))) return bn_fail(EAMM_ERR_ARG, "bad shape [%d,%d,%d,%d]", B, H, W, C);
    return bn_check(bn_nhwc_bwd_apply_launch(x, dy, mean, scale, bias, coef, B, H, W, C, relu, pool, dx,
                                             reinterpret_cast<hipStream_t>(stream)), "bn_nhwc_backward_apply");
}

}  // extern "C"
