// Shared internals of the C-ABI translation units (eamm_api.hip, eamm_kp_api.hip): handle base class, error
// reporting, device allocation, and the state_dict -> BatchNorm-folded, MFMA-packed layer builders.
#pragma once
#include "../../include/eamm_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace eamm {
namespace {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const {
        size_t n = 1;
        for (auto s : shape) n *= (size_t)s;
        return n;
    }
};

thread_local std::string g_create_error;

// A convolution layer packed for the register-staged 128 x BN kernel and, where it pays, also for an LDS-DMA
// big-tile kernel; conv launches pick by problem size (see pick()).
struct LayerSet {
    ConvLayer base, dma, big;
    ConvLayer skinny32, skinny64;   // S.dma's weights (N tile 128) behind the 32- / 64-row LDS-DMA tiles: M <= 32 / 64
    bool has_skinny = false;
    PatchLayer patch;           // UpBlock2d layers only: spatial-patch kernel (conv_mfma_patch.hip)
    bool has_dma = false, has_big = false, has_patch = false;
};



// What every handle type carries: device, last error, the raw state_dict until finalisation, owned device memory
// and the convolution-tile tuning knobs (environment, read at create).
struct CtxBase {
    int device = 0;
    std::string err;
    std::map<std::string, HostTensor> sd;
    bool finalized = false;
    std::vector<void*> owned;   // every device allocation, freed by free_owned()
    int dma_min_m = 4608;       // smallest per-phase M for which an LDS-DMA tile is preferred (1024 until round 3: the 512x64 tile left the
                                // last hourglass decoder level at 8-32 workgroups for calls of 1-4 frames; one frame 1044 -> 1076 frames/s)
    int big_min_m = 49152;      // ... and for which Cout % 256 == 0 layers use the 256x256 tile (>= 192 M tiles)
    int dma_cfg_n256 = 2, dma_cfg_n128 = 2, dma_cfg_n64 = 3;
    int patch_poly = 1;         // up blocks on the patch kernel in the polyphase minimal-filtering form (EAMM_PATCH_POLY; 0: collapsed-phase form)
    int skinny_max_m = 16384;   // largest per-phase pixel count served by the 32- / 64-row tiles (EAMM_SKINNY_MAX_M; 0 = off):
                                // 64x128 tiles need no split-K where 256x128 ones do (measured 256x256: 1 frame 753 -> 785, 4 frames 1969 -> 2067,
                                // 8 frames 2756 -> 2782 frames/s; 16 frames unchanged)
    int patch_split_max = 4;    // largest split of the channel reduction of the polyphase kernel over workgroups (EAMM_PATCH_SPLIT_MAX; 1 = off)
    int patch_min_blocks = 128; // fewest workgroups for which UpBlock2d layers use the spatial-patch kernel (< 0: never)
};

// Every C entry point that touches the device runs under one of these: the handle's device becomes current for
// the call and the caller's current device is restored on return (PyTorch reads the current device from the HIP
// runtime, so a library that leaves another device selected would silently redirect the caller's later work).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            status = hipSetDevice(device);
            switched = status == hipSuccess && prev >= 0;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(CtxBase* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail((c), EAMM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

template <typename T>
int dev_alloc(CtxBase* c, T** p, size_t elems) {
    void* q = nullptr;
    HIP_TRY(c, hipMalloc(&q, std::max<size_t>(elems, 1) * sizeof(T)));
    c->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

int upload(CtxBase* c, float** dst, const std::vector<float>& src) {
    int rc = dev_alloc(c, dst, src.size());
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

const HostTensor* find(const CtxBase* c, const std::string& key) {
    auto it = c->sd.find(key);
    return it == c->sd.end() ? nullptr : &it->second;
}

// conv (+ optional BatchNorm folded behind it) -> packed device layer.
// y = ((conv(x) + b) - mean) * gamma / sqrt(var + eps) + beta   (sync_batchnorm/batchnorm.py:48-53, eps 1e-5)
struct FoldSpec {
    std::string conv;            // key prefix holding .weight / .bias
    std::string norm;            // key prefix holding BatchNorm stats, or empty
};

enum LayerMode { MODE_PLAIN = 0, MODE_PHASE = 1, MODE_ROWSPLIT = 2 };

// MODE_PLAIN: ks x ks convolution.  MODE_PHASE: ks must be 3; the layer consumes the low-resolution input of an
// UpBlock2d and evaluates "nearest x2 + 3x3" as four 2x2 phase filters.  MODE_ROWSPLIT: ks x 1 convolution with
// N = (dx, co) -- the horizontal taps become output channels, gathered afterwards by final_shift_sum; the bias is
// returned in *row_bias instead of being applied by the convolution.
int build_layer(CtxBase* c, const std::vector<FoldSpec>& parts, int ks, int C0_real, int C0_packed, int C1_real,
                int C1_packed, ConvLayer* L, LayerMode mode = MODE_PLAIN, std::vector<float>* row_bias = nullptr,
                int dma_cfg = 0, float** w_swizzled = nullptr, int w_swizzled_bn = 0) {
    // `parts` are stacked along Cout (the flow head stacks mask + occlusion into one convolution)
    const int Cin = C0_real + C1_real;
    const int T = ks * ks;
    int Cout = 0;
    for (auto& ps : parts) {
        const HostTensor* wt = find(c, ps.conv + ".weight");
        if (!wt || wt->shape.size() != 4 || wt->shape[1] != Cin || wt->shape[2] != ks || wt->shape[3] != ks)
            return fail(c, EAMM_ERR_KEY, "state_dict entry %s.weight missing or mis-shaped (expected [*,%d,%d,%d])",
                        ps.conv.c_str(), Cin, ks, ks);
        Cout += (int)wt->shape[0];
    }
    std::vector<float> wf((size_t)Cout * Cin * T), bf(Cout);
    int o0 = 0;
    for (auto& ps : parts) {
        const HostTensor* wt = find(c, ps.conv + ".weight");
        const HostTensor* bt = find(c, ps.conv + ".bias");
        const int co = (int)wt->shape[0];
        if (!bt || (int)bt->numel() != co) return fail(c, EAMM_ERR_KEY, "%s.bias missing or mis-shaped", ps.conv.c_str());
        const HostTensor *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
        if (!ps.norm.empty()) {
            g = find(c, ps.norm + ".weight");
            be = find(c, ps.norm + ".bias");
            mu = find(c, ps.norm + ".running_mean");
            var = find(c, ps.norm + ".running_var");
            if (!g || !be || !mu || !var || (int)g->numel() != co || (int)be->numel() != co ||
                (int)mu->numel() != co || (int)var->numel() != co)
                return fail(c, EAMM_ERR_KEY, "BatchNorm entries of %s missing or mis-shaped", ps.norm.c_str());
        }
        for (int o = 0; o < co; ++o) {
            double s = 1.0, shift = 0.0, b = bt->data[o];
            if (g) {
                s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
                b = (b - (double)mu->data[o]) * s + (double)be->data[o];
            }
            (void)shift;
            bf[o0 + o] = (float)b;
            const float* src = wt->data.data() + (size_t)o * Cin * T;
            float* dst = wf.data() + (size_t)(o0 + o) * Cin * T;
            for (size_t i = 0; i < (size_t)Cin * T; ++i) dst[i] = (float)((double)src[i] * s);
        }
        o0 += co;
    }
    const int cin_packed = C0_packed + C1_packed;
    std::vector<int> map(cin_packed, -1);
    for (int i = 0; i < C0_real; ++i) map[i] = i;
    for (int i = 0; i < C1_real; ++i) map[C0_packed + i] = C0_real + i;
    int kh = ks, kw = ks;
    if (mode == MODE_ROWSPLIT) {
        // w'[dx*Cout+co][c][dy] = w[co][c][dy][dx]
        std::vector<float> wr((size_t)ks * Cout * Cin * ks);
        for (int dx = 0; dx < ks; ++dx)
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int dy = 0; dy < ks; ++dy)
                        wr[((size_t)(dx * Cout + co) * Cin + ci) * ks + dy] = wf[((size_t)co * Cin + ci) * T + dy * ks + dx];
        if (row_bias) *row_bias = bf;
        wf.swap(wr);
        Cout = ks * Cout;
        bf.assign(Cout, 0.f);
        kw = 1;
    }
    if (mode == MODE_PHASE && ks != 3) return fail(c, EAMM_ERR_ARG, "phase mode needs a 3x3 convolution");
    L->kh = kh;
    L->kw = kw;
    L->phase = mode == MODE_PHASE;
    L->C0 = C0_packed;
    L->C1 = C1_packed;
    L->Cout = Cout;
    L->BM = 128;
    L->BN = conv_tile_n(Cout);
    L->dma_cfg = dma_cfg;
    if (dma_cfg > 0 && !conv_dma_tile(dma_cfg, &L->BM, &L->BN)) return fail(c, EAMM_ERR_ARG, "unknown dma tile %d", dma_cfg);
    L->ntiles = (Cout + L->BN - 1) / L->BN;
    const int taps = L->phase ? 4 : kh * kw;
    L->nchunks = taps * (cin_packed / CONV_BK);
    std::vector<float> packed(conv_packed_elems(taps, cin_packed, Cout, L->BN, L->phase ? 4 : 1));
    conv_pack_host(wf.data(), Cout, Cin, kh, kw, map.data(), cin_packed, L->BN, L->phase, dma_cfg > 0, packed.data());
    std::vector<float> bias_pad((size_t)L->ntiles * L->BN, 0.f);
    std::copy(bf.begin(), bf.end(), bias_pad.begin());
    int rc = upload(c, &L->w, packed);
    if (rc) return rc;
    if (w_swizzled) {   // second image of the same weights in the LDS-DMA (XOR-swizzled) layout
        const int sbn = w_swizzled_bn ? w_swizzled_bn : L->BN;   // N tile of the swizzled image (the consumer kernel's)
        std::vector<float> ps(conv_packed_elems(taps, cin_packed, Cout, sbn, L->phase ? 4 : 1));
        conv_pack_host(wf.data(), Cout, Cin, kh, kw, map.data(), cin_packed, sbn, L->phase, true, ps.data());
        if ((rc = upload(c, w_swizzled, ps))) return rc;
    }
    return upload(c, &L->bias, bias_pad);
}

// UpBlock2d conv + BatchNorm -> spatial-patch packing (same fold as build_layer, phase filters pre-summed on the host)
int build_patch(CtxBase* c, const std::string& conv, const std::string& norm, int C0_real, int C0_packed, int C1_real,
                int C1_packed, PatchLayer* P) {
    const int Cin = C0_real + C1_real;
    const HostTensor *wt = find(c, conv + ".weight"), *bt = find(c, conv + ".bias");
    const HostTensor *g = find(c, norm + ".weight"), *be = find(c, norm + ".bias"), *mu = find(c, norm + ".running_mean"),
                     *var = find(c, norm + ".running_var");
    if (!wt || !bt || !g || !be || !mu || !var || wt->shape.size() != 4 || wt->shape[1] != Cin || wt->shape[2] != 3)
        return fail(c, EAMM_ERR_KEY, "%s mis-shaped for the patch kernel", conv.c_str());
    const int Cout = (int)wt->shape[0];
    std::vector<float> wf(wt->data), bf(Cout);
    for (int o = 0; o < Cout; ++o) {
        const double sc = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
        bf[o] = (float)(((double)bt->data[o] - (double)mu->data[o]) * sc + (double)be->data[o]);
        for (size_t i = 0; i < (size_t)Cin * 9; ++i) wf[(size_t)o * Cin * 9 + i] = (float)((double)wt->data[(size_t)o * Cin * 9 + i] * sc);
    }
    const int cin_packed = C0_packed + C1_packed;
    std::vector<int> map(cin_packed, -1);
    for (int i = 0; i < C0_real; ++i) map[i] = i;
    for (int i = 0; i < C1_real; ++i) map[C0_packed + i] = C0_real + i;
    P->C0 = C0_packed;
    P->C1 = C1_packed;
    P->Cout = Cout;
    std::vector<float> bias_pad((size_t)((Cout + 63) / 64) * 64, 0.f);
    std::copy(bf.begin(), bf.end(), bias_pad.begin());
    int rc = 0;
    if (c->patch_poly) {
        std::vector<float> pp(patch_poly_packed_elems(cin_packed, Cout));
        patch_poly_pack_host(wf.data(), Cout, Cin, map.data(), cin_packed, pp.data());
        if ((rc = upload(c, &P->w_poly, pp))) return rc;
    } else {
        std::vector<float> packed(patch_packed_elems(cin_packed, Cout));
        patch_pack_host(wf.data(), Cout, Cin, map.data(), cin_packed, packed.data());
        if ((rc = upload(c, &P->w, packed))) return rc;
    }
    return upload(c, &P->bias, bias_pad);
}

int build_set(CtxBase* c, const std::vector<FoldSpec>& parts, int C0_real, int C0_packed, int C1_real, int C1_packed,
              LayerSet* S, LayerMode mode) {
    int rc = build_layer(c, parts, 3, C0_real, C0_packed, C1_real, C1_packed, &S->base, mode);
    if (rc) return rc;
    const int Cout = S->base.Cout;
    int cfg = 0;
    if (Cout % 256 == 0) cfg = c->dma_cfg_n256;
    else if (Cout % 128 == 0) cfg = c->dma_cfg_n128;
    else if (Cout == 64) cfg = c->dma_cfg_n64;
    if (cfg > 0 && c->dma_min_m >= 0) {
        rc = build_layer(c, parts, 3, C0_real, C0_packed, C1_real, C1_packed, &S->dma, mode, nullptr, cfg);
        if (rc) return rc;
        S->has_dma = true;
        if (S->dma.BN == 128 && c->skinny_max_m > 0) {   // same packed weights, small-M tiles (conv_dma_tile ids 4, 5)
            S->skinny32 = S->dma;
            S->skinny64 = S->dma;
            S->skinny32.dma_cfg = 4;
            S->skinny64.dma_cfg = 5;
            conv_dma_tile(4, &S->skinny32.BM, &S->skinny32.BN);
            conv_dma_tile(5, &S->skinny64.BM, &S->skinny64.BN);
            S->has_skinny = true;
        }
        if (Cout % 256 == 0 && cfg != 1 && c->big_min_m >= 0) {
            rc = build_layer(c, parts, 3, C0_real, C0_packed, C1_real, C1_packed, &S->big, mode, nullptr, 1);
            if (rc) return rc;
            S->has_big = true;
        }
    }
    if (mode == MODE_PHASE && parts.size() == 1 && !parts[0].norm.empty() && (Cout & 3) == 0 && c->patch_min_blocks >= 0) {
        rc = build_patch(c, parts[0].conv, parts[0].norm, C0_real, C0_packed, C1_real, C1_packed, &S->patch);
        if (rc) return rc;
        S->has_patch = true;
    }
    return 0;
}



// Tile choice per launch (measured on MI355X, profiles/r01_convbench_*): M is the per-phase pixel count.
// conv (+ folded BatchNorm) -> Winograd-domain weights U = G g G^T
int build_wino(CtxBase* c, const std::string& conv, const std::string& norm, int C, WinoLayer* L, int tile = 2) {
    const HostTensor *wt = find(c, conv + ".weight"), *bt = find(c, conv + ".bias");
    if (!wt || !bt || wt->shape.size() != 4 || wt->shape[0] != C || wt->shape[1] != C || wt->shape[2] != 3)
        return fail(c, EAMM_ERR_KEY, "%s mis-shaped for the Winograd path", conv.c_str());
    std::vector<float> wf(wt->data), bf(bt->data);
    if (!norm.empty()) {
        const HostTensor *g = find(c, norm + ".weight"), *be = find(c, norm + ".bias"),
                         *mu = find(c, norm + ".running_mean"), *var = find(c, norm + ".running_var");
        if (!g || !be || !mu || !var) return fail(c, EAMM_ERR_KEY, "BatchNorm entries of %s missing", norm.c_str());
        for (int o = 0; o < C; ++o) {
            const double sc = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
            bf[o] = (float)(((double)bt->data[o] - (double)mu->data[o]) * sc + (double)be->data[o]);
            for (size_t i = 0; i < (size_t)C * 9; ++i) wf[(size_t)o * C * 9 + i] = (float)((double)wt->data[(size_t)o * C * 9 + i] * sc);
        }
    }
    L->Cin = C;
    L->Cout = C;
    L->tile = tile;
    L->BN = tile == 4 ? 64 : 128;
    L->ntiles = (C + L->BN - 1) / L->BN;
    std::vector<float> packed(tile == 4 ? wino4_packed_elems(C, C, L->BN) : wino_packed_elems(C, C, L->BN));
    if (tile == 4)
        wino4_pack_host(wf.data(), C, C, L->BN, packed.data());
    else
        wino_pack_host(wf.data(), C, C, L->BN, packed.data());
    std::vector<float> bias_pad((size_t)L->ntiles * L->BN, 0.f);
    std::copy(bf.begin(), bf.end(), bias_pad.begin());
    int rc = upload(c, &L->u, packed);
    if (rc) return rc;
    return upload(c, &L->bias, bias_pad);
}

// F(4x4,3x3) packing of a convolution whose input and output widths differ (hourglass DownBlock2d: conv -> BatchNorm
// folded; reference modules/util.py:903-921); input channels Cin_real sit first in a Cin_packed-wide activation
int build_wino4_rect(CtxBase* c, const std::string& conv, const std::string& norm, int Cin_real, int Cin_packed, WinoLayer* L) {
    const HostTensor *wt = find(c, conv + ".weight"), *bt = find(c, conv + ".bias");
    if (!wt || !bt || wt->shape.size() != 4 || wt->shape[1] != Cin_real || wt->shape[2] != 3 || wt->shape[3] != 3 ||
        Cin_packed < Cin_real || Cin_packed % 64)
        return fail(c, EAMM_ERR_KEY, "%s mis-shaped for the Winograd path", conv.c_str());
    const int Cout = (int)wt->shape[0];
    std::vector<float> wf((size_t)Cout * Cin_packed * 9, 0.f), bf(bt->data);
    const HostTensor *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
    if (!norm.empty()) {
        g = find(c, norm + ".weight"); be = find(c, norm + ".bias");
        mu = find(c, norm + ".running_mean"); var = find(c, norm + ".running_var");
        if (!g || !be || !mu || !var) return fail(c, EAMM_ERR_KEY, "BatchNorm entries of %s missing", norm.c_str());
    }
    for (int o = 0; o < Cout; ++o) {
        const double sc = g ? (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5) : 1.0;
        if (g) bf[o] = (float)(((double)bt->data[o] - (double)mu->data[o]) * sc + (double)be->data[o]);
        for (int ci = 0; ci < Cin_real; ++ci)
            for (int k = 0; k < 9; ++k)
                wf[((size_t)o * Cin_packed + ci) * 9 + k] = (float)((double)wt->data[((size_t)o * Cin_real + ci) * 9 + k] * sc);
    }
    L->Cin = Cin_packed;
    L->Cout = Cout;
    L->tile = 4;
    L->BN = 64;
    L->ntiles = (Cout + L->BN - 1) / L->BN;
    std::vector<float> packed(wino4_packed_elems(Cout, Cin_packed, L->BN));
    wino4_pack_host(wf.data(), Cout, Cin_packed, L->BN, packed.data());
    std::vector<float> bias_pad((size_t)L->ntiles * L->BN, 0.f);
    std::copy(bf.begin(), bf.end(), bias_pad.begin());
    int rc = upload(c, &L->u, packed);
    if (rc) return rc;
    return upload(c, &L->bias, bias_pad);
}

const ConvLayer& pick(const CtxBase* c, const LayerSet& S, size_t M) {
    // skinny GEMMs (deep hourglass levels at small batches): the weight stream is the cost, so tiles without padding rows
    if (S.has_skinny && M <= (size_t)c->skinny_max_m) return M <= 32 ? S.skinny32 : S.skinny64;
    if (S.has_big && M >= (size_t)c->big_min_m) return S.big;
    return (S.has_dma && M >= (size_t)c->dma_min_m) ? S.dma : S.base;
}

// UpBlock2d launch: the spatial-patch kernel when its 16x16-pixel tiles fill the chip, else the im2col-style kernels
int launch_up(CtxBase* c, const LayerSet& S, const ConvIO& io, hipStream_t s) {
    if (S.has_patch) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
        int splits = 1;
        if (S.patch.w_poly) {
            splits = patch_poly_splits(S.patch, io.B, io.Hin, io.Win, c->patch_split_max, cus);
            if ((size_t)splits * io.B * 4 * io.Hin * io.Win * S.patch.Cout > io.partial_cap) splits = 1;
        }
        const int blocks = ((io.Hin + 15) / 16) * ((io.Win + 15) / 16) * io.B * ((S.patch.Cout + 31) / 32) * splits;   // of the polyphase kernel
        if (blocks >= c->patch_min_blocks && io.Hin >= 16 && io.Win >= 16 && io.act == ACT_RELU && !io.resid && !io.out2 && !io.pool && !io.nchw) {
            if (S.patch.w_poly)
                HIP_TRY(c, patch_poly_launch(S.patch, io.in0, io.in1, io.B, io.Hin, io.Win, io.act, io.out, s, splits, io.partial,
                                             io.partial_cap));
            else
                HIP_TRY(c, patch_phase_launch(S.patch, io.in0, io.in1, io.B, io.Hin, io.Win, io.act, io.out, s));
            return EAMM_OK;
        }
    }
    HIP_TRY(c, conv_launch(pick(c, S, (size_t)io.B * io.Hin * io.Win), io, s));
    return EAMM_OK;
}

int env_int(const char* name, int dflt) { return (int)knob_int(name, dflt); }   // recorded: eamm_knobs_json

int round_up(int v, int m) { return (v + m - 1) / m * m; }

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }


inline void read_tile_knobs(CtxBase* c) {
    // tuning knobs (defaults measured on MI355X, profiles/): EAMM_DMA_MIN_M < 0 disables the LDS-DMA kernels
    c->dma_min_m = env_int("EAMM_DMA_MIN_M", c->dma_min_m);
    c->big_min_m = env_int("EAMM_BIG_MIN_M", c->big_min_m);
    c->patch_poly = env_int("EAMM_PATCH_POLY", c->patch_poly);
    c->dma_cfg_n256 = env_int("EAMM_DMA_CFG_N256", c->dma_cfg_n256);
    c->dma_cfg_n128 = env_int("EAMM_DMA_CFG_N128", c->dma_cfg_n128);
    c->dma_cfg_n64 = env_int("EAMM_DMA_CFG_N64", c->dma_cfg_n64);
    c->patch_min_blocks = env_int("EAMM_PATCH_MIN_BLOCKS", c->patch_min_blocks);
    c->patch_split_max = env_int("EAMM_PATCH_SPLIT_MAX", c->patch_split_max);
    c->skinny_max_m = env_int("EAMM_SKINNY_MAX_M", c->skinny_max_m);
}

inline void free_owned(CtxBase* c) {
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
}

inline int store_tensor(CtxBase* c, const char* key, const float* host, const int64_t* shape, int ndim) {
    if (!c || !key || !host || ndim < 0 || (ndim > 0 && !shape)) return fail(c, EAMM_ERR_ARG, "null argument");
    if (c->finalized) return fail(c, EAMM_ERR_STATE, "weights already finalised; create a new handle to reload");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(host, host + t.numel());
    c->sd[key] = std::move(t);
    return EAMM_OK;
}

// strict key-set check, like load_state_dict(strict=True) (reference demo.py:91)
inline int check_keys(CtxBase* c, const std::vector<std::string>& want) {
    std::string missing, unexpected;
    for (auto& k : want)
        if (!c->sd.count(k)) missing += (missing.empty() ? "" : ", ") + k;
    for (auto& kv : c->sd) {
        const std::string& k = kv.first;
        if (k.size() > 20 && k.compare(k.size() - 20, 20, ".num_batches_tracked") == 0) continue;
        if (std::find(want.begin(), want.end(), k) == want.end()) unexpected += (unexpected.empty() ? "" : ", ") + k;
    }
    if (!missing.empty() || !unexpected.empty())
        return fail(c, EAMM_ERR_KEY, "state_dict mismatch. Missing key(s): [%s]. Unexpected key(s): [%s].",
                    missing.substr(0, 400).c_str(), unexpected.substr(0, 400).c_str());
    return EAMM_OK;
}

inline void block_keys(const std::string& p, std::vector<std::string>* keys) {   // conv + BatchNorm block
    keys->push_back(p + ".conv.weight");
    keys->push_back(p + ".conv.bias");
    for (const char* s : {".norm.weight", ".norm.bias", ".norm.running_mean", ".norm.running_var"}) keys->push_back(p + s);
}

}  // namespace
}  // namespace eamm
