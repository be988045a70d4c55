// C ABI of the audio-to-feature deconvolution tail (reference modules/util.py:559-576
// `AT_net2.decon`, evaluated once per frame at util.py:604-607 on the LSTM output):
//
//   [B,256,1,1] -ConvT 6x6 s2 p1, BN, ReLU-> [B,256,4,4] -ConvT 4x4 s2 p1, BN, ReLU-> 8x8 -> 16x16 -> 32x32
//               -ConvT 4x4 s2 p1-> [B,35,64,64]           (the feature maps KPDetector_a consumes, demo.py:219)
//
// MI355X form, on the path's fp32-MFMA convolution kernel (conv_mfma.hip):
//  * layer 0 sees a 1x1 map, so every output pixel (Y,X) has exactly one tap: a 1x1 convolution with
//    N = (Y, X, co) -- the NHWC [B,4,4,C1] tensor is its [B,1,1,16*C1] output, no scatter;
//  * ConvTranspose2d(k4, s2, p1): output pixel (2y+py, 2x+px) reads the 2x2 input neighbourhood
//    (y-1+py+ty, x-1+px+tx) with tap (ky,kx) = (3-py-2ty, 3-px-2tx) -- the four-phase 2x2 form the UpBlock2d
//    layers already run in (PHASE mode), here with explicit phase filters; nothing is zero-stuffed;
//  * eval-mode BatchNorm2d folds into weights and bias (fp64 on the host), ReLU in the epilogue, the last layer
//    writes NCHW directly.
#include "api_common.h"

using namespace eamm;

struct eamm_deconv_ctx : eamm::CtxBase {
    eamm_deconv_config cfg{};
    int nl = 0;
    std::vector<ConvLayer> layers;
    ConvLayer last_dma;           // round 6: the last layer's filters packed for the LDS-DMA 512 x 64 tile (batched calls: the clip harness's
    bool has_last_dma = false;    // 64-frame front-end batches ran it at 0.29 of the fp32 matrix peak on the register-staged 128 x 64 tile)
    std::vector<float*> bufs;     // NHWC output of layer i (all but the last)
    float* partial = nullptr;
    size_t partial_elems = 0;
};

namespace {

int side_of(int layer) { return 4 << layer; }   // output side of layer i: 4, 8, 16, ...

// fold eval-mode BatchNorm (key prefix `norm`, may be empty) into per-channel scale / bias
int fold_norm(eamm_deconv_ctx* c, const std::string& conv, const std::string& norm, int Cout, std::vector<double>* scale,
              std::vector<double>* bias) {
    const HostTensor* bt = find(c, conv + ".bias");
    if (!bt || (int)bt->numel() != Cout) return fail(c, EAMM_ERR_KEY, "%s.bias missing or mis-shaped", conv.c_str());
    scale->assign(Cout, 1.0);
    bias->resize(Cout);
    for (int o = 0; o < Cout; ++o) (*bias)[o] = bt->data[o];
    if (norm.empty()) return EAMM_OK;
    const HostTensor *g = find(c, norm + ".weight"), *be = find(c, norm + ".bias"), *mu = find(c, norm + ".running_mean"),
                     *var = find(c, norm + ".running_var");
    if (!g || !be || !mu || !var || (int)g->numel() != Cout || (int)be->numel() != Cout || (int)mu->numel() != Cout ||
        (int)var->numel() != Cout)
        return fail(c, EAMM_ERR_KEY, "BatchNorm entries of %s missing or mis-shaped", norm.c_str());
    for (int o = 0; o < Cout; ++o) {
        const double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
        (*scale)[o] = s;
        (*bias)[o] = ((*bias)[o] - (double)mu->data[o]) * s + (double)be->data[o];
    }
    return EAMM_OK;
}

int finish_layer(eamm_deconv_ctx* c, ConvLayer* L, int kh, int kw, bool phase, int Cin, int Cout,
                 const std::vector<float>& wf, const std::vector<float>& bf, int dma_cfg = 0) {
    L->kh = kh;
    L->kw = kw;
    L->phase = phase;
    L->C0 = Cin;
    L->C1 = 0;
    L->Cout = Cout;
    L->BM = 128;
    L->BN = conv_tile_n(Cout);
    L->dma_cfg = dma_cfg;
    if (dma_cfg > 0 && !conv_dma_tile(dma_cfg, &L->BM, &L->BN)) return fail(c, EAMM_ERR_ARG, "unknown dma tile %d", dma_cfg);
    L->ntiles = (Cout + L->BN - 1) / L->BN;
    const int taps = phase ? 4 : kh * kw;
    L->nchunks = taps * (Cin / CONV_BK);
    std::vector<float> packed(conv_packed_elems(taps, Cin, Cout, L->BN, phase ? 4 : 1));
    if (phase)
        conv_pack_phases_host(wf.data(), Cout, Cin, nullptr, Cin, L->BN, dma_cfg > 0, packed.data());
    else
        conv_pack_host(wf.data(), Cout, Cin, kh, kw, nullptr, Cin, L->BN, false, dma_cfg > 0, packed.data());
    std::vector<float> bias_pad((size_t)L->ntiles * L->BN, 0.f);
    std::copy(bf.begin(), bf.end(), bias_pad.begin());
    if (int rc = upload(c, &L->w, packed)) return rc;
    return upload(c, &L->bias, bias_pad);
}

}  // namespace

extern "C" {

const char* eamm_deconv_last_error(const eamm_deconv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int eamm_deconv_create(const eamm_deconv_config* cfg, int device, eamm_deconv_ctx** out) {
    if (!cfg || !out) return fail(nullptr, EAMM_ERR_ARG, "null argument");
    *out = nullptr;
    const eamm_deconv_config& g = *cfg;
    if (g.num_layers < 2 || g.num_layers > 8) return fail(nullptr, EAMM_ERR_ARG, "num_layers must be in [2,8]");
    for (int i = 0; i < g.num_layers; ++i)
        if (g.channels[i] < 32 || g.channels[i] % 32)
            return fail(nullptr, EAMM_ERR_ARG, "channels[%d] = %d: layer inputs must be positive multiples of 32", i, g.channels[i]);
    if (g.channels[g.num_layers] < 1) return fail(nullptr, EAMM_ERR_ARG, "output channels < 1");
    if (g.max_batch < 1) return fail(nullptr, EAMM_ERR_ARG, "max_batch < 1");
    {   // validate the device without leaving it selected in the caller's thread
        DeviceGuard probe(device);
        if (probe.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice(%d) failed", device);
    }
    eamm_deconv_ctx* c = new eamm_deconv_ctx();
    c->cfg = g;
    c->device = device;
    c->nl = g.num_layers;
    read_tile_knobs(c);
    *out = c;
    return EAMM_OK;
}

void eamm_deconv_destroy(eamm_deconv_ctx* c) {
    if (!c) return;
    free_owned(c);
    delete c;
}

int eamm_deconv_load_tensor(eamm_deconv_ctx* c, const char* key, const float* host, const int64_t* shape, int ndim) {
    return store_tensor(c, key, host, shape, ndim);
}

int eamm_deconv_finalize_weights(eamm_deconv_ctx* c) {
    if (!c) return EAMM_ERR_ARG;
    if (c->finalized) return fail(c, EAMM_ERR_STATE, "weights already finalised");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const eamm_deconv_config& g = c->cfg;
    {   // nn.Sequential numbering: ConvTranspose2d at 3i, BatchNorm2d at 3i+1 (ReLU at 3i+2 has no entries)
        std::vector<std::string> want;
        for (int i = 0; i < c->nl; ++i) {
            const std::string cv = std::to_string(3 * i);
            want.push_back(cv + ".weight");
            want.push_back(cv + ".bias");
            if (i + 1 < c->nl)
                for (const char* sfx : {".weight", ".bias", ".running_mean", ".running_var"})
                    want.push_back(std::to_string(3 * i + 1) + sfx);
        }
        if (int rc = check_keys(c, want)) return rc;
    }
    c->layers.resize(c->nl);
    int rc;
    for (int i = 0; i < c->nl; ++i) {
        const int Cin = g.channels[i], Cout = g.channels[i + 1], k = i == 0 ? 6 : 4;
        const std::string cv = std::to_string(3 * i), nm = i + 1 < c->nl ? std::to_string(3 * i + 1) : std::string();
        const HostTensor* wt = find(c, cv + ".weight");   // ConvTranspose2d layout [Cin][Cout][k][k]
        if (!wt || wt->shape.size() != 4 || wt->shape[0] != Cin || wt->shape[1] != Cout || wt->shape[2] != k || wt->shape[3] != k)
            return fail(c, EAMM_ERR_KEY, "%s.weight missing or mis-shaped (expected [%d,%d,%d,%d])", cv.c_str(), Cin, Cout, k, k);
        std::vector<double> sc, bs;
        if ((rc = fold_norm(c, cv, nm, Cout, &sc, &bs))) return rc;
        auto W = [&](int ci, int o, int ky, int kx) { return (double)wt->data[(((size_t)ci * Cout + o) * k + ky) * k + kx]; };
        if (i == 0) {
            // 1x1 input, stride 2, padding 1: out[Y][X] = in * w[Y+1][X+1] for Y, X in 0..3 -> N = (Y*4 + X)*Cout + o
            const int N = 16 * Cout;
            std::vector<float> wf((size_t)N * Cin), bf(N);
            for (int Y = 0; Y < 4; ++Y)
                for (int X = 0; X < 4; ++X)
                    for (int o = 0; o < Cout; ++o) {
                        const int n = (Y * 4 + X) * Cout + o;
                        bf[n] = (float)bs[o];
                        for (int ci = 0; ci < Cin; ++ci) wf[(size_t)n * Cin + ci] = (float)(W(ci, o, Y + 1, X + 1) * sc[o]);
                    }
            if ((rc = finish_layer(c, &c->layers[i], 1, 1, false, Cin, N, wf, bf))) return rc;
        } else {
            std::vector<float> wf((size_t)4 * Cout * Cin * 4), bf(Cout);
            for (int o = 0; o < Cout; ++o) bf[o] = (float)bs[o];
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                for (int o = 0; o < Cout; ++o)
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ty = 0; ty < 2; ++ty)
                            for (int tx = 0; tx < 2; ++tx)
                                wf[(((size_t)ph * Cout + o) * Cin + ci) * 4 + ty * 2 + tx] =
                                    (float)(W(ci, o, 3 - py - 2 * ty, 3 - px - 2 * tx) * sc[o]);
            }
            if ((rc = finish_layer(c, &c->layers[i], 2, 2, true, Cin, Cout, wf, bf))) return rc;
            if (i + 1 == c->nl && Cout + 1 <= 64 && c->dma_min_m >= 0) {   // (+ 1: the split hand-over's zero column)
                if ((rc = finish_layer(c, &c->last_dma, 2, 2, true, Cin, Cout, wf, bf, 3))) return rc;
                c->has_last_dma = true;
            }
        }
    }
    const size_t F = g.max_batch;
    c->bufs.resize(c->nl - 1);
    for (int i = 0; i + 1 < c->nl; ++i)
        if ((rc = dev_alloc(c, &c->bufs[i], F * side_of(i) * side_of(i) * g.channels[i + 1]))) return rc;
    size_t need = 0;
    for (size_t f = 1; f <= F; ++f)
        for (int i = 0; i < c->nl; ++i) {
            const size_t M = i == 0 ? f : f * side_of(i - 1) * side_of(i - 1);
            need = std::max(need, conv_plan(c->layers[i], (int)M).partial_elems);
            if (i + 1 == c->nl && c->has_last_dma) need = std::max(need, conv_plan(c->last_dma, (int)M).partial_elems);
        }
    c->partial_elems = need;
    if ((rc = dev_alloc(c, &c->partial, need))) return rc;
    c->sd.clear();
    HIP_TRY(c, hipDeviceSynchronize());
    c->finalized = true;
    return EAMM_OK;
}

// The private hand-over to KPDetector_a (round 6): the last layer's C_last = 32 m + 3 channels leave as NHWC [B,S,S,32 m] (`wide`, the
// operand the heads' 7x7 MFMA convolution reads) + one float4 per pixel (`thin`: the last three channels and a zero, what the thin
// 7x7 kernel reads) -- no NCHW tensor in between, no layout kernels on the other side.  0 when the channel count has no such split.
int eamm_deconv_split_channels(const eamm_deconv_ctx* c) {
    if (!c) return 0;
    const int wide = c->cfg.channels[c->nl] - 3;
    return (wide >= 32 && wide % 32 == 0) ? wide : 0;
}

static int deconv_run(eamm_deconv_ctx* c, const float* x, int B, float* out, float* thin, void* stream_);

int eamm_deconv_forward_split(eamm_deconv_ctx* c, const float* x, int B, float* wide, float* thin, void* stream_) {
    if (!c || !x || !wide || !thin) return fail(c, EAMM_ERR_ARG, "null argument");
    if (eamm_deconv_split_channels(c) == 0)
        return fail(c, EAMM_ERR_ARG, "the last layer's %d channels are not 32 m + 3: no split hand-over", c->cfg.channels[c->nl]);
    return deconv_run(c, x, B, wide, thin, stream_);
}

int eamm_deconv_forward(eamm_deconv_ctx* c, const float* x, int B, float* out, void* stream_) {
    if (!c || !x || !out) return fail(c, EAMM_ERR_ARG, "null argument");
    return deconv_run(c, x, B, out, nullptr, stream_);
}

static int deconv_run(eamm_deconv_ctx* c, const float* x, int B, float* out, float* thin, void* stream_) {
    if (!c->finalized) return fail(c, EAMM_ERR_STATE, "call eamm_deconv_finalize_weights first");
    if (B < 1 || B > c->cfg.max_batch) return fail(c, EAMM_ERR_ARG, "batch %d outside [1,%d]", B, c->cfg.max_batch);
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    for (int i = 0; i < c->nl; ++i) {
        const bool last = i + 1 == c->nl;
        ConvIO io{};
        io.in0 = i == 0 ? x : c->bufs[i - 1];
        io.B = B;
        io.Hin = io.Win = i == 0 ? 1 : side_of(i - 1);
        io.act = last ? ACT_NONE : ACT_RELU;     // util.py:559-574: BatchNorm2d + ReLU after every layer but the last
        io.nchw = (last && !thin) ? 1 : 0;
        io.out = last ? out : c->bufs[i];
        io.partial = c->partial;
        io.partial_cap = c->partial_elems;
        const bool dma = last && c->has_last_dma && (long long)B * io.Hin * io.Win >= c->dma_min_m;
        if (last && thin) {   // split hand-over: column C_last (zero weights, zero bias) fills the float4's fourth slot
            ConvLayer L = dma ? c->last_dma : c->layers[i];
            io.split_n = L.Cout - 3;
            io.out2 = thin;
            L.Cout += 1;
            HIP_TRY(c, conv_launch(L, io, s));
            continue;
        }
        HIP_TRY(c, conv_launch(dma ? c->last_dma : c->layers[i], io, s));
    }
    return EAMM_OK;
}

}  // extern "C"
