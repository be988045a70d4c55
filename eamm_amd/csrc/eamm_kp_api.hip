// C ABI of the key-point detectors (reference modules/keypoint_detector.py):
//   KPDetector   (:7-105)   image -> anti-alias down -> Hourglass -> 7x7 heads -> spatial softmax -> key points
//   KPDetector_a (:110-205) the same heads on a given feature map (the audio-driven 35 x 64 x 64 maps, once per
//                           frame in demo.py:219)
// Both reuse the path's kernels: the hourglass is the same fp32-MFMA convolution stack as the dense-motion
// network's (pool in the epilogue, collapsed up-conv, concat in the loader), the `kp` and `jacobian` 7x7
// convolutions run as ONE "same"-padded MFMA convolution (the reference's pad-0 output is its interior), and
// kp_head_kernel does the temperature softmax, soft-argmax and jacobian reduction per (image, key point).
#include "api_common.h"

using namespace eamm;

struct eamm_kp_ctx : eamm::CtxBase {
    eamm_kp_config cfg{};
    int H = 0, W = 0, h = 0, w = 0, K = 0, nb = 0, njm = 0;
    int Cin_pad = 32;             // RGB (or feature-map) channels padded to a multiple of 32
    int feat_c = 0;               // hourglass out_filters = block_expansion + in_features
    std::vector<int> enc_c, dec_c;
    std::vector<LayerSet> hg_enc, hg_dec;
    ConvLayer head;               // kp (K) + jacobian (4*njm) stacked along Cout
    ConvLayer head_dma;           // the same filters packed for the LDS-DMA tile (512 x 64 / 256 x 128), batched calls (round 6: the wide 7x7 head was
    bool has_head_dma = false;    // 73 % of KPDetector_a's time per frame at 0.38 of the fp32 matrix peak on the register-staged tile; 0.75 here)
    int head_dma_min_m = 16384;   // smallest B*h*w for which it is used (EAMM_KP_HEAD_DMA_MIN_M; < 0: never)
    int head_cs = 64;             // pixel stride of the logits = K + 4*njm rounded up to 64 (round 6: 128 / 192 for num_kp 13 .. 30)
    // KPDetector_a whose feature map is 32 m + 3 channels wide (the shipped 35 = block_expansion 32 + num_channels_a 3): the heads
    // run as a 7x7 MFMA convolution over the first 32 m channels (K = 49 x 32 m instead of 49 x the next multiple of 32: 45 % fewer
    // multiplies at 35) PLUS the thin-channel 7x7 kernel (conv7_thin.hip: K = (tap, channel) = 196) over the last three, added in
    // the wide convolution's epilogue (`resid`).  EAMM_KPA_THIN = 0: the plain padded convolution (rounds 1-4).
    int thin_wide = 0;            // > 0: channels of the wide part
    float *thin_w = nullptr, *thin_x = nullptr, *thin_y = nullptr, *thin_ws = nullptr;   // [64,3,7,7] filter, [B,h,w,4] input, [B,h,w,64] output, workspace
    float* aa_w = nullptr;
    float* img_stage = nullptr;   // num_channels 1 / 2: the image zero-extended to the three planes the anti-alias kernel reads
    float* head_ws = nullptr;     // kp_head_launch's workspace (pixel-sliced reductions of batched calls)
    float *x_in = nullptr, *logits = nullptr, *partial = nullptr;
    size_t partial_elems = 0;
    std::vector<float*> e_buf, u_buf;
};

namespace {

int run_head(eamm_kp_ctx* c, const float* in0, const float* in1, int B, const eamm_kp_outputs* o, hipStream_t s,
             const float* resid = nullptr) {
    ConvIO io{};
    io.in0 = in0;
    io.in1 = in1;
    io.resid = resid;
    io.B = B;
    io.Hin = c->h;
    io.Win = c->w;
    io.act = ACT_NONE;
    io.out = c->logits;
    io.partial = c->partial;
    io.partial_cap = c->partial_elems;
    ConvLayer head = (c->has_head_dma && c->head_dma_min_m >= 0 && (long long)B * c->h * c->w >= c->head_dma_min_m) ? c->head_dma : c->head;
    head.Cout = c->head_cs;  // logits are written with a head_cs-float pixel stride; channels >= K + 4*njm have zero weights
    HIP_TRY(c, conv_launch(head, io, s));
    HIP_TRY(c, kp_head_launch(c->logits, B, c->K, c->njm, c->h, c->w, c->head_cs, c->cfg.pad, c->cfg.temperature, o->value,
                              o->jacobian, o->heatmap, s, c->head_ws));
    return EAMM_OK;
}

int check_call(eamm_kp_ctx* c, const void* in, int B, const eamm_kp_outputs* o) {
    if (!c || !in || !o || !o->value) return fail(c, EAMM_ERR_ARG, "null argument");
    if (!c->finalized) return fail(c, EAMM_ERR_STATE, "call eamm_kp_finalize_weights first");
    if (B < 1 || B > c->cfg.max_batch) return fail(c, EAMM_ERR_ARG, "batch %d outside [1,%d]", B, c->cfg.max_batch);
    if (o->jacobian && c->njm == 0) return fail(c, EAMM_ERR_ARG, "jacobian requested but estimate_jacobian is off");
    return EAMM_OK;
}

}  // namespace

extern "C" {

const char* eamm_kp_last_error(const eamm_kp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int eamm_kp_create(const eamm_kp_config* cfg, int device, eamm_kp_ctx** out) {
    if (!cfg || !out) return fail(nullptr, EAMM_ERR_ARG, "null argument");
    *out = nullptr;
    const eamm_kp_config& g = *cfg;
    if (g.num_channels < 1 || g.num_channels > 8) return fail(nullptr, EAMM_ERR_ARG, "num_channels must be 1 .. 8 (got %d)", g.num_channels);
    // the generator takes 1 .. 30 key points (eamm_create); the detectors feed it, so the same range holds here (round 6: the heads'
    // K + 4 K logits are no longer limited to one 64-float line)
    if (g.num_kp < 1 || g.num_kp > 30) return fail(nullptr, EAMM_ERR_ARG, "num_kp must be 1 .. 30 (got %d)", g.num_kp);
    if (g.block_expansion % 32 || g.max_features % 32 || g.num_blocks < 1)
        return fail(nullptr, EAMM_ERR_ARG, "channel widths must be multiples of 32");
    if (g.inv_scale != 1 && g.inv_scale != 2 && g.inv_scale != 4) return fail(nullptr, EAMM_ERR_ARG, "1/scale_factor must be 1, 2 or 4");
    if (g.pad != 0 && g.pad != 3) return fail(nullptr, EAMM_ERR_ARG, "pad must be 0 (reference default) or 3");
    if (!(g.temperature > 0.f)) return fail(nullptr, EAMM_ERR_ARG, "temperature must be positive");
    if (g.max_batch < 1) return fail(nullptr, EAMM_ERR_ARG, "max_batch < 1");
    const int div = g.inv_scale << (g.with_predictor ? g.num_blocks : 0);
    if (g.height % div || g.width % div || g.height / g.inv_scale < 8 || g.width / g.inv_scale < 8)
        return fail(nullptr, EAMM_ERR_ARG, "frame %dx%d not divisible for scale 1/%d and %d hourglass levels", g.height,
                    g.width, g.inv_scale, g.num_blocks);
    {   // validate the device without leaving it selected in the caller's thread
        DeviceGuard probe(device);
        if (probe.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice(%d) failed", device);
    }
    eamm_kp_ctx* c = new eamm_kp_ctx();
    c->cfg = g;
    c->device = device;
    c->H = g.height;
    c->W = g.width;
    c->h = c->H / g.inv_scale;
    c->w = c->W / g.inv_scale;
    c->K = g.num_kp;
    c->nb = g.num_blocks;
    c->njm = g.estimate_jacobian ? (g.single_jacobian_map ? 1 : g.num_kp) : 0;
    c->head_cs = (c->K + 4 * c->njm + 63) / 64 * 64;
    c->feat_c = g.block_expansion + g.in_features;
    for (int i = 0; i < c->nb; ++i) c->enc_c.push_back(std::min(g.max_features, g.block_expansion << (i + 1)));
    for (int i = c->nb - 1; i >= 0; --i) c->dec_c.push_back(std::min(g.max_features, g.block_expansion << i));
    read_tile_knobs(c);
    *out = c;
    return EAMM_OK;
}

void eamm_kp_destroy(eamm_kp_ctx* c) {
    if (!c) return;
    free_owned(c);
    delete c;
}

int eamm_kp_load_tensor(eamm_kp_ctx* c, const char* key, const float* host, const int64_t* shape, int ndim) {
    return store_tensor(c, key, host, shape, ndim);
}

int eamm_kp_finalize_weights(eamm_kp_ctx* c) {
    if (!c) return EAMM_ERR_ARG;
    if (c->finalized) return fail(c, EAMM_ERR_STATE, "weights already finalised");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const eamm_kp_config& g = c->cfg;
    {   // key set: the heads always; the predictor hourglass + anti-alias buffer only when this handle runs them
        std::vector<std::string> want = {"kp.weight", "kp.bias"};
        if (c->njm) {
            want.push_back("jacobian.weight");
            want.push_back("jacobian.bias");
        }
        if (g.with_predictor) {
            for (int i = 0; i < c->nb; ++i) block_keys("predictor.encoder.down_blocks." + std::to_string(i), &want);
            for (int i = 0; i < c->nb; ++i) block_keys("predictor.decoder.up_blocks." + std::to_string(i), &want);
            if (g.inv_scale != 1) want.push_back("down.weight");
        }
        if (int rc = check_keys(c, want)) return rc;
    }
    int rc;
    const int cin = g.in_features;
    c->Cin_pad = (cin + 31) / 32 * 32;
    std::vector<FoldSpec> parts = {{"kp", ""}};
    if (c->njm) parts.push_back({"jacobian", ""});
    // LDS-DMA twin of the heads' 7x7 convolution: N = K + 4 K logits in one 64-column tile -> 512 x 64 (id 3), up to 128 -> 256 x 128 (id 2)
    c->head_dma_min_m = env_int("EAMM_KP_HEAD_DMA_MIN_M", c->head_dma_min_m);
    const int head_dma_cfg = c->head_dma_min_m < 0 ? 0 : (c->head_cs == 64 ? 3 : (c->head_cs == 128 ? 2 : 0));
    c->has_head_dma = head_dma_cfg != 0;
    if (g.with_predictor) {
        c->hg_enc.resize(c->nb);
        c->hg_dec.resize(c->nb);
        for (int i = 0; i < c->nb; ++i) {
            const std::string p = "predictor.encoder.down_blocks." + std::to_string(i);
            const int cr = i == 0 ? cin : c->enc_c[i - 1], cp = i == 0 ? c->Cin_pad : c->enc_c[i - 1];
            if ((rc = build_set(c, {{p + ".conv", p + ".norm"}}, cr, cp, 0, 0, &c->hg_enc[i], MODE_PLAIN))) return rc;
        }
        for (int i = 0; i < c->nb; ++i) {
            const std::string p = "predictor.decoder.up_blocks." + std::to_string(i);
            const int c0 = i == 0 ? c->enc_c[c->nb - 1] : c->dec_c[i - 1];
            const int c1 = i == 0 ? 0 : c->enc_c[c->nb - 1 - i];
            if ((rc = build_set(c, {{p + ".conv", p + ".norm"}}, c0, c0, c1, c1, &c->hg_dec[i], MODE_PHASE))) return rc;
        }
        // head input = cat[last up output (block_expansion), hourglass input (cin, stored padded)]  util.py:981-987
        if ((rc = build_layer(c, parts, 7, c->dec_c.back(), c->dec_c.back(), cin, c->Cin_pad, &c->head))) return rc;
        if (head_dma_cfg && (rc = build_layer(c, parts, 7, c->dec_c.back(), c->dec_c.back(), cin, c->Cin_pad, &c->head_dma, MODE_PLAIN, nullptr, head_dma_cfg))) return rc;
        std::vector<float> aa((size_t)std::max(3, g.num_channels) * 169, 0.f);
        if (g.inv_scale != 1) {
            const HostTensor* t = find(c, "down.weight");
            if (!t || (int)t->numel() != g.num_channels * 169)
                return fail(c, EAMM_ERR_KEY, "down.weight must be [%d,1,13,13]", g.num_channels);
            std::copy(t->data.begin(), t->data.end(), aa.begin());   // (the planes an image does not have keep zero filters)
        }
        if ((rc = upload(c, &c->aa_w, aa))) return rc;
        if (g.num_channels < 3) {   // one / two image channels: the hourglass's first block has filters for those only (cin above)
            const size_t n = (size_t)g.max_batch * 3 * c->H * c->W;
            if ((rc = dev_alloc(c, &c->img_stage, n))) return rc;
            HIP_TRY(c, hipMemset(c->img_stage, 0, n * sizeof(float)));
        }
    } else {
        // KPDetector_a: the caller's feature map already is the hourglass output (feat_c channels)
        const int cp = (c->feat_c + 31) / 32 * 32;
        const int wide = c->feat_c - 3;
        if (wide >= 32 && wide % 32 == 0 && c->head_cs == 64 && env_int("EAMM_KPA_THIN", 1)) {   // (the thin kernel's N is one 64-column tile)
            // slice every head's filter [co, feat_c, 7, 7] into its wide part (replaces the entry) and its last three input channels
            std::vector<float> thin((size_t)64 * 3 * 49, 0.f);
            int o0 = 0;
            for (auto& ps : parts) {
                auto it = c->sd.find(ps.conv + ".weight");
                if (it == c->sd.end() || it->second.shape.size() != 4 || it->second.shape[1] != c->feat_c || it->second.shape[2] != 7 ||
                    it->second.shape[3] != 7)
                    return fail(c, EAMM_ERR_KEY, "state_dict entry %s.weight missing or mis-shaped (expected [*,%d,7,7])", ps.conv.c_str(), c->feat_c);
                HostTensor& wt = it->second;
                const int co = (int)wt.shape[0];
                if (o0 + co > 64) return fail(c, EAMM_ERR_KEY, "kp / jacobian heads have more than 64 output channels");
                HostTensor nw;
                nw.shape = {co, wide, 7, 7};
                nw.data.resize((size_t)co * wide * 49);
                for (int o = 0; o < co; ++o) {
                    std::copy(wt.data.begin() + (size_t)o * c->feat_c * 49, wt.data.begin() + ((size_t)o * c->feat_c + wide) * 49,
                              nw.data.begin() + (size_t)o * wide * 49);
                    std::copy(wt.data.begin() + ((size_t)o * c->feat_c + wide) * 49, wt.data.begin() + (size_t)(o + 1) * c->feat_c * 49,
                              thin.begin() + (size_t)(o0 + o) * 3 * 49);
                }
                wt = std::move(nw);
                o0 += co;
            }
            if ((rc = build_layer(c, parts, 7, wide, wide, 0, 0, &c->head))) return rc;
            if (head_dma_cfg && (rc = build_layer(c, parts, 7, wide, wide, 0, 0, &c->head_dma, MODE_PLAIN, nullptr, head_dma_cfg))) return rc;
            if ((rc = upload(c, &c->thin_w, thin))) return rc;
            c->thin_wide = wide;
        } else {
            if ((rc = build_layer(c, parts, 7, c->feat_c, cp, 0, 0, &c->head))) return rc;
            if (head_dma_cfg && (rc = build_layer(c, parts, 7, c->feat_c, cp, 0, 0, &c->head_dma, MODE_PLAIN, nullptr, head_dma_cfg))) return rc;
        }
    }
    if (c->head.Cout != c->K + 4 * c->njm) return fail(c, EAMM_ERR_KEY, "kp / jacobian heads have the wrong channel count");

    const size_t F = g.max_batch, hw = (size_t)c->h * c->w;
    const int xin_c = g.with_predictor ? c->Cin_pad : (c->thin_wide ? c->thin_wide : (c->feat_c + 31) / 32 * 32);
    if ((rc = dev_alloc(c, &c->x_in, F * hw * xin_c))) return rc;
    if (c->thin_wide) {
        if ((rc = dev_alloc(c, &c->thin_x, F * hw * 4))) return rc;
        if ((rc = dev_alloc(c, &c->thin_y, F * hw * 64))) return rc;
        if ((rc = dev_alloc(c, &c->thin_ws, conv7_thin_workspace_floats((int)F, c->h, c->w, 64)))) return rc;
    }
    if ((rc = dev_alloc(c, &c->logits, F * hw * c->head_cs))) return rc;
    if ((rc = dev_alloc(c, &c->head_ws, kp_head_workspace_floats((int)F, c->K)))) return rc;
    size_t need = 0;
    auto upd1 = [&](const ConvLayer& L, size_t M) { need = std::max(need, conv_plan(L, (int)M).partial_elems); };
    if (g.with_predictor) {
        c->e_buf.resize(c->nb);
        c->u_buf.resize(c->nb);
        for (int i = 0; i < c->nb; ++i) {
            if ((rc = dev_alloc(c, &c->e_buf[i], F * (hw >> (2 * (i + 1))) * c->enc_c[i]))) return rc;
            if ((rc = dev_alloc(c, &c->u_buf[i], F * (hw >> (2 * (c->nb - 1 - i))) * c->dec_c[i]))) return rc;
        }
        for (size_t f = 1; f <= F; ++f)
            for (int i = 0; i < c->nb; ++i)
                for (const LayerSet* S : {&c->hg_enc[i], &c->hg_dec[i]}) {
                    const size_t M = S == &c->hg_enc[i] ? f * (hw >> (2 * i)) : f * (hw >> (2 * (c->nb - i)));
                    upd1(S->base, M);
                    if (S->has_dma) upd1(S->dma, M);
                    if (S->has_big) upd1(S->big, M);
                    if (S->has_skinny) {
                        upd1(S->skinny32, M);
                        upd1(S->skinny64, M);
                    }
                }
    }
    for (size_t f = 1; f <= F; ++f) {
        upd1(c->head, f * hw);
        if (c->has_head_dma) upd1(c->head_dma, f * hw);
    }
    c->partial_elems = need;
    if ((rc = dev_alloc(c, &c->partial, need))) return rc;
    c->sd.clear();
    HIP_TRY(c, hipDeviceSynchronize());
    c->finalized = true;
    return EAMM_OK;
}

int eamm_kp_detect(eamm_kp_ctx* c, const float* image, int B, const eamm_kp_outputs* o, void* stream_) {
    if (int rc = check_call(c, image, B, o)) return rc;
    if (!c->cfg.with_predictor) return fail(c, EAMM_ERR_STATE, "this handle was created without the predictor hourglass");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const int h = c->h, w = c->w;
    if (c->img_stage) {   // [B,C,H,W] into the first C planes of [B,3,H,W]
        const size_t HW = (size_t)c->H * c->W * sizeof(float), C = (size_t)c->cfg.num_channels;
        HIP_TRY(c, hipMemcpy2DAsync(c->img_stage, 3 * HW, image, C * HW, C * HW, (size_t)B, hipMemcpyDeviceToDevice, s));
        image = c->img_stage;
    }
    // x = down(x): anti-aliased, NHWC zero-padded to Cin_pad channels            keypoint_detector.py:79-80
    if (c->cfg.num_channels <= 3) {
        HIP_TRY(c, antialias_down_launch(image, c->aa_w, B, c->H, c->W, c->cfg.inv_scale, c->Cin_pad, c->x_in, s));
    } else {   // four to eight channels: four per float4 slot, straight from the caller's [B,C,H,W] (real channels stay contiguous)
        const int C = c->cfg.num_channels;
        for (int g0 = 0, slot = 0; g0 < C; g0 += 4, ++slot)
            HIP_TRY(c, antialias_down_launch(image, c->aa_w, B, c->H, c->W, c->cfg.inv_scale, c->Cin_pad, c->x_in, s, C, g0,
                                             std::min(4, C - g0), slot));
    }
    for (int i = 0; i < c->nb; ++i) {   // hourglass encoder                      util.py:956-960
        ConvIO io{};
        io.in0 = i == 0 ? c->x_in : c->e_buf[i - 1];
        io.B = B;
        io.Hin = h >> i;
        io.Win = w >> i;
        io.act = ACT_RELU;
        io.pool = 1;
        io.out = c->e_buf[i];
        io.partial = c->partial;
        io.partial_cap = c->partial_elems;
        HIP_TRY(c, conv_launch(pick(c, c->hg_enc[i], (size_t)B * io.Hin * io.Win), io, s));
    }
    for (int i = 0; i < c->nb; ++i) {   // hourglass decoder                      util.py:981-987
        ConvIO io{};
        io.in0 = i == 0 ? c->e_buf[c->nb - 1] : c->u_buf[i - 1];
        io.in1 = i == 0 ? nullptr : c->e_buf[c->nb - 1 - i];
        io.B = B;
        io.Hin = h >> (c->nb - i);
        io.Win = w >> (c->nb - i);
        io.act = ACT_RELU;
        io.out = c->u_buf[i];
        io.partial = c->partial;
        io.partial_cap = c->partial_elems;
        if (int urc = launch_up(c, c->hg_dec[i], io, s)) return urc;
    }
    return run_head(c, c->u_buf[c->nb - 1], c->x_in, B, o, s);   // keypoint_detector.py:83-103
}

int eamm_kp_detect_features(eamm_kp_ctx* c, const float* feature_map, int B, const eamm_kp_outputs* o, void* stream_) {
    if (int rc = check_call(c, feature_map, B, o)) return rc;
    if (c->cfg.with_predictor) return fail(c, EAMM_ERR_STATE, "this handle runs the predictor; use eamm_kp_detect");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (c->thin_wide) {   // wide part -> NHWC, last three channels -> one float4 per pixel, thin 7x7 -> added in the wide one's epilogue
        HIP_TRY(c, nchw_to_nhwc_pad_launch(feature_map, B, c->thin_wide, c->h, c->w, c->thin_wide, c->x_in, s, c->feat_c));
        HIP_TRY(c, antialias_down_launch(feature_map, nullptr, B, c->h, c->w, 1, 4, c->thin_x, s, c->feat_c, c->thin_wide, 3, 0));
        HIP_TRY(c, conv7_thin_in_launch(c->thin_x, c->thin_w, nullptr, B, c->h, c->w, 64, 0, c->thin_y, c->thin_ws, s));
        return run_head(c, c->x_in, nullptr, B, o, s, c->thin_y);   // keypoint_detector.py:180-203
    }
    const int cp = (c->feat_c + 31) / 32 * 32;
    HIP_TRY(c, nchw_to_nhwc_pad_launch(feature_map, B, c->feat_c, c->h, c->w, cp, c->x_in, s));
    return run_head(c, c->x_in, nullptr, B, o, s);               // keypoint_detector.py:180-203
}

// Round 6: the feature map as DeconvTail hands it over privately (eamm_deconv_forward_split): `wide` NHWC [B,h,w,32 m] + `thin` one
// float4 per pixel -- the two operands the heads read, so no layout kernel runs here.  eamm_kp_split_channels: 32 m, or 0 when this
// handle's heads do not run in the wide + thin form.
int eamm_kp_split_channels(const eamm_kp_ctx* c) { return (c && c->finalized && !c->cfg.with_predictor) ? c->thin_wide : 0; }

int eamm_kp_detect_features_split(eamm_kp_ctx* c, const float* wide, const float* thin, int B, const eamm_kp_outputs* o, void* stream_) {
    if (int rc = check_call(c, wide, B, o)) return rc;
    if (!thin) return fail(c, EAMM_ERR_ARG, "null argument");
    if (c->cfg.with_predictor || !c->thin_wide) return fail(c, EAMM_ERR_STATE, "this handle's heads do not take the split feature map");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(c, conv7_thin_in_launch(thin, c->thin_w, nullptr, B, c->h, c->w, 64, 0, c->thin_y, c->thin_ws, s));
    return run_head(c, wide, nullptr, B, o, s, c->thin_y);   // keypoint_detector.py:180-203
}

}  // extern "C"
