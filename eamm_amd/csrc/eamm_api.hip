// C ABI of libeamm_hip.so (see include/eamm_hip.h): handle, strict state_dict intake, BatchNorm folding
// and weight repacking, workspace, and the launch sequence of the two halves of
// OcclusionAwareGenerator.forward (reference modules/generator.py:59-97):
//   eamm_encode_source  -- frame-invariant: first 7x7 block, down blocks, anti-alias down-sampling
//   eamm_forward_frames -- per frame batch: key-point records, motion front end, hourglass, flow head,
//                          feature warp, bottleneck, up blocks, final 7x7 + sigmoid
#include "eamm_ctx.h"
#include <mutex>

namespace {

void expected_keys(const eamm_ctx* c, std::vector<std::string>* keys) {
    auto block = [&](const std::string& p) { block_keys(p, keys); };
    const std::string dm = "dense_motion_network.";
    for (int i = 0; i < c->nb; ++i) block(dm + "hourglass.encoder.down_blocks." + std::to_string(i));
    for (int i = 0; i < c->nb; ++i) block(dm + "hourglass.decoder.up_blocks." + std::to_string(i));
    if (c->nb > 0) {
        keys->push_back(dm + "mask.weight");
        keys->push_back(dm + "mask.bias");
        if (c->cfg.estimate_occlusion_map) {
            keys->push_back(dm + "occlusion.weight");
            keys->push_back(dm + "occlusion.bias");
        }
        if (c->cfg.dm_inv_scale != 1) keys->push_back(dm + "down.weight");
    }
    block("first");
    for (int i = 0; i < c->nd; ++i) block("down_blocks." + std::to_string(i));
    for (int i = 0; i < c->nd; ++i) block("up_blocks." + std::to_string(i));
    for (int i = 0; i < c->cfg.num_bottleneck_blocks; ++i) {
        const std::string r = "bottleneck.r" + std::to_string(i);
        for (const char* s : {".conv1.weight", ".conv1.bias", ".conv2.weight", ".conv2.bias", ".norm1.weight",
                              ".norm1.bias", ".norm1.running_mean", ".norm1.running_var", ".norm2.weight",
                              ".norm2.bias", ".norm2.running_mean", ".norm2.running_var"})
            keys->push_back(r + s);
    }
    keys->push_back("final.weight");
    keys->push_back("final.bias");
}

double conv_flops(int ks, int cin, int cout, double pixels) { return 2.0 * ks * ks * (double)cin * cout * pixels; }

// The F(4x4) kernels address V ([36][tiles][Cin]) and Z ([24][tiles][Cout]) through 32-bit buffer descriptors
// (wino4_gemm_launch refuses from 0xFFFFF000 bytes): a call with more tiles than this takes the direct kernels.
bool wino4_fits(size_t tiles, int cin, int cout) {
    return 36.0 * tiles * cin * 4.0 < 4294963200.0 && 24.0 * tiles * cout * 4.0 < 4294963200.0;
}


// ---- widths that are not multiples of 32 (reference modules/generator.py:14-48 accepts any): the state_dict is padded into
// the equivalent network whose widths are the next multiples of 32 -- extra output channels get zero filters and a zero bias,
// their BatchNorm is the identity (weight 1, bias 0, mean 0, variance 1), extra input channels get zero filter columns -- so
// the extra channels carry exact zeros through ReLU, pooling, the residual adds and the warps, and every layer builder and
// kernel below runs unchanged on multiples of 32.
struct PadPart { int real, packed; };
int pad_conv(eamm_ctx* c, const std::string& conv, int co_r, int co_p, const std::vector<PadPart>& in) {
    auto wi = c->sd.find(conv + ".weight");
    auto bi = c->sd.find(conv + ".bias");
    if (wi == c->sd.end() || bi == c->sd.end()) return fail(c, EAMM_ERR_KEY, "state_dict entry %s.weight / .bias missing", conv.c_str());
    HostTensor& w = wi->second;
    int cin_r = 0, cin_p = 0;
    for (auto& p : in) { cin_r += p.real; cin_p += p.packed; }
    if (w.shape.size() != 4 || w.shape[0] != co_r || w.shape[1] != cin_r || (int)bi->second.numel() != co_r)
        return fail(c, EAMM_ERR_KEY, "state_dict entry %s.weight mis-shaped (expected [%d,%d,k,k])", conv.c_str(), co_r, cin_r);
    const int T = (int)(w.shape[2] * w.shape[3]);
    HostTensor nw;
    nw.shape = {co_p, cin_p, w.shape[2], w.shape[3]};
    nw.data.assign((size_t)co_p * cin_p * T, 0.f);
    for (int o = 0; o < co_r; ++o) {
        int src = 0, dst = 0;
        for (auto& p : in) {
            std::copy(w.data.begin() + ((size_t)o * cin_r + src) * T, w.data.begin() + ((size_t)o * cin_r + src + p.real) * T,
                      nw.data.begin() + ((size_t)o * cin_p + dst) * T);
            src += p.real;
            dst += p.packed;
        }
    }
    w = std::move(nw);
    HostTensor& b = bi->second;
    b.data.resize(co_p, 0.f);
    b.shape = {co_p};
    return 0;
}
int pad_norm(eamm_ctx* c, const std::string& norm, int c_r, int c_p) {
    for (const char* leaf : {".weight", ".bias", ".running_mean", ".running_var"}) {
        auto it = c->sd.find(norm + leaf);
        if (it == c->sd.end() || (int)it->second.numel() != c_r)
            return fail(c, EAMM_ERR_KEY, "BatchNorm entry %s%s missing or mis-shaped (expected %d)", norm.c_str(), leaf, c_r);
        const bool one = !strcmp(leaf, ".weight") || !strcmp(leaf, ".running_var");
        it->second.data.resize(c_p, one ? 1.f : 0.f);
        it->second.shape = {c_p};
    }
    return 0;
}
// The chains' side streams come from a per-device pool shared by every handle of the process (created on first use, never
// destroyed): the runtime multiplexes streams onto a few hardware queues (four by default), and a second handle with its own
// three streams made two chains of one call share a queue -- measured: a 64-frame call's four chains 4199 frames/s alone, 4007
// beside another handle's streams.  The pool is LEASED for the duration of one call's enqueue (StreamLease below): a call that
// finds it taken by another host thread, or whose caller's stream is being captured into a HIP graph (the side streams join
// that capture until it ends, so nobody else may touch them meanwhile), runs on the handle's own private streams instead.
// Calls of different handles that follow one another on the pool share the streams' ORDER, never their dependencies: every
// call forks from and joins to its caller's stream with its own events.
constexpr int POOL_MAXDEV = 64, POOL_MAXCHAIN = 15;
std::mutex g_pool_mu;
hipStream_t g_pool[POOL_MAXDEV][POOL_MAXCHAIN] = {};
std::mutex g_pool_lease[POOL_MAXDEV];

hipStream_t chain_stream(int device, int k) {
    if (device < 0 || device >= POOL_MAXDEV || k < 0 || k >= POOL_MAXCHAIN) return nullptr;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (!g_pool[device][k] && hipStreamCreateWithFlags(&g_pool[device][k], hipStreamNonBlocking) != hipSuccess) g_pool[device][k] = nullptr;
    return g_pool[device][k];
}

// Picks the side streams of ONE eamm_forward_frames call (c->side_streams) and holds the pool's lease while the call enqueues.
struct StreamLease {
    eamm_ctx* c;
    bool held = false;
    int rc = EAMM_OK;
    StreamLease(eamm_ctx* ctx, hipStream_t caller) : c(ctx) {
        if (c->pool_streams.empty()) return;   // a handle without chains
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(caller, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        const bool capturing = cap != hipStreamCaptureStatusNone;
        const bool in_range = c->device >= 0 && c->device < POOL_MAXDEV;
        if (!capturing && !c->private_streams && in_range && g_pool_lease[c->device].try_lock()) {
            held = true;
            c->side_streams = c->pool_streams;
            c->last_streams = 0;
            return;
        }
        // (the private twins are created with the pool streams in eamm_finalize_weights, outside any capture)
        if (c->own_streams.size() < c->pool_streams.size()) {
            rc = fail(c, EAMM_ERR_STATE, "handle has no private chain streams (weights not finalized?)");
            return;
        }
        c->side_streams = c->own_streams;
        c->last_streams = capturing ? 2 : 1;
    }
    ~StreamLease() {
        if (held) g_pool_lease[c->device].unlock();
    }
};

int pad_state_dict(eamm_ctx* c) {
    const std::string dm = "dense_motion_network.";
    const int nb = c->nb, nd = c->nd;
    // the hourglass input line: per motion (heat-map, image channels) -- C + 1 values of the reference land in the kernels' four
    const std::vector<PadPart> motion_line((size_t)c->K + 1, PadPart{c->Cimg + 1, 4 * c->G});
    int rc;
#define PAD_TRY(e) if ((rc = (e))) return rc
    for (int i = 0; i < nb; ++i) {
        const std::string p = dm + "hourglass.encoder.down_blocks." + std::to_string(i);
        const std::vector<PadPart> in = i == 0 ? motion_line : std::vector<PadPart>{PadPart{c->enc_r[i - 1], c->enc_c[i - 1]}};
        PAD_TRY(pad_conv(c, p + ".conv", c->enc_r[i], c->enc_c[i], in));
        PAD_TRY(pad_norm(c, p + ".norm", c->enc_r[i], c->enc_c[i]));
    }
    for (int i = 0; i < nb; ++i) {
        const std::string p = dm + "hourglass.decoder.up_blocks." + std::to_string(i);
        std::vector<PadPart> in;
        if (i == 0) in = {{c->enc_r[nb - 1], c->enc_c[nb - 1]}};
        else in = {{c->dec_r[i - 1], c->dec_c[i - 1]}, {c->enc_r[nb - 1 - i], c->enc_c[nb - 1 - i]}};
        PAD_TRY(pad_conv(c, p + ".conv", c->dec_r[i], c->dec_c[i], in));
        PAD_TRY(pad_norm(c, p + ".norm", c->dec_r[i], c->dec_c[i]));
    }
    if (nb > 0) {
        std::vector<PadPart> in = {{c->dec_r.back(), c->dec_c.back()}};
        in.insert(in.end(), motion_line.begin(), motion_line.end());
        PAD_TRY(pad_conv(c, dm + "mask", c->K + 1, c->K + 1, in));
        if (c->cfg.estimate_occlusion_map) PAD_TRY(pad_conv(c, dm + "occlusion", 1, 1, in));
        if (c->cfg.dm_inv_scale != 1 && c->Cimg != 3) {   // anti-alias buffer [C,1,13,13] (util.py:1038): one filter per channel
            auto it = c->sd.find(dm + "down.weight");
            if (it == c->sd.end() || (int)it->second.numel() != c->Cimg * 169)
                return fail(c, EAMM_ERR_KEY, "%sdown.weight must be [%d,1,13,13]", dm.c_str(), c->Cimg);
            it->second.data.resize((size_t)c->Cpl * 169, 0.f);
            it->second.shape = {c->Cpl, 1, 13, 13};
        }
    }
    PAD_TRY(pad_conv(c, "first.conv", c->down_r[0], c->down_c[0], {{c->Cimg, c->Cpl}}));
    PAD_TRY(pad_norm(c, "first.norm", c->down_r[0], c->down_c[0]));
    for (int i = 0; i < nd; ++i) {
        const std::string p = "down_blocks." + std::to_string(i);
        PAD_TRY(pad_conv(c, p + ".conv", c->down_r[i + 1], c->down_c[i + 1], {{c->down_r[i], c->down_c[i]}}));
        PAD_TRY(pad_norm(c, p + ".norm", c->down_r[i + 1], c->down_c[i + 1]));
    }
    for (int i = 0; i < c->cfg.num_bottleneck_blocks; ++i) {
        const std::string r = "bottleneck.r" + std::to_string(i);
        for (const char* k : {".conv1", ".conv2"}) PAD_TRY(pad_conv(c, r + k, c->Cb_r, c->Cb, {{c->Cb_r, c->Cb}}));
        for (const char* k : {".norm1", ".norm2"}) PAD_TRY(pad_norm(c, r + k, c->Cb_r, c->Cb));
    }
    for (int i = 0; i < nd; ++i) {
        const std::string p = "up_blocks." + std::to_string(i);
        const PadPart in = i == 0 ? PadPart{c->Cb_r, c->Cb} : PadPart{c->up_r[i - 1], c->up_c[i - 1]};
        PAD_TRY(pad_conv(c, p + ".conv", c->up_r[i], c->up_c[i], {in}));
        PAD_TRY(pad_norm(c, p + ".norm", c->up_r[i], c->up_c[i]));
    }
    // (more than three image channels: `final` keeps its C outputs and runs on the generic kernel, see eamm_finalize_weights)
    PAD_TRY(pad_conv(c, "final", c->Cimg, c->Cimg > 3 ? c->Cimg : 3, {{c->up_r.back(), c->up_c.back()}}));
#undef PAD_TRY
    return 0;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
extern "C" {

int eamm_abi_version(void) { return EAMM_ABI_VERSION; }

const char* eamm_last_error(const eamm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int eamm_create(const eamm_config* cfg, int device, eamm_ctx** out) {
    if (!cfg || !out) return fail(nullptr, EAMM_ERR_ARG, "null argument");
    *out = nullptr;
    const eamm_config& g = *cfg;
    if (g.num_channels < 1 || g.num_channels > 6)
        return fail(nullptr, EAMM_ERR_ARG, "num_channels must be 1 .. 6 (got %d): the motion kernels keep a pixel's image channels in "
                    "groups of three (one float4 beside its heat-map value per group), two groups at most", g.num_channels);
    if (g.num_kp < 1 || g.num_kp + 2 > 32)
        return fail(nullptr, EAMM_ERR_ARG, "num_kp must be 1 .. 30 (got %d): the flow head keeps the K + 1 motions' softmax and the "
                    "occlusion logit of a pixel in one 32-lane group", g.num_kp);
    // dm_num_blocks == 0: a generator without a motion network (dense_motion_params=None, generator.py:22-23)
    const bool has_dm = g.dm_num_blocks > 0;
    if (g.dm_num_blocks < 0) return fail(nullptr, EAMM_ERR_ARG, "dm_num_blocks < 0");
    if (has_dm && (!is_pow2(g.dm_inv_scale) || g.dm_inv_scale > 4 || g.dm_inv_scale == 3))
        return fail(nullptr, EAMM_ERR_ARG, "1/scale_factor must be 1, 2 or 4");
    if (!has_dm && g.estimate_occlusion_map) return fail(nullptr, EAMM_ERR_ARG, "estimate_occlusion_map needs a motion network");
    if (g.block_expansion < 1 || g.max_features < 1 || (has_dm && (g.dm_block_expansion < 1 || g.dm_max_features < 1)))
        return fail(nullptr, EAMM_ERR_ARG, "channel widths must be positive");
    if (g.num_down_blocks < 1 || g.num_bottleneck_blocks < 1)
        return fail(nullptr, EAMM_ERR_ARG, "need at least one down block and one bottleneck block");
    if (g.max_frames < 1 || g.max_sources < 1) return fail(nullptr, EAMM_ERR_ARG, "max_frames / max_sources < 1");
    const int div_g = 1 << g.num_down_blocks, div_m = has_dm ? g.dm_inv_scale << g.dm_num_blocks : 1;
    if (g.height % div_g || g.width % div_g || g.height % div_m || g.width % div_m)
        return fail(nullptr, EAMM_ERR_ARG, "frame %dx%d not divisible for %d down blocks / %d hourglass levels",
                    g.height, g.width, g.num_down_blocks, g.dm_num_blocks);
    {   // the kernels address tensors through 32-bit buffer descriptors: every activation must stay below 4 GiB
        const double hw = (double)g.height * g.width, F = g.max_frames;
        const double biggest = std::max({F * hw * 32.0,                                  // final-conv partial products
                                         F * hw * (double)round_up(g.block_expansion, 32),   // last up-block output
                                         4.0 * F * (hw / (1 << (2 * g.num_down_blocks))) *
                                             round_up(std::min(g.max_features, g.block_expansion << g.num_down_blocks), 32)}) * 4.0;
        if (biggest >= 4294967280.0)
            return fail(nullptr, EAMM_ERR_ARG, "max_frames=%d at %dx%d needs a %.1f GiB activation tensor; the limit is 4 GiB "
                        "per tensor -- lower max_frames", g.max_frames, g.height, g.width, biggest / 1073741824.0);
    }
    {   // validate the device without leaving it selected in the caller's thread
        DeviceGuard probe(device);
        if (probe.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice(%d) failed", device);
    }

    eamm_ctx* c = new eamm_ctx();
    c->cfg = g;
    c->device = device;
    c->H = g.height;
    c->W = g.width;
    if (!has_dm) c->cfg.dm_inv_scale = 1;
    c->h = c->H / c->cfg.dm_inv_scale;
    c->w = c->W / c->cfg.dm_inv_scale;
    c->nd = g.num_down_blocks;
    c->nb = g.dm_num_blocks;
    c->K = g.num_kp;
    c->hf = c->H >> c->nd;
    c->wf = c->W >> c->nd;
    c->Cimg = g.num_channels;
    c->G = c->Cimg > 3 ? 2 : 1;            // groups of three image channels (motion.hip: motion_front_kernel)
    c->Cpl = 3 * c->G;                     // planes of the zero-extended source copy
    c->Cp0 = round_up((c->K + 1) * 4 * c->G, 32);
    // hourglass channel plan (reference modules/util.py:941-987)
    for (int i = 0; i < c->nb; ++i) c->enc_r.push_back(std::min(g.dm_max_features, g.dm_block_expansion << (i + 1)));
    for (int i = c->nb - 1; i >= 0; --i) c->dec_r.push_back(std::min(g.dm_max_features, g.dm_block_expansion << i));
    // generator channel plan (reference modules/generator.py:27-44)
    c->down_r.push_back(g.block_expansion);
    for (int i = 0; i < c->nd; ++i) c->down_r.push_back(std::min(g.max_features, g.block_expansion << (i + 1)));
    for (int i = 0; i < c->nd; ++i) c->up_r.push_back(std::min(g.max_features, g.block_expansion << (c->nd - i - 1)));
    c->Cb_r = c->down_r.back();
    // the kernels' widths: the 32-channel granule of the MFMA operand loaders (any width the reference accepts is taken)
    auto pad32 = [&](const std::vector<int>& r, std::vector<int>* p) {
        for (int v : r) {
            p->push_back(round_up(v, 32));
            c->padded_widths |= p->back() != v;
        }
    };
    pad32(c->enc_r, &c->enc_c);
    pad32(c->dec_r, &c->dec_c);
    pad32(c->down_r, &c->down_c);
    pad32(c->up_r, &c->up_c);
    c->Cb = c->down_c.back();
    read_tile_knobs(c);
    c->wino_min_m = env_int("EAMM_WINO_MIN_M", c->wino_min_m);
    c->wino4_min_m = env_int("EAMM_WINO4_MIN_M", c->wino4_min_m);
    c->wino_variant = env_int("EAMM_WINO_VARIANT", c->wino_variant);   // pipeline variant of wino_gemm_kernel (EAMM_WINO_MIN_M < 0 disables the Winograd bottleneck)
    c->wino_tile = env_int("EAMM_WINO_TILE", c->wino_tile);
    c->col7 = env_int("EAMM_COL7", c->col7);
    c->final_fused = env_int("EAMM_FINAL_FUSED", c->final_fused);
    c->final_fused_min_rows = env_int("EAMM_FINAL_FUSED_MIN_ROWS", c->final_fused_min_rows);
    c->bneck_chains = env_int("EAMM_BNECK_CHAINS", c->bneck_chains);
    c->pass_chains = env_int("EAMM_PASS_CHAINS", c->pass_chains);
    // clamped ONCE, here: negative = off (as the other knobs), at most four chains; the split-K slab count and the runtime
    // chain count both derive from this value
    c->pass_chains = c->pass_chains < 0 ? 1 : std::min(c->pass_chains, 4);
    c->bneck_chains = std::max(1, std::min(c->bneck_chains, 16));
    c->bneck_stagger = env_int("EAMM_BNECK_STAGGER", c->bneck_stagger);
    c->warp_joint = env_int("EAMM_WARP_JOINT", c->warp_joint);
    c->enc_cus_pct = std::max(1, env_int("EAMM_ENC_CUS_PCT", c->enc_cus_pct));
    c->wino4_variant = env_int("EAMM_WINO4_VARIANT", c->wino4_variant);
    c->wino4_variant_pinned = getenv("EAMM_WINO4_VARIANT") != nullptr;   // an explicit choice holds for every call size
    // read ONCE here (both used to be looked up -- a getenv, the knob registry's mutex and map, a device query -- in every
    // forward_view of the per-frame path)
    c->wino4_groups_knob = env_int("EAMM_WINO4_GROUPS", 0);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
        else (void)hipGetLastError();
    }
#ifdef EAMM_EXPERIMENTS
    c->epi_v = env_int("EAMM_WINO4_EPI_V", c->epi_v);
#else
    // timing experiments that compute wrong results exist only in a -DEAMM_EXPERIMENTS build (make EXPERIMENTS=1): the product
    // library refuses to run with their knobs set rather than ignoring them silently (or, worse, obeying them)
    for (const char* bad : {"EAMM_WINO4_EPI_V", "EAMM_COL7_DBG"})
        if (getenv(bad) && atoi(getenv(bad)) != 0) {
            delete c;
            return fail(nullptr, EAMM_ERR_ARG, "%s is a wrong-results timing experiment: it needs a library built with make EXPERIMENTS=1", bad);
        }
    if (c->wino4_variant == 10 || c->wino4_variant == 16 || c->wino4_variant == 17 || c->wino4_variant == 50) {
        const int v = c->wino4_variant;
        delete c;
        return fail(nullptr, EAMM_ERR_ARG, "EAMM_WINO4_VARIANT=%d is a wrong-results timing experiment: it needs make EXPERIMENTS=1", v);
    }
#endif
    c->pass_chains_min_frames = env_int("EAMM_PASS_CHAINS_MIN_FRAMES", c->pass_chains_min_frames);
    c->pass_chains_min_blocks = env_int("EAMM_PASS_CHAINS_MIN_BLOCKS", c->pass_chains_min_blocks);
    c->head_col7_min_tiles = env_int("EAMM_HEAD_COL7_MIN_TILES", c->head_col7_min_tiles);
    c->enc_wino = env_int("EAMM_ENC_WINO", c->enc_wino);
    c->first7 = env_int("EAMM_FIRST7", c->first7);
    c->enc_wino_min_mflop = env_int("EAMM_ENC_WINO_MIN_MFLOP", c->enc_wino_min_mflop);
    c->enc_wino_min_tiles = env_int("EAMM_ENC_WINO_MIN_TILES", c->enc_wino_min_tiles);
    c->dma_cfg_n256 = env_int("EAMM_DMA_CFG_N256", c->dma_cfg_n256);
    c->dma_cfg_n128 = env_int("EAMM_DMA_CFG_N128", c->dma_cfg_n128);
    c->dma_cfg_n64 = env_int("EAMM_DMA_CFG_N64", c->dma_cfg_n64);
    *out = c;
    return EAMM_OK;
}

void eamm_destroy(eamm_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    free_owned(c);
    for (auto& e : c->prof_events) (void)hipEventDestroy(e);
    for (auto& e : c->prof_chain_ev) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_stagger) (void)hipEventDestroy(c->ev_stagger);
    if (c->ev_warp) (void)hipEventDestroy(c->ev_warp);
    for (auto& e : c->ev_join) (void)hipEventDestroy(e);
    for (auto& st : c->own_streams) (void)hipStreamDestroy(st);   // (drains the stream's work first; the pool's streams stay)
    delete c;
}

int eamm_load_tensor(eamm_ctx* c, const char* key, const float* host, const int64_t* shape, int ndim) {
    return store_tensor(c, key, host, shape, ndim);
}

int eamm_finalize_weights(eamm_ctx* c) {
    if (!c) return EAMM_ERR_ARG;
    if (c->finalized) return fail(c, EAMM_ERR_STATE, "weights already finalised");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const eamm_config& g = c->cfg;
    {
        std::vector<std::string> want;
        expected_keys(c, &want);
        if (int krc = check_keys(c, want)) return krc;
    }
    if (c->padded_widths || c->Cimg != 3) {
        if (c->train_mode)
            return fail(c, EAMM_ERR_ARG, "a training-mode handle needs three image channels and channel widths that are multiples of 32 "
                        "(its BatchNorm kernels read the module's own statistics tensors); the module takes the operator composition "
                        "for such a generator");
        if (int prc = pad_state_dict(c)) return prc;
    }
    const std::string dm = "dense_motion_network.";
    int rc;
    // training mode (eamm_set_training): batch statistics cannot be folded -- every convolution is packed with its raw
    // weights and the BatchNorm runs as its own kernels (eamm_train_api.hip)
    const bool tr = c->train_mode;
    auto nm = [&](const std::string& norm) { return tr ? std::string() : norm; };
    // hourglass encoder: e_0 = 44-channel motion tensor (padded to Cp0), e_i = DownBlock2d_i(e_{i-1})
    c->hg_enc.resize(c->nb);
    const int cin0 = (c->K + 1) * 4 * c->G;
    for (int i = 0; i < c->nb; ++i) {
        const std::string p = dm + "hourglass.encoder.down_blocks." + std::to_string(i);
        const int cr = i == 0 ? cin0 : c->enc_c[i - 1], cp = i == 0 ? c->Cp0 : c->enc_c[i - 1];
        if ((rc = build_set(c, {{p + ".conv", nm(p + ".norm")}}, cr, cp, 0, 0, &c->hg_enc[i], MODE_PLAIN))) return rc;
    }
    // hourglass decoder: u_i = UpBlock2d_i(cat[u_{i-1}, e_{nb-i}])  (util.py:981-987)
    c->hg_dec.resize(c->nb);
    for (int i = 0; i < c->nb; ++i) {
        const std::string p = dm + "hourglass.decoder.up_blocks." + std::to_string(i);
        const int c0 = i == 0 ? c->enc_c[c->nb - 1] : c->dec_c[i - 1];
        const int c1 = i == 0 ? 0 : c->enc_c[c->nb - 1 - i];
        if ((rc = build_set(c, {{p + ".conv", nm(p + ".norm")}}, c0, c0, c1, c1, &c->hg_dec[i], MODE_PHASE))) return rc;
    }
    // flow head: mask (K+1) and occlusion (1) 7x7 convolutions share one launch (dense_motion.py:98,110)
    if (c->nb > 0) {
        std::vector<FoldSpec> parts = {{dm + "mask", ""}};
        if (g.estimate_occlusion_map) parts.push_back({dm + "occlusion", ""});
        const int nc = c->K + 1 + (g.estimate_occlusion_map ? 1 : 0);
        if (7 * nc <= 128 && env_int("EAMM_HEAD_ROWSPLIT", 1)) {
            // 7x1 MFMA convolution with N = (dx, co) + horizontal gather in the head kernel: pads N to 7*nc (84) of
            // 128 instead of nc (12) of 32 -> 2.3x fewer executed MACs
            std::vector<float> hb;
            const bool col7s_ok = c->col7 && 7 * nc <= 96 && c->dec_c.back() % 32 == 0 && c->Cp0 % 32 == 0;
            if ((rc = build_layer(c, parts, 7, c->dec_c.back(), c->dec_c.back(), cin0, c->Cp0, &c->head, MODE_ROWSPLIT, &hb, 0,
                                  col7s_ok ? &c->head_w_swz : nullptr, 96)))
                return rc;
            if ((rc = upload(c, &c->head_bias, hb))) return rc;
            c->head_nc = nc;
        } else if ((rc = build_layer(c, parts, 7, c->dec_c.back(), c->dec_c.back(), cin0, c->Cp0, &c->head))) {
            return rc;
        }
    }
    // generator encoder
    if ((rc = build_layer(c, {{"first.conv", nm("first.norm")}}, 7, c->Cpl, c->Csrc, 0, 0, &c->first))) return rc;
    if (c->first7 && c->Cpl == 3 && first7_supported(c->down_c[0])) {   // K = 147 (196 with the zero rows) instead of 49 x 32
        const HostTensor *wt = find(c, "first.conv.weight"), *bt = find(c, "first.conv.bias"), *gm = find(c, "first.norm.weight"),
                         *be = find(c, "first.norm.bias"), *mu = find(c, "first.norm.running_mean"), *vr = find(c, "first.norm.running_var");
        const int co = c->down_c[0];
        if (wt && bt && gm && be && mu && vr && wt->shape.size() == 4 && wt->shape[0] == co && wt->shape[1] == 3 &&
            wt->shape[2] == 7 && wt->shape[3] == 7) {
            std::vector<float> wf(wt->data), bf(co), packed((size_t)196 * co);
            for (int o = 0; o < co; ++o) {
                const double sc = (double)gm->data[o] / std::sqrt((double)vr->data[o] + 1e-5);
                bf[o] = (float)(((double)bt->data[o] - (double)mu->data[o]) * sc + (double)be->data[o]);
                for (int i = 0; i < 147; ++i) wf[(size_t)o * 147 + i] = (float)((double)wt->data[(size_t)o * 147 + i] * sc);
            }
            first7_pack_host(wf.data(), co, packed.data());
            if ((rc = upload(c, &c->first7_w, packed))) return rc;
            if ((rc = upload(c, &c->first7_bias, bf))) return rc;
        }
    }
    c->down.resize(c->nd);
    for (int i = 0; i < c->nd; ++i) {
        const std::string p = "down_blocks." + std::to_string(i);
        if ((rc = build_set(c, {{p + ".conv", nm(p + ".norm")}}, c->down_c[i], c->down_c[i], 0, 0, &c->down[i], MODE_PLAIN)))
            return rc;
    }
    // bottleneck: conv1 absorbs norm2 (conv1 -> norm2 -> relu), norm1 becomes the producer's second output
    const int nr = g.num_bottleneck_blocks;
    c->res1.resize(nr);
    c->res2.resize(nr);
    c->pre_s.resize(nr);
    c->pre_t.resize(nr);
    for (int i = 0; i < nr; ++i) {
        const std::string r = "bottleneck.r" + std::to_string(i);
        if ((rc = build_set(c, {{r + ".conv1", nm(r + ".norm2")}}, c->Cb, c->Cb, 0, 0, &c->res1[i], MODE_PLAIN))) return rc;
        if ((rc = build_set(c, {{r + ".conv2", ""}}, c->Cb, c->Cb, 0, 0, &c->res2[i], MODE_PLAIN))) return rc;
        if (c->wino_min_m >= 0 && c->Cb % 64 == 0) {
            c->wres1.resize(nr);
            c->wres2.resize(nr);
            if ((rc = build_wino(c, r + ".conv1", r + ".norm2", c->Cb, &c->wres1[i]))) return rc;
            if ((rc = build_wino(c, r + ".conv2", "", c->Cb, &c->wres2[i]))) return rc;
            if (c->wino_tile == 4) {
                c->w4res1.resize(nr);
                c->w4res2.resize(nr);
                if ((rc = build_wino(c, r + ".conv1", r + ".norm2", c->Cb, &c->w4res1[i], 4))) return rc;
                if ((rc = build_wino(c, r + ".conv2", "", c->Cb, &c->w4res2[i], 4))) return rc;
            }
        }
        const HostTensor *gm = find(c, r + ".norm1.weight"), *bt = find(c, r + ".norm1.bias"),
                         *mu = find(c, r + ".norm1.running_mean"), *vr = find(c, r + ".norm1.running_var");
        if ((int)gm->numel() != c->Cb || (int)bt->numel() != c->Cb || (int)mu->numel() != c->Cb ||
            (int)vr->numel() != c->Cb)
            return fail(c, EAMM_ERR_KEY, "%s.norm1 mis-shaped", r.c_str());
        std::vector<float> s(c->Cb), t(c->Cb);
        for (int ch = 0; ch < c->Cb; ++ch) {
            const double sc = (double)gm->data[ch] / std::sqrt((double)vr->data[ch] + 1e-5);
            s[ch] = (float)sc;
            t[ch] = (float)((double)bt->data[ch] - (double)mu->data[ch] * sc);
        }
        if ((rc = upload(c, &c->pre_s[i], s))) return rc;
        if ((rc = upload(c, &c->pre_t[i], t))) return rc;
    }
    // hourglass encoder levels in F(4x4,3x3) form: they borrow the bottleneck's V / Z workspaces (a frame's share must fit)
    if (c->enc_wino && !c->w4res1.empty()) {
        c->w4enc.resize(c->nb);
        const size_t hw_l = (size_t)c->h * c->w, hwf_l = (size_t)c->hf * c->wf;
        for (int i = 0; i < c->nb; ++i) {
            const int cr = i == 0 ? cin0 : c->enc_c[i - 1], cp = i == 0 ? c->Cp0 : c->enc_c[i - 1], co = c->enc_c[i];
            const size_t tiles = (hw_l >> (2 * i)) / 16;
            if (((c->h >> i) & 3) || ((c->w >> i) & 3) || cp % 64 || (co & 3) || 36 * tiles * cp > 4 * hwf_l * c->Cb ||
                tiles * co > (hwf_l / 16) * c->Cb ||
                (size_t)g.max_frames * tiles < (size_t)c->enc_wino_min_tiles * (144.0 * cp * co > 100e6 ? 4 : 1))
                continue;   // (the last: never used at any call size -- do not pack 36 / 9 x the weights for nothing)
            const std::string p = dm + "hourglass.encoder.down_blocks." + std::to_string(i);
            if ((rc = build_wino4_rect(c, p + ".conv", p + ".norm", cr, cp, &c->w4enc[i]))) return rc;
        }
    }
    if (c->enc_wino && !c->w4res1.empty()) {   // the source encoder's down blocks likewise (workspaces sized for them below)
        c->w4down.resize(c->nd);
        for (int i = 0; i < c->nd; ++i) {
            const int ci = c->down_c[i], co = c->down_c[i + 1];
            if (((c->H >> i) & 3) || ((c->W >> i) & 3) || ci % 64 || (co & 3)) continue;
            const std::string p = "down_blocks." + std::to_string(i);
            if ((rc = build_wino4_rect(c, p + ".conv", p + ".norm", ci, ci, &c->w4down[i]))) return rc;
        }
    }
    c->up.resize(c->nd);
    for (int i = 0; i < c->nd; ++i) {
        const std::string p = "up_blocks." + std::to_string(i);
        const int ci = i == 0 ? c->Cb : c->up_c[i - 1];
        if ((rc = build_set(c, {{p + ".conv", nm(p + ".norm")}}, ci, ci, 0, 0, &c->up[i], MODE_PHASE))) return rc;
    }
    if (c->Cimg > 3) {   // more than three outputs: the plain 7x7 kernel (N padded to its 32-column tile), sigmoid and the NCHW
        // store in its epilogue -- correct for any channel count, ten times the minimal work: RGBA is not the shipped case
        if ((rc = build_layer(c, {{"final", ""}}, 7, c->up_c.back(), c->up_c.back(), 0, 0, &c->final_conv))) return rc;
    } else {   // final 7x7 (Cout = 3): 7x1 MFMA convolution over (dx, co) + horizontal gather
        std::vector<float> fb;
        const bool col7_ok = c->col7 && c->up_c.back() % 32 == 0 && c->up_c.back() <= 64;
        if ((rc = build_layer(c, {{"final", ""}}, 7, c->up_c.back(), c->up_c.back(), 0, 0, &c->final_conv,
                              MODE_ROWSPLIT, &fb, 0, col7_ok ? &c->final_w_swz : nullptr)))
            return rc;
        if (c->final_conv.Cout != 21) return fail(c, EAMM_ERR_KEY, "final.weight must have 3 output channels");
        if ((rc = upload(c, &c->final_bias, fb))) return rc;
    }
    // anti-alias buffer [planes,1,13,13]
    {
        std::vector<float> aa((size_t)c->Cpl * 169, 0.f);
        if (g.dm_inv_scale != 1) {
            const HostTensor* t = find(c, dm + "down.weight");
            if (!t || (int)t->numel() != c->Cpl * 169) return fail(c, EAMM_ERR_KEY, "%sdown.weight must be [%d,1,13,13]", dm.c_str(), c->Cimg);
            aa = t->data;
        }
        if ((rc = upload(c, &c->aa_w, aa))) return rc;
    }

    // ---- workspace --------------------------------------------------------------------------
    const size_t S = g.max_sources, F = g.max_frames;
    const size_t HW = (size_t)c->H * c->W, hw = (size_t)c->h * c->w, hwf = (size_t)c->hf * c->wf;
    if ((rc = dev_alloc(c, &c->feat, S * hwf * c->Cb))) return rc;
    if ((rc = dev_alloc(c, &c->src_small, (size_t)c->G * S * hw * 4))) return rc;     // [group][source][h][w][4]
    if ((rc = dev_alloc(c, &c->src_full, S * c->Cpl * HW))) return rc;
    if (c->Cimg != 3)    // the channels a source does not have stay zero for the handle's life (eamm_encode_source copies only the real ones)
        HIP_TRY(c, hipMemset(c->src_full, 0, S * c->Cpl * HW * sizeof(float)));
    if (c->Cimg < 3) {   // one or two channels: the three-channel kernels write staging tensors (four and more write the caller's)
        if ((rc = dev_alloc(c, &c->stage_pred, F * 3 * HW))) return rc;
        if (c->nb > 0 && (rc = dev_alloc(c, &c->stage_deformed, F * 3 * HW))) return rc;
        if (c->nb > 0 && (rc = dev_alloc(c, &c->stage_sparse, F * (c->K + 1) * 3 * hw))) return rc;
    }
    if ((rc = dev_alloc(c, &c->src_nhwc, S * HW * c->Csrc))) return rc;
    c->enc_tmp.resize(c->nd);
    for (int i = 0; i < c->nd; ++i)  // enc_tmp[0] = first output @HxW; enc_tmp[i] = down[i-1] output
        if ((rc = dev_alloc(c, &c->enc_tmp[i], S * (HW >> (2 * i)) * c->down_c[i]))) return rc;
    if ((rc = dev_alloc(c, &c->kp_rec, F * c->K * KP_STRIDE))) return rc;
    if ((rc = dev_alloc(c, &c->bad_flag, 1))) return rc;
    HIP_TRY(c, hipMemset(c->bad_flag, 0, sizeof(int)));
    if (c->nb > 0 && (rc = dev_alloc(c, &c->hg_in, F * hw * c->Cp0))) return rc;
    c->e_buf.resize(c->nb);
    c->u_buf.resize(c->nb);
    for (int i = 0; i < c->nb; ++i) {
        if ((rc = dev_alloc(c, &c->e_buf[i], F * (hw >> (2 * (i + 1))) * c->enc_c[i]))) return rc;
        if ((rc = dev_alloc(c, &c->u_buf[i], F * (hw >> (2 * (c->nb - 1 - i))) * c->dec_c[i]))) return rc;
    }
    if (c->nb > 0 && (rc = dev_alloc(c, &c->logits, F * hw * (c->head_nc ? 128 : 32)))) return rc;
    if (c->nb > 0 && (rc = dev_alloc(c, &c->deformation, F * hw * 2))) return rc;
    if (c->nb > 0 && (rc = dev_alloc(c, &c->occlusion, F * hw))) return rc;
    if ((rc = dev_alloc(c, &c->xa, F * hwf * c->Cb))) return rc;
    if ((rc = dev_alloc(c, &c->xb, F * hwf * c->Cb))) return rc;
    if ((rc = dev_alloc(c, &c->act, F * hwf * c->Cb))) return rc;
    if ((rc = dev_alloc(c, &c->tmp, F * hwf * c->Cb))) return rc;
    if ((rc = dev_alloc(c, &c->final_part, F * HW * 32))) return rc;
    size_t v_elems = 4 * F * hwf * c->Cb, z_elems = 24 * ((F * hwf + 15) / 16) * c->Cb;
    for (size_t i = 0; i < c->w4down.size(); ++i)   // the encoder's down blocks in F(4x4) form borrow the same workspaces
        if (c->w4down[i].Cout) {   // ... for as many sources per call as the 32-bit descriptors allow (more take the direct kernels)
            size_t ns_fit = S;
            while (ns_fit > 0 && !wino4_fits(ns_fit * (HW >> (2 * i)) / 16, c->w4down[i].Cin, c->w4down[i].Cout)) --ns_fit;
            const size_t tiles = ns_fit * (HW >> (2 * i)) / 16;
            v_elems = std::max(v_elems, 36 * tiles * c->w4down[i].Cin);
            z_elems = std::max(z_elems, 24 * tiles * c->w4down[i].Cout);
        }
    if (!c->wres1.empty() && (rc = dev_alloc(c, &c->wino_v, v_elems))) return rc;
    {   // the half-row split of one- / few-frame calls (wino4_groups() == 12) leaves 48 planes of partial x folds instead of 24
        const size_t tiles12 = std::min<size_t>((size_t)(c->cus / 12 / std::max(1, (c->Cb + 63) / 64)) * 64, (F * hwf + 15) / 16);
        z_elems = std::max(z_elems, 48 * tiles12 * c->Cb);
    }
    c->wino_z_elems = z_elems;
    if (!c->w4res1.empty() && (rc = dev_alloc(c, &c->wino_z, z_elems))) return rc;
    if (c->epi_v && !c->w4res1.empty() && (rc = dev_alloc(c, &c->epi_scratch, (size_t)3 * c->cfg.max_frames * c->hf * c->wf * c->Cb))) return rc;
    c->up_buf.resize(c->nd);
    for (int i = 0; i < c->nd; ++i)
        if ((rc = dev_alloc(c, &c->up_buf[i], F * (hwf << (2 * (i + 1))) * c->up_c[i]))) return rc;
    if (tr) {   // pre-BatchNorm convolution outputs (DownBlock2d: at the un-pooled size) and the statistics of the site in flight
        size_t raw_elems = F * HW * c->down_c[0];
        int cmax = std::max(c->down_c[0], c->Cb);
        for (int i = 0; i < c->nd; ++i) {
            raw_elems = std::max(raw_elems, F * (HW >> (2 * i)) * c->down_c[i + 1]);
            raw_elems = std::max(raw_elems, F * (hwf << (2 * (i + 1))) * c->up_c[i]);
            cmax = std::max({cmax, c->down_c[i + 1], c->up_c[i]});
        }
        for (int i = 0; i < c->nb; ++i) {
            raw_elems = std::max(raw_elems, F * (hw >> (2 * i)) * c->enc_c[i]);
            raw_elems = std::max(raw_elems, F * (hw >> (2 * (c->nb - 1 - i))) * c->dec_c[i]);
            cmax = std::max({cmax, c->enc_c[i], c->dec_c[i]});
        }
        if (cmax > 1024 || (cmax & 3)) return fail(c, EAMM_ERR_ARG, "training mode: BatchNorm sites of up to 1024 channels (got %d)", cmax);
        if (S < F) return fail(c, EAMM_ERR_ARG, "training mode needs max_sources >= max_frames (one source per frame of the batch)");
        c->train_cmax = cmax;
        if ((rc = dev_alloc(c, &c->raw, raw_elems))) return rc;
        if ((rc = dev_alloc(c, &c->train_stat, (size_t)2 * cmax))) return rc;
        if ((rc = dev_alloc(c, &c->bn_work, (size_t)cmax * 1024 * 4))) return rc;
    }
    // split-K slab: the largest any layer asks for at any batch size up to the maximum (a smaller batch
    // can pick more K slices than the full one); conv_launch also clamps its slice count to the slab.
    {
        size_t need = 0;
        auto upd1 = [&](const ConvLayer& L, size_t M) { need = std::max(need, conv_plan(L, (int)M).partial_elems); };
        auto upd = [&](const LayerSet& S, size_t M) {
            upd1(S.base, M);
            if (S.has_dma) upd1(S.dma, M);
            if (S.has_big) upd1(S.big, M);
            if (S.has_skinny) {
                upd1(S.skinny32, M);
                upd1(S.skinny64, M);
            }
        };
        for (size_t f = 1; f <= S; ++f) {
            upd1(c->first, f * HW);
            for (int i = 0; i < c->nd; ++i) upd(c->down[i], f * (HW >> (2 * i)));
        }
        const int cus = c->cus;
        auto upd_poly = [&](const LayerSet& S, size_t f, int Hin, int Win) {   // split polyphase launches: raw slabs of the output
            if (!S.has_patch || !S.patch.w_poly || Hin < 16 || Win < 16) return;
            const int sp = patch_poly_splits(S.patch, (int)f, Hin, Win, c->patch_split_max, cus);
            if (sp > 1) need = std::max(need, (size_t)sp * f * 4 * Hin * Win * S.patch.Cout);
        };
        for (size_t f = 1; f <= F; ++f) {
            for (int i = 0; i < c->nb; ++i) {
                upd(c->hg_enc[i], f * (hw >> (2 * i)));
                upd(c->hg_dec[i], f * (hw >> (2 * (c->nb - i))));
                upd_poly(c->hg_dec[i], f, c->h >> (c->nb - i), c->w >> (c->nb - i));
            }
            for (int i = 0; i < c->nd; ++i) upd_poly(c->up[i], f, c->hf << i, c->wf << i);
            if (c->nb > 0) upd1(c->head, f * hw);
            upd(c->res1[0], f * hwf);
            for (int i = 0; i < c->nd; ++i) upd(c->up[i], f * (hwf << (2 * i)));
            upd1(c->final_conv, f * HW);
        }
        c->partial_elems = need;   // one slab per whole-pass chain
        const int slabs = c->pass_chains == 0 ? (g.max_frames >= 64 ? 4 : 2) : c->pass_chains;   // (clamped to 0..4 in eamm_create; 0 = automatic: two chains, four from 64 frames per call)
        if ((rc = dev_alloc(c, &c->partial, c->partial_elems * (size_t)slabs))) return rc;
    }

    // ---- algorithmic FLOPs (reference layer shapes, real channel counts)
    {
        const int cin0_r = (c->K + 1) * (c->Cimg + 1);
        double fe = conv_flops(7, c->Cimg, c->down_r[0], (double)HW);
        for (int i = 0; i < c->nd; ++i) fe += conv_flops(3, c->down_r[i], c->down_r[i + 1], (double)(HW >> (2 * i)));
        c->flops_encode = fe;
        double ff = 0;
        for (int i = 0; i < c->nb; ++i) {
            ff += conv_flops(3, i == 0 ? cin0_r : c->enc_r[i - 1], c->enc_r[i], (double)(hw >> (2 * i)));
            const int ci = i == 0 ? c->enc_r[c->nb - 1] : c->dec_r[i - 1] + c->enc_r[c->nb - 1 - i];
            ff += conv_flops(3, ci, c->dec_r[i], (double)(hw >> (2 * (c->nb - 1 - i))));
        }
        if (c->nb > 0) ff += conv_flops(7, c->dec_r.back() + cin0_r, c->K + 1 + (g.estimate_occlusion_map ? 1 : 0), (double)hw);
        ff += 2.0 * nr * conv_flops(3, c->Cb_r, c->Cb_r, (double)hwf);
        for (int i = 0; i < c->nd; ++i)
            ff += conv_flops(3, i == 0 ? c->Cb_r : c->up_r[i - 1], c->up_r[i], (double)(hwf << (2 * (i + 1))));
        ff += conv_flops(7, c->up_r.back(), c->Cimg, (double)HW);
        if (c->nb > 0) ff += 9.0 * hwf * c->Cb_r;  // bilinear feature warp + occlusion multiply
        c->flops_frame = ff;
    }
    const int max_chains = std::max(c->bneck_chains, c->pass_chains == 0 ? (g.max_frames >= 64 ? 4 : 2) : c->pass_chains);
    c->private_streams = env_int("EAMM_PRIVATE_STREAMS", 0);
    if (max_chains > 1) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_warp, hipEventDisableTiming));
        for (int k = 1; k < max_chains; ++k) {
            hipStream_t st = chain_stream(c->device, k - 1);
            hipEvent_t ev = nullptr;
            if (!st) return fail(c, EAMM_ERR_HIP, "hipStreamCreate failed for chain %d", k);
            c->pool_streams.push_back(st);
            HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->ev_join.push_back(ev);
            // the private twin (StreamLease): created here, outside any graph capture; idle unless a call needs it
            hipStream_t own = nullptr;
            HIP_TRY(c, hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
            c->own_streams.push_back(own);
        }
    }
    c->side_streams = c->pool_streams;   // (plan queries count them; every call picks its set: StreamLease)
    c->sd.clear();
    HIP_TRY(c, hipDeviceSynchronize());
    c->finalized = true;
    return EAMM_OK;
}

int eamm_encode_source(eamm_ctx* c, const float* source, int ns, void* stream_) {
    if (!c || !source) return fail(c, EAMM_ERR_ARG, "null argument");
    if (!c->finalized) return fail(c, EAMM_ERR_STATE, "call eamm_finalize_weights first");
    if (ns < 1 || ns > c->cfg.max_sources) return fail(c, EAMM_ERR_ARG, "ns=%d outside [1,%d]", ns, c->cfg.max_sources);
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const size_t HW = (size_t)c->H * c->W;
    if (c->Cimg == 3) {
        HIP_TRY(c, hipMemcpyAsync(c->src_full, source, (size_t)ns * 3 * HW * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {   // [ns,C,H,W] into the first C planes of [ns,3,H,W]; the kernels below read the zero-extended copy
        HIP_TRY(c, hipMemcpy2DAsync(c->src_full, c->Cpl * HW * sizeof(float), source, c->Cimg * HW * sizeof(float), c->Cimg * HW * sizeof(float),
                                    (size_t)ns, hipMemcpyDeviceToDevice, s));
        source = c->src_full;
    }
    if (c->G == 1) {
        HIP_TRY(c, source_prepare_launch(source, c->aa_w, ns, c->H, c->W, c->cfg.dm_inv_scale, c->Csrc, c->src_nhwc,
                                         c->src_small, s));
    } else {   // two groups of three planes: the padded NHWC copy for `first`, one down-sampled float4 image per group
        HIP_TRY(c, nchw_to_nhwc_pad_launch(source, ns, c->Cpl, c->H, c->W, c->Csrc, c->src_nhwc, s));
        for (int g = 0; g < c->G; ++g)
            HIP_TRY(c, antialias_down_launch(source, c->aa_w, ns, c->H, c->W, c->cfg.dm_inv_scale, 4,
                                             c->src_small + (size_t)g * c->cfg.max_sources * c->h * c->w * 4, s, c->Cpl, 3 * g));
    }
    ConvIO io{};
    io.in0 = c->src_nhwc;
    io.B = ns;
    io.Hin = c->H;
    io.Win = c->W;
    io.act = ACT_RELU;
    io.out = c->enc_tmp[0];
    io.partial = c->partial;
        io.partial_cap = c->partial_elems;
    if (c->first7_w)                                              // SameBlock2d 7x7   generator.py:61
        HIP_TRY(c, first7_launch(source, c->first7_w, c->first7_bias, ns, c->H, c->W, c->down_c[0], c->enc_tmp[0], s));
    else
        HIP_TRY(c, conv_launch(c->first, io, s));
    for (int i = 0; i < c->nd; ++i) {                             // DownBlock2d       generator.py:62-63
        ConvIO d{};
        d.in0 = c->enc_tmp[i];
        d.B = ns;
        d.Hin = c->H >> i;
        d.Win = c->W >> i;
        d.act = ACT_RELU;
        d.pool = 1;
        d.out = (i == c->nd - 1) ? c->feat : c->enc_tmp[i + 1];
        d.partial = c->partial;
        d.partial_cap = c->partial_elems;
        const int tiles = ns * (d.Hin / 4) * (d.Win / 4);
        if (i < (int)c->w4down.size() && c->w4down[i].Cout && c->wino_v && c->wino_z && tiles >= c->enc_wino_min_tiles &&
            wino4_fits((size_t)tiles, c->w4down[i].Cin, c->w4down[i].Cout) &&
            288e-6 * tiles * c->w4down[i].Cin * c->w4down[i].Cout >= (double)c->enc_wino_min_mflop) {
            const WinoLayer& L = c->w4down[i];   // as the hourglass encoder levels (forward_view)
            const int nblk = ((tiles + 63) / 64) * L.ntiles;
            int cus = c->cus, g = 2;
            for (int gsel : {6, 3})
                if (nblk * gsel <= cus) {
                    g = gsel;
                    break;
                }
            HIP_TRY(c, wino4_transform_launch(d.in0, nullptr, nullptr, ns, d.Hin, d.Win, L.Cin, c->wino_v, s));
            HIP_TRY(c, wino4_gemm_launch(L, c->wino_v, ns, d.Hin, d.Win, ACT_RELU, nullptr, d.out, s, 0, g, c->wino_z, 1));
            continue;
        }
        HIP_TRY(c, conv_launch(pick(c, c->down[i], (size_t)ns * d.Hin * d.Win), d, s));
    }
    c->ns_cached = ns;
    return EAMM_OK;
}

// 0 = direct, 2 = Winograd F(2x2,3x3), 4 = Winograd F(4x4,3x3) for a call of n frames
static int bottleneck_form(const eamm_ctx* c, int n) {
    const int hf = c->hf, wf = c->wf;
    if (c->wres1.empty()) return 0;
    const size_t px = (size_t)n * hf * wf;
    if (!c->w4res1.empty() && hf % 4 == 0 && wf % 4 == 0 && px >= (size_t)c->wino4_min_m) return 4;
    // F(2x2) tiles are the 2x2 pixel quads: odd map sides (a configuration eamm_create accepts) take the direct form
    return (px >= (size_t)c->wino_min_m && !(hf & 1) && !(wf & 1)) ? 2 : 0;
}

// F(4x4) with few tiles: split the six rows of transform points over 2, 3 or 6 workgroups per (tile, cout) block --
// the largest split that still fits one round of the chip (64 x 64 blocks, one per CU)
static int wino4_groups(const eamm_ctx* c, int n) {
    const int nb = ((n * (c->hf / 4) * (c->wf / 4) + 63) / 64) * ((c->Cb + 63) / 64);
    if (c->wino4_groups_knob == 12) {   // the half-row split needs Cb % 128 == 0 and 48 planes of workspace: else the rule below
        if (c->Cb % (4 * CONV_BK) == 0 && (size_t)48 * n * (c->hf / 4) * (c->wf / 4) * c->Cb <= c->wino_z_elems) return 12;
    } else if (c->wino4_groups_knob) {
        return c->wino4_groups_knob;
    }
    // round 6: 12 = HALF rows of transform points.  One 256x256 frame is 16 blocks of 64 x 64: split 6 gave 96 workgroups (run as 192
    // narrow 32-tile four-wave blocks, one wave per SIMD, 21.5 us per launch); split 12 gives 192 workgroups of the eight-wave kernel
    // (two waves per SIMD, twice the flops per operand byte) at the price of 48 instead of 24 planes of x-folded products
    // MEASURED (profiles/r06_experiments.txt): 21.3 us per launch against the narrow plan's 21.5 and an output transform of 6.6 instead of 5.4 us --
    // one frame 0.808 vs 0.811 ms: no gain, so the rule keeps 6; 12 stays a tested option (EAMM_WINO4_GROUPS=12, eamm_op_conv tile 2162)
    for (int gsel : {6, 3, 2})
        if (nb * gsel <= c->cus) return gsel;
    return 1;
}

// GEMM variant 6 (= 3 with the V stream loaded non-temporally) pays only while all the GEMM workgroups of a call are co-resident,
// one per CU: the four cout workgroups of a tile row then pull a V line through the L2 together.  16 frames at 256x256 (two chains
// of 128 workgroups): + 0.3 %; 128 frames (four chains of 512): - 2 % (profiles/r05_experiments.txt 13) -- those keep variant 3.
static int wino4_variant_for(const eamm_ctx* c, int call_frames) {
    if (c->wino4_variant != 6 || c->wino4_variant_pinned) return c->wino4_variant;
    const size_t wgs = (((size_t)call_frames * (c->hf / 4) * (c->wf / 4) + 63) / 64) * ((c->Cb + 63) / 64);
    return wgs <= (size_t)c->cus ? 6 : 3;
}

// Chains: the frames of a call are independent, so the F(4x4) bottleneck can run as K groups of frames on K streams.
// Each chain alternates an HBM-bound input transform with an MFMA-bound GEMM that fills 1/K of the chip; while one
// chain's transform streams through HBM the other chains' GEMMs own the matrix pipes, and a chain's transform moves only
// its share of the data -- measured at 16 frames, 256x256: bottleneck 2.74 -> 2.50 ms per step with K = 2 (K = 3, 4 lose
// it again to launch / queue overheads).  Every chain gets whole frames and a whole number of 64-tile GEMM blocks.
static int bottleneck_chains(const eamm_ctx* c, int n) {
    if (bottleneck_form(c, n) != 4 || c->bneck_chains < 2 || c->side_streams.empty() || wino4_groups(c, n) != 1) return 1;
    int chains = std::min<int>(c->bneck_chains, (int)c->side_streams.size() + 1);
    const int tiles_pf = (c->hf / 4) * (c->wf / 4);
    while (chains > 1 && (n < chains || ((n / chains) * tiles_pf) % 64 != 0 || ((n / chains + 1) * tiles_pf) % 64 != 0)) --chains;
    return chains;
}

// The slice of the per-frame workspace one launch sequence works on: frames [f0, f0 + n) of a call (every per-frame
// buffer is frame-major, so a slice is a pointer offset) with its own split-K slab and Winograd buffers.
struct FrameView {
    int n = 0, ns = 1;
    const float *kd_val = nullptr, *kd_jac = nullptr, *ks_val = nullptr, *ks_jac = nullptr;
    const float *feat = nullptr, *src_small = nullptr, *src_full = nullptr;
    float *kp_rec = nullptr, *hg_in = nullptr, *logits = nullptr, *deformation = nullptr, *occlusion = nullptr;
    float *xa = nullptr, *xb = nullptr, *act = nullptr, *tmp = nullptr, *final_part = nullptr;
    std::vector<float*> e_buf, u_buf, up_buf;
    float* partial = nullptr;
    size_t partial_elems = 0;
    float *wino_v = nullptr, *wino_z = nullptr;
    eamm_outputs out{};
};

static FrameView make_view(const eamm_ctx* c, int f0, int n, int ns_call, int slab, const float* kd_val, const float* kd_jac,
                           const float* ks_val, const float* ks_jac, const eamm_outputs* o) {
    FrameView v;
    const size_t F = (size_t)f0, K = c->K;
    const size_t HW = (size_t)c->H * c->W, hw = (size_t)c->h * c->w, hwf = (size_t)c->hf * c->wf;
    const size_t sf = ns_call > 1 ? F : 0;   // per-frame sources (the module's batch contract) move with the frames
    v.n = n;
    v.ns = ns_call > 1 ? n : 1;
    v.kd_val = kd_val + F * K * 2;
    v.kd_jac = kd_jac ? kd_jac + F * K * 4 : nullptr;
    v.ks_val = ks_val + sf * K * 2;
    v.ks_jac = ks_jac ? ks_jac + sf * K * 4 : nullptr;
    v.feat = c->feat + sf * hwf * c->Cb;
    v.src_small = c->src_small + sf * hw * 4;
    v.src_full = c->src_full + sf * c->Cpl * HW;
    v.kp_rec = c->kp_rec + F * K * KP_STRIDE;
    v.hg_in = c->hg_in ? c->hg_in + F * hw * c->Cp0 : nullptr;            // (no motion network: these four do not exist)
    for (int i = 0; i < c->nb; ++i) {
        v.e_buf.push_back(c->e_buf[i] + F * (hw >> (2 * (i + 1))) * c->enc_c[i]);
        v.u_buf.push_back(c->u_buf[i] + F * (hw >> (2 * (c->nb - 1 - i))) * c->dec_c[i]);
    }
    v.logits = c->logits ? c->logits + F * hw * (c->head_nc ? 128 : 32) : nullptr;
    v.deformation = c->deformation ? c->deformation + F * hw * 2 : nullptr;
    v.occlusion = c->occlusion ? c->occlusion + F * hw : nullptr;
    v.xa = c->xa + F * hwf * c->Cb;
    v.xb = c->xb + F * hwf * c->Cb;
    v.act = c->act + F * hwf * c->Cb;
    v.tmp = c->tmp + F * hwf * c->Cb;
    for (int i = 0; i < c->nd; ++i) v.up_buf.push_back(c->up_buf[i] + F * (hwf << (2 * (i + 1))) * c->up_c[i]);
    v.final_part = c->final_part + F * HW * 32;
    v.partial = c->partial + (size_t)slab * c->partial_elems;
    v.partial_elems = c->partial_elems;
    v.wino_v = c->wino_v ? c->wino_v + 4 * F * hwf * c->Cb : nullptr;                  // sized 4 x activations per frame
    v.wino_z = c->wino_z ? c->wino_z + 24 * (F * hwf / 16) * c->Cb : nullptr;
    v.out = *o;
    const size_t CO = c->Cimg > 3 ? c->Cimg : 3;   // channels of the tensors the pass writes (staging tensors for one or two)
    if (o->prediction) v.out.prediction = o->prediction + F * CO * HW;
    if (o->mask) v.out.mask = o->mask + F * (K + 1) * hw;
    if (o->sparse_deformed) v.out.sparse_deformed = o->sparse_deformed + F * (K + 1) * CO * hw;
    if (o->occlusion_map) v.out.occlusion_map = o->occlusion_map + F * hw;
    if (o->deformed) v.out.deformed = o->deformed + F * CO * HW;
    if (o->deformation) v.out.deformation = o->deformation + F * hw * 2;
    if (o->frames_u8) v.out.frames_u8 = o->frames_u8 + F * HW * 3;
    return v;
}

// Chains over the WHOLE per-frame pass: the frames of a call are independent, so a call of n frames can run as K launch
// sequences of n/K frames on K streams.  The sequences drift apart (the host enqueues them one after the other), and the
// chip then runs one chain's HBM- or latency-bound kernels (input transforms, warps, the small-grid deep hourglass levels
// and their split-K reductions) beside the other chain's MFMA-bound ones.  Needs the F(4x4) bottleneck with whole 64-tile
// GEMM blocks per chain; EAMM_PASS_CHAINS (0 / 1 = off).
static int pass_chains(const eamm_ctx* c, int n) {
    const bool automatic = c->pass_chains == 0;
    const int tiles_pf = (c->hf / 4) * (c->wf / 4);
    auto fits = [&](int K) {
        if (K < 2 || (int)c->side_streams.size() + 1 < K) return false;
        if (n < c->pass_chains_min_frames || n < K) return false;
        if (automatic) {   // only while a chain's bottleneck GEMM keeps >= 80 one-per-CU blocks (measured 256x256, frames/s off -> on:
            // 8 frames = 64 blocks per chain 2981 -> 2397; 10: 2695 -> 2794; 12: 3018 -> 3105; 16: 3720 -> 3790; 24: 3716 -> 3864;
            // 32: 3871 -> 3980; 512x512 x 4: 941 -> 952, x 8: 985 -> 1006)
            if (((n / K) * tiles_pf / 64) * ((c->Cb + 63) / 64) < c->pass_chains_min_blocks) return false;
        }
        for (int k = 0; k < 2; ++k) {   // both chain sizes (n/K and n/K + 1 when n % K != 0)
            const int nk = n / K + k;
            if (k == 1 && n % K == 0) break;
            if (bottleneck_form(c, nk) != 4 || (nk * tiles_pf) % 64 != 0) return false;
        }
        return true;
    };
    if (!automatic) return fits(c->pass_chains) ? c->pass_chains : 1;
    // automatic: two chains; four from 64 frames per call, where every chain still gets 16 (256x256: 4150 -> 4197 frames/s,
    // profiles/r04_experiments.txt section 16; four chains of 8 at 32 frames gain nothing over two of 16)
    if (n >= 64 && fits(4)) return 4;
    return fits(2) ? 2 : 1;
}

// One launch sequence over the frames of `v` on stream s.  `chained`: another sequence runs beside this one.
// phase: the whole pass, or its two halves around the feature warp (PH_PRE: key points ... flow head; PH_POST: warp ... final
// layer).  With `joint` (first chain's PH_POST) the warps run ONCE for all the call's frames -- the view `joint` -- and the
// other chains' PH_POST skip them and start behind c->ev_warp.
enum { PH_ALL = 0, PH_PRE = 1, PH_POST = 2 };
static int forward_view(eamm_ctx* c, const FrameView& v, hipStream_t s, hipEvent_t* ev, bool chained, hipEvent_t* cev = nullptr,
                        int chain_idx = 0, int phase = PH_ALL, const FrameView* joint = nullptr) {
    const int n = v.n, ns = v.ns;
    const int h = c->h, w = c->w, hf = c->hf, wf = c->wf, K = c->K;
    const bool occ = c->cfg.estimate_occlusion_map != 0;
    // executed matrix-core flops launched since the last mark belong to the stage interval that ends here
    auto account = [&](int interval) {
        const double f = take_mfma_flops();
        c->call_flops += f;
        c->call_stage_flops[interval] += f;
        if (interval == 5) c->call_flops_bneck += f;
    };
#define STAGE_MARK(i)                                  \
    do {                                               \
        if ((i) > 0) account((i) - 1);                 \
        if (ev) {                                      \
            HIP_TRY(c, hipEventRecord(ev[i], s));      \
            c->prof_marks.back() = (i) + 1;            \
        }                                              \
    } while (0)
    const int nr = c->cfg.num_bottleneck_blocks;
    const int form = bottleneck_form(c, n);
    const bool wino = form != 0;
    if (c->nb == 0) {   // no motion network (generator.py:64 is false): the encoder features enter the bottleneck as they are
        STAGE_MARK(0); STAGE_MARK(1); STAGE_MARK(2); STAGE_MARK(3); STAGE_MARK(4);
        HIP_TRY(c, broadcast_features_launch(v.feat, n, ns, hf, wf, c->Cb, v.xa, wino ? nullptr : v.act, c->pre_s[0], c->pre_t[0], s));
        STAGE_MARK(5);
    } else {
    if (phase != PH_POST) {
    STAGE_MARK(0);
    // key-point records; 'jacobian' missing from kp_driving => identity (dense_motion.py:55)
    HIP_TRY(c, kp_prepare_launch(v.kd_val, v.kd_jac, v.ks_val, v.kd_jac ? v.ks_jac : nullptr, n, ns, K, v.kp_rec, c->bad_flag, s));
    // heat-maps + sparse motions + K+1 warped sources -> hourglass input     dense_motion.py:88-94
    HIP_TRY(c, motion_front_launch(v.kp_rec, v.src_small, n, ns, K, h, w, c->cfg.kp_variance, c->Cp0, v.hg_in,
                                   v.out.sparse_deformed, s, c->G, c->Cimg > 3 ? c->Cimg : 3, (size_t)c->cfg.max_sources * h * w * 4));
    STAGE_MARK(1);
    // hourglass encoder                                                       util.py:956-960
    for (int i = 0; i < c->nb; ++i) {
        ConvIO io{};
        io.in0 = i == 0 ? v.hg_in : v.e_buf[i - 1];
        io.B = n;
        io.Hin = h >> i;
        io.Win = w >> i;
        io.act = ACT_RELU;
        io.pool = 1;
        io.out = v.e_buf[i];
        io.partial = v.partial;
        io.partial_cap = v.partial_elems;
        const int tiles = n * (io.Hin / 4) * (io.Win / 4);
        // (levels whose transformed weights exceed ~100 MB -- the 4x4-map level: 151 MB -- need four times the tiles to pay)
        const bool heavy = i < (int)c->w4enc.size() && 144.0 * c->w4enc[i].Cin * c->w4enc[i].Cout > 100e6;
        if (i < (int)c->w4enc.size() && c->w4enc[i].Cout && tiles >= c->enc_wino_min_tiles * (heavy ? 4 : 1) &&
            288e-6 * tiles * c->w4enc[i].Cin * c->w4enc[i].Cout >= (double)c->enc_wino_min_mflop) {   // 2 * 9 * 16 pixels per tile
            // F(4x4,3x3): 4x fewer MACs; the transform-point rows are split over workgroups (>= 2: the pooled epilogue is
            // the output-transform kernel's) -- the largest split that still fits one round of the chip
            const WinoLayer& L = c->w4enc[i];
            const int nblk = ((tiles + 63) / 64) * L.ntiles;
            int cus = c->cus, g = 2;
            cus = cus * c->enc_cus_pct / 100;   // EAMM_ENC_CUS_PCT (experiment): the CU count the split of the point rows is sized for
            for (int gsel : {6, 3})
                if (nblk * gsel <= cus) {
                    g = gsel;
                    break;
                }
            HIP_TRY(c, wino4_transform_launch(io.in0, nullptr, nullptr, n, io.Hin, io.Win, L.Cin, v.wino_v, s));
            // pipeline variant 0 (one DMA piece per 4 MFMAs): within noise of the bottleneck's variant at these sizes, and a
            // different instantiation -- per-kernel averages in traces / PMC passes stay those of the bottleneck launches alone
            HIP_TRY(c, wino4_gemm_launch(L, v.wino_v, n, io.Hin, io.Win, ACT_RELU, nullptr, io.out, s, 0, g, v.wino_z, 1));
            continue;
        }
        HIP_TRY(c, conv_launch(pick(c, c->hg_enc[i], (size_t)n * io.Hin * io.Win), io, s));
    }
    STAGE_MARK(2);
    // hourglass decoder: nearest x2 and the skip concatenation are folded into the operand loader
    for (int i = 0; i < c->nb; ++i) {
        ConvIO io{};
        io.in0 = i == 0 ? v.e_buf[c->nb - 1] : v.u_buf[i - 1];
        io.in1 = i == 0 ? nullptr : v.e_buf[c->nb - 1 - i];
        io.B = n;
        io.Hin = h >> (c->nb - i);
        io.Win = w >> (c->nb - i);
        io.act = ACT_RELU;
        io.out = v.u_buf[i];
        io.partial = v.partial;
        io.partial_cap = v.partial_elems;
        if (int urc = launch_up(c, c->hg_dec[i], io, s)) return urc;
    }
    STAGE_MARK(3);
    // mask / occlusion logits, then softmax + flow combine + sigmoid         dense_motion.py:98-111
    {
        ConvIO io{};
        io.in0 = v.u_buf[c->nb - 1];
        io.in1 = v.hg_in;
        io.B = n;
        io.Hin = h;
        io.Win = w;
        io.act = ACT_NONE;
        io.out = v.logits;
        io.partial = v.partial;
        io.partial_cap = v.partial_elems;
        ConvLayer head = c->head;
        float* defo = v.deformation;
        if (c->head_nc) {
            head.Cout = 128;  // partial products written with a 128-float pixel stride (7*nc used)
            // the column-patch kernel runs one workgroup per 16x16-pixel tile: below ~half a chip of tiles (4 frames at
            // 64x64) the split-K im2col kernel is faster (measured: 1 frame 0.122 -> 0.073 ms, 4 frames 0.122 -> 0.094 ms)
            const int head_tiles = n * ((h + 15) / 16) * ((w + 15) / 16);
            if (c->head_w_swz && head_tiles >= c->head_col7_min_tiles)
                HIP_TRY(c, conv_col7s_launch(io.in0, head.C0, io.in1, head.C1, n, h, w, c->head_w_swz, 3, v.logits, 128, s));
            else
                HIP_TRY(c, conv_launch(head, io, s));
            HIP_TRY(c, motion_head_rowsplit_launch(v.logits, 128, c->head_nc, c->head_bias, v.kp_rec, n, K, h, w,
                                                   occ ? 1 : 0, defo, v.occlusion, v.out.mask, v.out.occlusion_map, s));
        } else {
            head.Cout = 32;  // logits are written with a 32-float pixel stride; channels >= K+2 have zero weights
            HIP_TRY(c, conv_launch(head, io, s));
            HIP_TRY(c, motion_head_launch(v.logits, v.kp_rec, n, K, h, w, occ ? 1 : 0, defo, v.occlusion, v.out.mask,
                                          v.out.occlusion_map, s));
        }
        if (v.out.deformation)
            HIP_TRY(c, hipMemcpyAsync(v.out.deformation, defo, (size_t)n * h * w * 2 * sizeof(float),
                                      hipMemcpyDeviceToDevice, s));
    }
    }   // phase != PH_POST
    if (phase == PH_PRE) {
        account(3);
        return EAMM_OK;
    }
    if (phase == PH_POST && joint == nullptr) {
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_warp, 0));    // the first chain's launch warped this chain's frames as well
    } else {
        if (phase == PH_POST)   // every chain's flow is ready (the caller made this stream wait for the others' PH_PRE)
            for (size_t k = 0; k < c->ev_join.size(); ++k) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[k], 0));
        STAGE_MARK(4);
        // feature warp x occlusion (+ r0's pre-activation for the direct form)       generator.py:79-84
        const FrameView& wv = joint ? *joint : v;
        HIP_TRY(c, warp_features_launch(wv.feat, wv.deformation, occ ? wv.occlusion : nullptr, wv.n, wv.ns, hf, wf, c->Cb, h, w,
                                        wv.xa, wino ? nullptr : wv.act, c->pre_s[0], c->pre_t[0], s));
        if (wv.out.deformed)                                                        // generator.py:86
            HIP_TRY(c, warp_image_launch(wv.src_full, wv.deformation, wv.n, wv.ns, c->H, c->W, h, w, wv.out.deformed, s, c->Cpl,
                                         c->Cimg > 3 ? c->Cimg : 3));
        if (joint) HIP_TRY(c, hipEventRecord(c->ev_warp, s));
        STAGE_MARK(5);
    }
    }
    // bottleneck                                                               generator.py:89
    float *x = v.xa, *xn = v.xb;
    if (cev) HIP_TRY(c, hipEventRecord(cev[0], s));
    hipEvent_t* sub = (ev && wino && 4 * nr + 1 <= eamm_ctx::NSUB) ? ev + eamm_ctx::NMARK + 1 : nullptr;
    int nsub = 0;
#define SUB_MARK()                                               \
    do {                                                         \
        if (sub) HIP_TRY(c, hipEventRecord(sub[nsub++], s));     \
    } while (0)
    const bool wino4 = form == 4;
    int w4g_sel = (wino4 && !chained) ? wino4_groups(c, n) : 1;   // a chained view shares the chip: never split the point rows
    if (w4g_sel == 12 && v.wino_z != c->wino_z) w4g_sel = 6;      // (the 48-plane form is sized for a call's first frames only)
    const int w4g = w4g_sel;
    // (round 5: splitting a whole-pass chain's bottleneck once more -- 2 x 2 sub-chains of 4 frames -- was built and measured
    //  neutral, 3917 vs 3958 frames/s: the input transform is latency-bound and does not shrink with the frames per launch;
    //  profiles/r05_experiments.txt section 2, git 9500c6c)
    const int chains = chained ? 1 : bottleneck_chains(c, n);
    const int w4var = wino4_variant_for(c, chained ? c->cur_call_frames : n);
    auto split_stream = [&](int k) { return c->side_streams[k - 1]; };
    hipEvent_t split_fork = c->ev_fork;
    auto split_join = [&](int k) { return c->ev_join[k - 1]; };
    if (chains > 1) {
        const size_t per_frame = (size_t)hf * wf * c->Cb;
        const int nbase = n / chains, nrem = n % chains;   // the first nrem chains take one more frame
        HIP_TRY(c, hipEventRecord(split_fork, s));
        for (int k = 1; k < chains; ++k) HIP_TRY(c, hipStreamWaitEvent(split_stream(k), split_fork, 0));
        for (int i = 0; i < nr; ++i) {
            for (int k = 0; k < chains; ++k) {
                hipStream_t sk = k ? split_stream(k) : s;
                const int nk = nbase + (k < nrem ? 1 : 0);
                const size_t f0 = (size_t)k * nbase + std::min(k, nrem);
                float* xk = x + f0 * per_frame;
                float* xnk = xn + f0 * per_frame;
                float* tk = v.tmp + f0 * per_frame;
                float* vk = v.wino_v + 4 * f0 * per_frame;   // 2.25 nk frames used of the 4 nk reserved for this chain
                // events around the main stream's (chain 0's) kernels only: they time those launches while the other
                // chains' run beside them
                if (k == 0) SUB_MARK();
                HIP_TRY(c, wino4_transform_launch(xk, c->pre_s[i], c->pre_t[i], nk, hf, wf, c->Cb, vk, sk));
                if (k == 0) SUB_MARK();
                HIP_TRY(c, wino4_gemm_launch(c->w4res1[i], vk, nk, hf, wf, ACT_RELU, nullptr, tk, sk, w4var, 1, nullptr));
                if (k == 0) SUB_MARK();
                HIP_TRY(c, wino4_transform_launch(tk, nullptr, nullptr, nk, hf, wf, c->Cb, vk, sk));
                if (k == 0) SUB_MARK();
                HIP_TRY(c, wino4_gemm_launch(c->w4res2[i], vk, nk, hf, wf, ACT_NONE, xk, xnk, sk, w4var, 1, nullptr));
            }
            std::swap(x, xn);
        }
        SUB_MARK();
        for (int k = 1; k < chains; ++k) {
            HIP_TRY(c, hipEventRecord(split_join(k), split_stream(k)));
            HIP_TRY(c, hipStreamWaitEvent(s, split_join(k), 0));
        }
    }
    for (int i = 0; i < nr && wino && chains == 1; ++i) {
        // conv1(relu(norm1(x))): the pre-activation rides on the input transform; norm2 + relu in the epilogue
        SUB_MARK();
        if (wino4) {
            const bool stagger = chained && c->bneck_stagger && c->ev_stagger && i == 0;
            if (stagger && chain_idx > 0) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_stagger, 0));
            // EAMM_WINO4_EPI_V (timing experiment, wrong results): only the stage's FIRST transform runs (every GEMM then reads that
            // V: realistic operand values), and the GEMMs' epilogues write V-sized extra output instead (variant 50)
            const bool epi = c->epi_v && c->epi_scratch && w4g == 1;
            float* escr = epi ? c->epi_scratch + (size_t)3 * (v.xa - c->xa) : v.wino_z;
            const int wvar = epi ? 50 : w4var;
            if (!epi || i == 0) HIP_TRY(c, wino4_transform_launch(x, c->pre_s[i], c->pre_t[i], n, hf, wf, c->Cb, v.wino_v, s));
            if (stagger && chain_idx == 0) HIP_TRY(c, hipEventRecord(c->ev_stagger, s));
            SUB_MARK();
            HIP_TRY(c, wino4_gemm_launch(c->w4res1[i], v.wino_v, n, hf, wf, ACT_RELU, nullptr, v.tmp, s, wvar, w4g, escr));
            SUB_MARK();
            if (!epi) HIP_TRY(c, wino4_transform_launch(v.tmp, nullptr, nullptr, n, hf, wf, c->Cb, v.wino_v, s));
            SUB_MARK();
            HIP_TRY(c, wino4_gemm_launch(c->w4res2[i], v.wino_v, n, hf, wf, ACT_NONE, x, xn, s, wvar, w4g, escr));   // out += x
        } else {
            HIP_TRY(c, wino_transform_launch(x, c->pre_s[i], c->pre_t[i], n, hf, wf, c->Cb, v.wino_v, s));
            SUB_MARK();
            HIP_TRY(c, wino_gemm_launch(c->wres1[i], v.wino_v, n, hf, wf, ACT_RELU, nullptr, v.tmp, s, c->wino_variant));
            SUB_MARK();
            HIP_TRY(c, wino_transform_launch(v.tmp, nullptr, nullptr, n, hf, wf, c->Cb, v.wino_v, s));
            SUB_MARK();
            HIP_TRY(c, wino_gemm_launch(c->wres2[i], v.wino_v, n, hf, wf, ACT_NONE, x, xn, s, c->wino_variant));   // out += x
        }
        std::swap(x, xn);
    }
    if (chains == 1) SUB_MARK();
#undef SUB_MARK
    if (sub) c->prof_sub.back() = nsub;
    for (int i = 0; i < nr && !wino; ++i) {
        ConvIO a{};
        a.in0 = v.act;
        a.B = n;
        a.Hin = hf;
        a.Win = wf;
        a.act = ACT_RELU;  // conv1 -> norm2 (folded) -> relu
        a.out = v.tmp;
        a.partial = v.partial;
        a.partial_cap = v.partial_elems;
        HIP_TRY(c, conv_launch(pick(c, c->res1[i], (size_t)n * hf * wf), a, s));
        ConvIO b{};
        b.in0 = v.tmp;
        b.B = n;
        b.Hin = hf;
        b.Win = wf;
        b.act = ACT_NONE;
        b.resid = x;       // out += x
        b.out = xn;
        if (i + 1 < nr) {  // next block's relu(norm1(.))
            b.out2 = v.act;
            b.s2 = c->pre_s[i + 1];
            b.t2 = c->pre_t[i + 1];
        }
        b.partial = v.partial;
        b.partial_cap = v.partial_elems;
        HIP_TRY(c, conv_launch(pick(c, c->res2[i], (size_t)n * hf * wf), b, s));
        std::swap(x, xn);
    }
    STAGE_MARK(6);
    if (cev) HIP_TRY(c, hipEventRecord(cev[1], s));
    // up blocks                                                                generator.py:90-91
    const float* cur = x;
    for (int i = 0; i < c->nd; ++i) {
        ConvIO io{};
        io.in0 = cur;
        io.B = n;
        io.Hin = hf << i;
        io.Win = wf << i;
        io.act = ACT_RELU;
        io.out = v.up_buf[i];
        io.partial = v.partial;
        io.partial_cap = v.partial_elems;
        if (int urc = launch_up(c, c->up[i], io, s)) return urc;
        cur = v.up_buf[i];
    }
    STAGE_MARK(7);
    // final 7x7 + sigmoid, written NCHW straight into the caller's buffer      generator.py:92-93
    {
        ConvIO io{};
        io.in0 = cur;
        io.B = n;
        io.Hin = c->H;
        io.Win = c->W;
        io.act = ACT_NONE;
        io.out = v.final_part;
        io.partial = v.partial;
        io.partial_cap = v.partial_elems;
        if (c->Cimg > 3) {   // plain 7x7 kernel, sigmoid + NCHW store in the epilogue, straight into the caller's [n,C,H,W]
            io.act = ACT_SIGMOID;
            io.nchw = 1;
            io.out = v.out.prediction;
            HIP_TRY(c, conv_launch(c->final_conv, io, s));
            account(7);
            return EAMM_OK;
        }
        ConvLayer fl = c->final_conv;
        fl.Cout = 32;  // 32-float pixel stride; channels >= 21 have zero weights
        // fused form: one workgroup per ROW of 16x16 tiles -- only when those rows fill at least half the chip (measured
        // 256x256: 16 frames 0.289 -> 0.283 ms per step, 65 MB written + read back per 8 frames gone; ONE frame has 16 tile rows:
        // 0.029 -> 0.262 ms, so small calls keep the two-kernel form)
        if (c->final_w_swz && fl.BN == 32 && c->final_fused && n * ((c->H + 15) / 16) >= c->final_fused_min_rows) {
            // gather of the seven horizontal taps + bias + sigmoid in the column-patch kernel's tile epilogue (round 3)
            HIP_TRY(c, conv_col7_fused_launch(cur, fl.C0, n, c->H, c->W, c->final_w_swz, c->final_bias, v.out.prediction, s));
        } else {
            if (c->final_w_swz && fl.BN == 32)
                HIP_TRY(c, conv_col7_launch(cur, fl.C0, n, c->H, c->W, c->final_w_swz, v.final_part, s));
            else
                HIP_TRY(c, conv_launch(fl, io, s));
            HIP_TRY(c, final_shift_sum_launch(v.final_part, c->final_bias, n, c->H, c->W, v.out.prediction, s));
        }
    }
    if (v.out.frames_u8) HIP_TRY(c, to_u8_launch(v.out.prediction, n, c->H, c->W, v.out.frames_u8, s));
    account(7);
#undef STAGE_MARK
    return EAMM_OK;   // (the caller records the last stage mark, after joining the chains)
}


int eamm_forward_frames(eamm_ctx* c, int n, const float* kd_val, const float* kd_jac, const float* ks_val,
                        const float* ks_jac, const eamm_outputs* o, void* stream_) {
    if (!c || !kd_val || !ks_val || !o || !o->prediction) return fail(c, EAMM_ERR_ARG, "null argument");
    if (!c->finalized) return fail(c, EAMM_ERR_STATE, "call eamm_finalize_weights first");
    if (c->ns_cached < 1) return fail(c, EAMM_ERR_STATE, "no source cached: call eamm_encode_source first");
    if (n < 1 || n > c->cfg.max_frames) return fail(c, EAMM_ERR_ARG, "n=%d outside [1,%d]", n, c->cfg.max_frames);
    if (kd_jac != nullptr && ks_jac == nullptr)
        return fail(c, EAMM_ERR_ARG, "kp_driving jacobian given without kp_source jacobian");
    const int ns = c->ns_cached;
    if (ns != 1 && ns != n) return fail(c, EAMM_ERR_ARG, "%d cached sources cannot serve %d frames (need 1 or n)", ns, n);
    if ((o->occlusion_map) && !c->cfg.estimate_occlusion_map)
        return fail(c, EAMM_ERR_ARG, "occlusion_map requested but estimate_occlusion_map is off");
    if (c->nb == 0 && (o->mask || o->sparse_deformed || o->deformed || o->deformation))
        return fail(c, EAMM_ERR_ARG, "this generator has no motion network: only 'prediction' exists (generator.py:64-95)");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const eamm_outputs* const user = o;
    eamm_outputs staged = *o;
    if (o->frames_u8 && c->Cimg != 3) return fail(c, EAMM_ERR_ARG, "uint8 RGB frames need num_channels == 3");
    if (c->Cimg < 3) {   // one or two image channels: the pass writes three-channel staging tensors (eamm_ctx.h: Cimg)
        if (o->frames_u8) return fail(c, EAMM_ERR_ARG, "uint8 RGB frames need num_channels == 3");
        staged.prediction = c->stage_pred;
        if (o->deformed) staged.deformed = c->stage_deformed;
        if (o->sparse_deformed) staged.sparse_deformed = c->stage_sparse;
        o = &staged;
    }

    StreamLease lease(c, s);
    if (lease.rc) return lease.rc;
    hipEvent_t* ev = nullptr;  // stage boundaries (of the main stream's sequence), recorded only while profiling
    hipEvent_t* cev = nullptr; // bottleneck window of every whole-pass chain
    const int chains = pass_chains(c, n);
    c->cur_pass_chains = chains;
    c->cur_call_frames = n;
    if (c->profiling && c->prof_used < eamm_ctx::PROF_CALLS) {
        ev = c->prof_events.data() + (size_t)c->prof_used * (eamm_ctx::NMARK + 1 + eamm_ctx::NSUB);
        cev = c->prof_chain_ev.data() + (size_t)c->prof_used * eamm_ctx::MAXCHAIN * 2;
        c->prof_n.push_back(n);
        c->prof_sub.push_back(0);
        c->prof_marks.push_back(0);
        c->prof_nchain.push_back(std::min(chains, (int)eamm_ctx::MAXCHAIN));
        c->prof_flops.push_back(0.0);
        c->prof_flops_bneck.push_back(0.0);
        ++c->prof_used;
    }
    (void)take_mfma_flops();
    c->call_flops = c->call_flops_bneck = 0.0;
    for (double& f : c->call_stage_flops) f = 0.0;
    if (chains == 1) {
        if (int rc = forward_view(c, make_view(c, 0, n, ns, 0, kd_val, kd_jac, ks_val, ks_jac, o), s, ev, false, cev)) return rc;
    } else {
        const int nbase = n / chains, nrem = n % chains;   // the first nrem chains take one more frame
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        for (int k = 1; k < chains; ++k) HIP_TRY(c, hipStreamWaitEvent(c->side_streams[k - 1], c->ev_fork, 0));
        // EAMM_WARP_JOINT: the chains meet once, after their flow heads: ONE warp launch over all frames (an HBM-bound kernel
        // that fills the chip alone; per chain it runs beside the other chain's kernels at half the rate), then they part again
        const bool joint = c->warp_joint && c->nb > 0 && c->ev_warp != nullptr;
        for (int pass = 0; pass < (joint ? 2 : 1); ++pass) {
            for (int k = 0; k < chains; ++k) {
                const int nk = nbase + (k < nrem ? 1 : 0), f0 = k * nbase + std::min(k, nrem);
                const FrameView v = make_view(c, f0, nk, ns, k, kd_val, kd_jac, ks_val, ks_jac, o);
                const FrameView all = make_view(c, 0, n, ns, 0, kd_val, kd_jac, ks_val, ks_jac, o);
                hipStream_t sk = k ? c->side_streams[k - 1] : s;
                if (int rc = forward_view(c, v, sk, k ? nullptr : ev, true, (cev && k < eamm_ctx::MAXCHAIN) ? cev + 2 * k : nullptr, k,
                                          joint ? (pass == 0 ? PH_PRE : PH_POST) : PH_ALL, (joint && pass == 1 && k == 0) ? &all : nullptr))
                    return rc;
                if (joint && pass == 0 && k > 0) HIP_TRY(c, hipEventRecord(c->ev_join[k - 1], sk));
            }
        }
        for (int k = 1; k < chains; ++k) {
            HIP_TRY(c, hipEventRecord(c->ev_join[k - 1], c->side_streams[k - 1]));
            HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[k - 1], 0));
        }
    }
    if (c->Cimg < 3) {   // the real channels of every frame: rows of C planes out of rows of three
        const size_t HW = (size_t)c->H * c->W * sizeof(float), hw = (size_t)c->h * c->w * sizeof(float), C = (size_t)c->Cimg;
        HIP_TRY(c, hipMemcpy2DAsync(user->prediction, C * HW, c->stage_pred, 3 * HW, C * HW, (size_t)n, hipMemcpyDeviceToDevice, s));
        if (user->deformed)
            HIP_TRY(c, hipMemcpy2DAsync(user->deformed, C * HW, c->stage_deformed, 3 * HW, C * HW, (size_t)n, hipMemcpyDeviceToDevice, s));
        if (user->sparse_deformed)
            HIP_TRY(c, hipMemcpy2DAsync(user->sparse_deformed, C * hw, c->stage_sparse, 3 * hw, C * hw, (size_t)n * (c->K + 1),
                                        hipMemcpyDeviceToDevice, s));
    }
    if (ev) {   // the last stage ends where the call ends (after the join)
        HIP_TRY(c, hipEventRecord(ev[eamm_ctx::NMARK], s));
        c->prof_marks.back() = eamm_ctx::NMARK + 1;
        c->prof_flops.back() = c->call_flops;
        c->prof_flops_bneck.back() = c->call_flops_bneck;
        for (int i = 0; i < eamm_ctx::NMARK; ++i)
            c->prof_stage_flops[(size_t)(c->prof_used - 1) * eamm_ctx::NMARK + i] = c->call_stage_flops[i];
    }
    return EAMM_OK;
}

int eamm_profile_enable(eamm_ctx* c, int on) {
    if (!c) return EAMM_ERR_ARG;
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (on && c->prof_events.empty()) {
        c->prof_events.resize((size_t)eamm_ctx::PROF_CALLS * (eamm_ctx::NMARK + 1 + eamm_ctx::NSUB));
        for (auto& e : c->prof_events) HIP_TRY(c, hipEventCreate(&e));
        c->prof_chain_ev.resize((size_t)eamm_ctx::PROF_CALLS * eamm_ctx::MAXCHAIN * 2);
        for (auto& e : c->prof_chain_ev) HIP_TRY(c, hipEventCreate(&e));
        c->prof_stage_flops.assign((size_t)eamm_ctx::PROF_CALLS * eamm_ctx::NMARK, 0.0);
    }
    c->profiling = on != 0;
    return EAMM_OK;
}

int eamm_profile_read(eamm_ctx* c, double* stage_ms, int nstage, int64_t* calls, int64_t* frames, int reset) {
    if (!c || !stage_ms || nstage != eamm_ctx::NSTAGE) return fail(c, EAMM_ERR_ARG, "nstage must be %d", eamm_ctx::NSTAGE);
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    // stage index of each of the NMARK recorded intervals; the bottleneck interval (5) is split below
    static const int stage_of[eamm_ctx::NMARK] = {0, 1, 2, 3, 4, 6, 7, 8};
    int dropped = 0;
    for (int k = 0; k < c->prof_used; ++k) {  // fold finished event sets into the totals
        hipEvent_t* ev = c->prof_events.data() + (size_t)k * (eamm_ctx::NMARK + 1 + eamm_ctx::NSUB);
        // a call that failed midway leaves a partly recorded set: skip it (its events are re-recorded by a later call)
        bool ok = c->prof_marks[k] == eamm_ctx::NMARK + 1 && hipEventSynchronize(ev[eamm_ctx::NMARK]) == hipSuccess;
        double ms_set[eamm_ctx::NSTAGE] = {0};
        for (int i = 0; ok && i < eamm_ctx::NMARK; ++i) {
            float ms = 0.f;
            ok = hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess;
            ms_set[stage_of[i]] += ms;
        }
        const int nsub = c->prof_sub[k];
        if (ok && nsub > 1) {  // Winograd form: even intervals are input transforms, odd ones the GEMM kernels
            hipEvent_t* sub = ev + eamm_ctx::NMARK + 1;
            double tr = 0;
            for (int i = 0; ok && i + 1 < nsub; i += 2) {
                float ms = 0.f;
                ok = hipEventElapsedTime(&ms, sub[i], sub[i + 1]) == hipSuccess;
                tr += ms;
            }
            ms_set[5] += tr;
            ms_set[6] -= tr;
            for (int i = 1; ok && i + 1 < nsub; i += 2) {   // the GEMM kernels' own durations on the main stream
                float ms = 0.f;
                ok = hipEventElapsedTime(&ms, sub[i], sub[i + 1]) == hipSuccess;
                ms_set[9] += ms;
            }
        }
        if (ok) {   // bottleneck windows of the whole-pass chains, on the clock of this call's first event
            hipEvent_t* cv = c->prof_chain_ev.data() + (size_t)k * eamm_ctx::MAXCHAIN * 2;
            const int nc = c->prof_nchain[k];
            float t0[eamm_ctx::MAXCHAIN], t1[eamm_ctx::MAXCHAIN];
            for (int j = 0; ok && j < nc; ++j)
                ok = hipEventSynchronize(cv[2 * j + 1]) == hipSuccess && hipEventElapsedTime(&t0[j], ev[0], cv[2 * j]) == hipSuccess &&
                     hipEventElapsedTime(&t1[j], ev[0], cv[2 * j + 1]) == hipSuccess;
            if (ok) {
                int order[eamm_ctx::MAXCHAIN];
                for (int j = 0; j < nc; ++j) order[j] = j;
                std::sort(order, order + nc, [&](int a, int b) { return t0[a] < t0[b]; });
                double uni = 0, sum = 0, cur0 = 0, cur1 = -1;
                for (int q = 0; q < nc; ++q) {
                    const int j = order[q];
                    sum += t1[j] - t0[j];
                    if (cur1 < cur0 || t0[j] > cur1) {   // a new disjoint interval
                        if (cur1 >= cur0) uni += cur1 - cur0;
                        cur0 = t0[j];
                        cur1 = t1[j];
                    } else {
                        cur1 = std::max<double>(cur1, t1[j]);
                    }
                }
                if (cur1 >= cur0) uni += cur1 - cur0;
                ms_set[10] = uni;
                ms_set[11] = sum;
                ms_set[12] = c->prof_flops[k] * 1e-9;
                ms_set[13] = c->prof_flops_bneck[k] * 1e-9;
                for (int i = 0; i < eamm_ctx::NMARK; ++i) ms_set[14 + i] = c->prof_stage_flops[(size_t)k * eamm_ctx::NMARK + i] * 1e-9;
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            ++dropped;
            continue;
        }
        for (int i = 0; i < eamm_ctx::NSTAGE; ++i) c->prof_ms[i] += ms_set[i];
        c->prof_calls += 1;
        c->prof_frames += c->prof_n[k];
    }
    (void)dropped;
    c->prof_used = 0;
    c->prof_n.clear();
    c->prof_sub.clear();
    c->prof_marks.clear();
    c->prof_nchain.clear();
    c->prof_flops.clear();
    c->prof_flops_bneck.clear();
    for (int i = 0; i < nstage; ++i) stage_ms[i] = c->prof_ms[i];
    if (calls) *calls = c->prof_calls;
    if (frames) *frames = c->prof_frames;
    if (reset) {
        for (double& v : c->prof_ms) v = 0;
        c->prof_calls = c->prof_frames = 0;
    }
    return EAMM_OK;
}

int eamm_check_numeric(eamm_ctx* c, void* stream_) {
    if (!c) return EAMM_ERR_ARG;
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int flag = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(c, hipMemcpyAsync(&flag, c->bad_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (flag) {
        HIP_TRY(c, hipMemsetAsync(c->bad_flag, 0, sizeof(int), s));
        return fail(c, EAMM_ERR_NUMERIC, "singular key-point jacobian (torch.inverse would raise, dense_motion.py:56)");
    }
    return EAMM_OK;
}

size_t eamm_source_cache_bytes(const eamm_ctx* c, int ns) {
    if (!c) return 0;
    const size_t per = (size_t)c->hf * c->wf * c->Cb + (size_t)c->G * c->h * c->w * 4 + (size_t)c->Cpl * c->H * c->W;
    return per * (size_t)ns * sizeof(float);
}

int eamm_export_source_cache(eamm_ctx* c, void* dst, int ns, void* stream_) {
    if (!c || !dst) return fail(c, EAMM_ERR_ARG, "null argument");
    if (ns < 1 || ns > c->ns_cached) return fail(c, EAMM_ERR_STATE, "only %d sources are cached", c->ns_cached);
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const size_t a = (size_t)ns * c->hf * c->wf * c->Cb, b = (size_t)ns * c->h * c->w * 4, d = (size_t)ns * c->Cpl * c->H * c->W;
    const size_t gstride = (size_t)c->cfg.max_sources * c->h * c->w * 4;   // src_small is [group][max_sources][h][w][4]
    float* p = reinterpret_cast<float*>(dst);
    HIP_TRY(c, hipMemcpyAsync(p, c->feat, a * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int g = 0; g < c->G; ++g)
        HIP_TRY(c, hipMemcpyAsync(p + a + g * b, c->src_small + g * gstride, b * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(p + a + c->G * b, c->src_full, d * sizeof(float), hipMemcpyDeviceToDevice, s));
    return EAMM_OK;
}

int eamm_import_source_cache(eamm_ctx* c, const void* src, int ns, void* stream_) {
    if (!c || !src) return fail(c, EAMM_ERR_ARG, "null argument");
    if (!c->finalized) return fail(c, EAMM_ERR_STATE, "call eamm_finalize_weights first");
    if (ns < 1 || ns > c->cfg.max_sources) return fail(c, EAMM_ERR_ARG, "ns=%d outside [1,%d]", ns, c->cfg.max_sources);
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const size_t a = (size_t)ns * c->hf * c->wf * c->Cb, b = (size_t)ns * c->h * c->w * 4, d = (size_t)ns * c->Cpl * c->H * c->W;
    const size_t gstride = (size_t)c->cfg.max_sources * c->h * c->w * 4;
    const float* p = reinterpret_cast<const float*>(src);
    HIP_TRY(c, hipMemcpyAsync(c->feat, p, a * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int g = 0; g < c->G; ++g)
        HIP_TRY(c, hipMemcpyAsync(c->src_small + g * gstride, p + a + g * b, b * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->src_full, p + a + c->G * b, d * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->ns_cached = ns;
    return EAMM_OK;
}

double eamm_flops_per_frame(const eamm_ctx* c) { return c ? c->flops_frame : 0.0; }

int eamm_bottleneck_form(const eamm_ctx* c, int n) { return (c && n > 0) ? bottleneck_form(c, n) : EAMM_ERR_ARG; }
int eamm_bottleneck_chains(const eamm_ctx* c, int n) {
    if (!c || n <= 0) return EAMM_ERR_ARG;
    const int whole = pass_chains(c, n);   // chains over the whole pass take precedence over chains inside the bottleneck
    return whole > 1 ? whole : bottleneck_chains(c, n);
}
int eamm_pass_chains(const eamm_ctx* c, int n) {
    if (!c || n <= 0) return EAMM_ERR_ARG;
    return pass_chains(c, n);
}
int eamm_last_stream_set(const eamm_ctx* c) { return c ? c->last_streams : EAMM_ERR_ARG; }
int eamm_build_experiments(void) {
#ifdef EAMM_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
int eamm_knobs_json(char* buf, int cap) { return knobs_json(buf, cap); }
double eamm_total_mfma_flops(void) { return total_mfma_flops(); }
int eamm_describe_plan(const eamm_ctx* c, int n, char* buf, int cap) {
    if (!c || n <= 0 || n > c->cfg.max_frames || !c->finalized) return EAMM_ERR_ARG;
    const int pc = pass_chains(c, n), nk = pc > 1 ? n / pc : n;
    const int form = bottleneck_form(c, nk), bc = pc > 1 ? pc : bottleneck_chains(c, n);
    std::string enc = "[";
    for (int i = 0; i < c->nb; ++i) {   // the test forward_view applies per hourglass encoder level, for one chain's frames
        const int tiles = nk * ((c->h >> i) / 4) * ((c->w >> i) / 4);
        const bool heavy = i < (int)c->w4enc.size() && 144.0 * c->w4enc[i].Cin * c->w4enc[i].Cout > 100e6;
        const bool w4 = i < (int)c->w4enc.size() && c->w4enc[i].Cout && tiles >= c->enc_wino_min_tiles * (heavy ? 4 : 1) &&
                        288e-6 * tiles * c->w4enc[i].Cin * c->w4enc[i].Cout >= (double)c->enc_wino_min_mflop;
        enc += std::string(i ? ", " : "") + (w4 ? "\"wino4\"" : "\"direct\"");
    }
    enc += "]";
    const bool fused = c->final_w_swz && c->final_fused && nk * ((c->H + 15) / 16) >= c->final_fused_min_rows;
    char tmp[1024];
    const int len = snprintf(tmp, sizeof tmp,
                             "{\"frames\": %d, \"pass_chains\": %d, \"frames_per_chain\": %d, \"bottleneck_form\": %d, \"bottleneck_chains\": %d, "
                             "\"wino4_groups\": %d, \"wino4_variant\": %d, \"hg_encoder\": %s, \"final\": \"%s\", \"warp_joint\": %d, "
                             "\"bneck_stagger\": %d, \"side_streams\": %d, \"experiments_build\": %d}",
                             n, pc, nk, form, bc, (form == 4 && pc == 1 && bc == 1) ? wino4_groups(c, n) : 1, wino4_variant_for(c, n), enc.c_str(),
                             fused ? "col7q fused" : "col7 + shift-sum", c->warp_joint, c->bneck_stagger, (int)c->pool_streams.size(),
                             eamm_build_experiments());
    if (buf && cap > 0) {
        const int m = std::min(len, cap - 1);
        std::memcpy(buf, tmp, (size_t)m);
        buf[m] = 0;
    }
    return len;
}
double eamm_encode_flops(const eamm_ctx* c) { return c ? c->flops_encode : 0.0; }

int eamm_op_conv(int device, const float* in0, int C0, const float* in1, int C1, int B, int Hin, int Win, int up,
                 const float* w_host, const float* b_host, int Cout, int kh, int kw, int act, int pool,
                 const float* resid, int splitk, int tile_n, float* out, int iters, float* avg_ms, void* stream_) {
    const bool shape_ok = (kh == 3 && kw == 3) || (kh == 7 && kw == 7) || (kh == 7 && kw == 1);
    if (!in0 || !w_host || !b_host || !out || !shape_ok || C0 % CONV_BK || C1 % CONV_BK || (up && (kh != 3 || kw != 3)))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv: bad argument");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (tile_n == 4002) {   // the final layer whole: 7x7, 3 output channels, bias, sigmoid -> NCHW, on the fused column-patch kernel
        if (kh != 7 || kw != 7 || Cout != 3 || act != 2 || up || pool || resid || C1 || C0 % 32 || C0 > 64 || C0 < 32)
            return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv: unsupported fused final-layer configuration");
        std::vector<float> wr((size_t)21 * C0 * 7), packed(conv_packed_elems(7, C0, 21, 32, 1));
        for (int dx = 0; dx < 7; ++dx)          // w'[dx*3+co][c][dy] = w[co][c][dy][dx]  (MODE_ROWSPLIT of build_layer)
            for (int co = 0; co < 3; ++co)
                for (int ci = 0; ci < C0; ++ci)
                    for (int dy = 0; dy < 7; ++dy)
                        wr[((size_t)(dx * 3 + co) * C0 + ci) * 7 + dy] = w_host[((size_t)co * C0 + ci) * 49 + dy * 7 + dx];
        conv_pack_host(wr.data(), 21, C0, 7, 1, nullptr, C0, 32, false, true, packed.data());
        float *wd = nullptr, *bd = nullptr;
        hipError_t e = hipMalloc((void**)&wd, packed.size() * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&bd, 3 * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(wd, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bd, b_host, 3 * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = conv_col7_fused_launch(in0, C0, B, Hin, Win, wd, bd, out, s);
        if (e == hipSuccess && iters > 0 && avg_ms) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < iters && e == hipSuccess; ++i) e = conv_col7_fused_launch(in0, C0, B, Hin, Win, wd, bd, out, s);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            *avg_ms = ms / iters;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (wd) (void)hipFree(wd);
        if (bd) (void)hipFree(bd);
        return e == hipSuccess ? EAMM_OK : fail(nullptr, EAMM_ERR_HIP, "fused final layer failed: %s", hipGetErrorString(e));
    }
    if (tile_n == 4000 || tile_n == 4001) {  // column-patch kernels of the 7x1 convolutions: 4000 final layer (Cout = the 32-float
        // pixel stride, weights resident), 4001 flow head (two inputs, Cout = 96, streamed weights); no bias (the gather kernels add it)
        const bool streamed = tile_n == 4001;
        const int bn = streamed ? 96 : 32;
        if (kh != 7 || kw != 1 || up || pool || resid || splitk > 1 || Cout != bn || act != 0 || C0 % 32 || C1 % 32 ||
            (!streamed && (C1 || C0 > 64)))
            return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv: unsupported column-patch configuration");
        std::vector<float> packed(conv_packed_elems(7, C0 + C1, bn, bn, 1));
        conv_pack_host(w_host, bn, C0 + C1, 7, 1, nullptr, C0 + C1, bn, false, true, packed.data());
        float* wd = nullptr;
        int rc = EAMM_OK;
        auto bad = [&](hipError_t e, const char* what) {
            if (e != hipSuccess) rc = fail(nullptr, EAMM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
            return e != hipSuccess;
        };
        if (!bad(hipMalloc((void**)&wd, packed.size() * sizeof(float)), "hipMalloc") &&
            !bad(hipMemcpy(wd, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy")) {
            auto run = [&]() {
                return streamed ? conv_col7s_launch(in0, C0, in1, C1, B, Hin, Win, wd, 3, out, 96, s)
                                : conv_col7_launch(in0, C0, B, Hin, Win, wd, out, s);
            };
            if (!bad(run(), "col7 launch") && iters > 0 && avg_ms) {
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0);
                (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, s);
                for (int i = 0; i < iters; ++i) (void)run();
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *avg_ms = ms / iters;
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
            bad(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
        if (wd) (void)hipFree(wd);
        return rc;   // bias is not applied here (the gather kernel adds it in the pipeline): callers pass zeros
    }
    if (tile_n == 3000 || tile_n == 3003) {   // spatial-patch kernels for the up-convolution: 3000 collapsed-phase form, 3003 polyphase minimal-filtering form
        const bool pp = tile_n == 3003;
        const int psplit = splitk > 1 ? splitk : 1;   // polyphase form only: the channel reduction over that many workgroups
        if (kh != 3 || kw != 3 || !up || pool || resid || (Cout & 3) || (psplit > 1 && (!pp || ((C0 + C1) / 32) % psplit)))
            return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv: unsupported patch-kernel configuration");
        float* ppart = nullptr;
        const size_t ppart_elems = psplit > 1 ? (size_t)psplit * B * 4 * Hin * Win * Cout : 0;
        PatchLayer P;
        P.C0 = C0;
        P.C1 = C1;
        P.Cout = Cout;
        std::vector<float> packed(pp ? patch_poly_packed_elems(C0 + C1, Cout) : patch_packed_elems(C0 + C1, Cout));
        std::vector<float> bias((size_t)((Cout + 63) / 64) * 64, 0.f);
        if (pp) patch_poly_pack_host(w_host, Cout, C0 + C1, nullptr, C0 + C1, packed.data());
        else patch_pack_host(w_host, Cout, C0 + C1, nullptr, C0 + C1, packed.data());
        std::copy(b_host, b_host + Cout, bias.begin());
        float** wslot = pp ? &P.w_poly : &P.w;
        int rc = EAMM_OK;
        auto bad = [&](hipError_t e, const char* what) {
            if (e != hipSuccess) rc = fail(nullptr, EAMM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
            return e != hipSuccess;
        };
        if (!bad(hipMalloc((void**)wslot, packed.size() * sizeof(float)), "hipMalloc") &&
            !bad(hipMalloc((void**)&P.bias, bias.size() * sizeof(float)), "hipMalloc") &&
            !bad(hipMemcpy(*wslot, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy") &&
            !bad(hipMemcpy(P.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy") &&
            (psplit == 1 || !bad(hipMalloc((void**)&ppart, ppart_elems * sizeof(float)), "hipMalloc"))) {
            auto run = [&]() {
                return pp ? patch_poly_launch(P, in0, in1, B, Hin, Win, act, out, s, psplit, ppart, ppart_elems)
                          : patch_phase_launch(P, in0, in1, B, Hin, Win, act, out, s);
            };
            if (!bad(run(), "patch launch") && iters > 0 && avg_ms) {
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0);
                (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, s);
                for (int i = 0; i < iters; ++i) (void)run();
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *avg_ms = ms / iters;
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
            bad(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
        if (P.w) (void)hipFree(P.w);
        if (P.w_poly) (void)hipFree(P.w_poly);
        if (ppart) (void)hipFree(ppart);
        if (P.bias) (void)hipFree(P.bias);
        return rc;
    }
    if (tile_n >= 2000 && tile_n < 2200) {  // Winograd: input transform + GEMM; 2000 + variant F(2x2,3x3), 2100 + variant F(4x4,3x3),
        // 2150 + groups (2, 3, 6): F(4x4,3x3) with the transform-point rows split over workgroups + output-transform kernel
        const bool w4 = tile_n >= 2100;
        const int w4_groups = tile_n >= 2150 ? tile_n - 2150 : 1;
        if (kh != 3 || kw != 3 || up || (pool && (w4_groups == 1 || resid)) || C1 || splitk > 1 || C0 % 64 || (Cout & 3) ||
            (w4 && ((Hin & 3) || (Win & 3))))
            return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv: unsupported Winograd configuration");
        WinoLayer W;
        W.Cin = C0;
        W.Cout = Cout;
        W.tile = w4 ? 4 : 2;
        W.BN = w4 ? 64 : 128;
        W.ntiles = (Cout + W.BN - 1) / W.BN;
        std::vector<float> packed(w4 ? wino4_packed_elems(Cout, C0, W.BN) : wino_packed_elems(Cout, C0, W.BN)),
            bias((size_t)W.ntiles * W.BN, 0.f);
        if (w4)
            wino4_pack_host(w_host, Cout, C0, W.BN, packed.data());
        else
            wino_pack_host(w_host, Cout, C0, W.BN, packed.data());
        std::copy(b_host, b_host + Cout, bias.begin());
        float *V = nullptr, *Zb = nullptr;
        int rc = EAMM_OK;
        auto done = [&](hipError_t e, const char* what) {
            if (e != hipSuccess) rc = fail(nullptr, EAMM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
            return e != hipSuccess;
        };
        const size_t vel = (size_t)16 * B * (Hin / 2) * (Win / 2) * C0;
        if (!done(hipMalloc((void**)&W.u, packed.size() * sizeof(float)), "hipMalloc") &&
            !done(hipMalloc((void**)&W.bias, bias.size() * sizeof(float)), "hipMalloc") &&
            !done(hipMalloc((void**)&V, vel * sizeof(float)), "hipMalloc") &&
            !done(hipMalloc((void**)&Zb, (size_t)(w4_groups == 12 ? 48 : 24) * B * ((Hin + 3) / 4) * ((Win + 3) / 4) * Cout * sizeof(float) + 16), "hipMalloc") &&
            !done(hipMemcpy(W.u, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy") &&
            !done(hipMemcpy(W.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy")) {
            // timing-mode knob: EAMM_OP_WINO_PART = 1 times the input transform alone, 2 the GEMM alone
            int part = env_int("EAMM_OP_WINO_PART", 0);
            bool timing = false;
            auto run = [&]() {
                const bool tr = !timing || part != 2, gm = !timing || part != 1;
                hipError_t e = hipSuccess;
                if (w4) {
                    if (tr) e = wino4_transform_launch(in0, nullptr, nullptr, B, Hin, Win, C0, V, s);
                    if (e != hipSuccess || !gm) return e;
                    return wino4_gemm_launch(W, V, B, Hin, Win, act, resid, out, s, w4_groups > 1 ? 0 : tile_n - 2100, w4_groups, Zb, pool);
                }
                if (tr) e = wino_transform_launch(in0, nullptr, nullptr, B, Hin, Win, C0, V, s);
                if (e != hipSuccess || !gm) return e;
                return wino_gemm_launch(W, V, B, Hin, Win, act, resid, out, s, tile_n - 2000);
            };
            if (!done(run(), "winograd launch") && iters > 0 && avg_ms) {
                timing = true;
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0);
                (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, s);
                for (int i = 0; i < iters; ++i) (void)run();
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *avg_ms = ms / iters;
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
            done(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
        if (W.u) (void)hipFree(W.u);
        if (W.bias) (void)hipFree(W.bias);
        if (V) (void)hipFree(V);
        if (Zb) (void)hipFree(Zb);
        return rc;
    }
    ConvLayer L;
    L.kh = kh;
    L.kw = kw;
    L.phase = up != 0;
    L.C0 = C0;
    L.C1 = C1;
    L.Cout = Cout;
    L.BN = tile_n > 0 ? tile_n : conv_tile_n(Cout);
    if (tile_n > 1000) {  // 1000 + id selects an LDS-DMA big-tile configuration
        L.dma_cfg = tile_n - 1000;
        if (!conv_dma_tile(L.dma_cfg, &L.BM, &L.BN) || kw == 1 || kh == 7)
            return fail(nullptr, EAMM_ERR_ARG, "unknown / unsupported dma tile");
    } else if (L.BN != 32 && L.BN != 64 && L.BN != 128) {
        return fail(nullptr, EAMM_ERR_ARG, "tile_n must be 32/64/128 or 1000+dma id");
    }
    L.ntiles = (Cout + L.BN - 1) / L.BN;
    const int taps = L.phase ? 4 : kh * kw;
    L.nchunks = taps * ((C0 + C1) / CONV_BK);
    std::vector<float> packed(conv_packed_elems(taps, C0 + C1, Cout, L.BN, L.phase ? 4 : 1));
    conv_pack_host(w_host, Cout, C0 + C1, kh, kw, nullptr, C0 + C1, L.BN, L.phase, L.dma_cfg > 0, packed.data());
    std::vector<float> bias((size_t)L.ntiles * L.BN, 0.f);
    std::copy(b_host, b_host + Cout, bias.begin());
    int rc = EAMM_OK;
    float* partial = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        if (L.w) (void)hipFree(L.w);
        if (L.bias) (void)hipFree(L.bias);
        if (partial) (void)hipFree(partial);
    };
#define OP_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            rc = fail(nullptr, EAMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));      \
            cleanup();                                                                            \
            return rc;                                                                            \
        }                                                                                         \
    } while (0)
    OP_TRY(hipMalloc((void**)&L.w, packed.size() * sizeof(float)));
    OP_TRY(hipMalloc((void**)&L.bias, bias.size() * sizeof(float)));
    OP_TRY(hipMemcpy(L.w, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(L.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    const int M = B * Hin * Win;
    const ConvPlan pl = conv_plan(L, M, splitk);
    if (pl.partial_elems) OP_TRY(hipMalloc((void**)&partial, pl.partial_elems * sizeof(float)));
    ConvIO io{};
    io.in0 = in0;
    io.in1 = C1 ? in1 : nullptr;
    io.B = B;
    io.Hin = Hin;
    io.Win = Win;
    io.act = act;
    io.pool = pool;
    io.resid = resid;
    io.out = out;
    io.partial = partial;
    io.partial_cap = pl.partial_elems;
    OP_TRY(conv_launch(L, io, s, splitk));
    if (iters > 0 && avg_ms) {  // kernel timing: `iters` back-to-back launches between two HIP events on `s`
        hipEvent_t e0, e1;
        OP_TRY(hipEventCreate(&e0));
        OP_TRY(hipEventCreate(&e1));
        OP_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) OP_TRY(conv_launch(L, io, s, splitk));
        OP_TRY(hipEventRecord(e1, s));
        OP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        OP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    cleanup();
#undef OP_TRY
    return EAMM_OK;
}

int eamm_op_one_euro(int device, const float* x, int T, int E, float mincutoff, float beta, float dcutoff, float freq, float scale,
                     float* out, float* state, int resume, void* stream_) {
    if (!x || !out || T < 0 || E < 1 || (resume && !state)) return fail(nullptr, EAMM_ERR_ARG, "eamm_op_one_euro: bad argument");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    const hipError_t e = one_euro_launch(x, T, E, mincutoff, beta, dcutoff, freq, scale, out, reinterpret_cast<hipStream_t>(stream_), state, resume);
    return e == hipSuccess ? EAMM_OK : fail(nullptr, EAMM_ERR_HIP, "one_euro_kernel: %s", hipGetErrorString(e));
}

int eamm_op_warp(int device, const float* feat, const float* deformation, const float* occlusion, int n, int ns, int hf,
                 int wf, int C, int h, int w, float* out, int iters, float* avg_ms, void* stream_) {
    if (!feat || !deformation || !out || n < 1 || (ns != 1 && ns != n) || hf < 1 || wf < 1 || h < 1 || w < 1 || C < 8 || (C & 7))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_warp: bad argument");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    auto run = [&]() { return warp_features_launch(feat, deformation, occlusion, n, ns, hf, wf, C, h, w, out, nullptr, nullptr, nullptr, s); };
    hipError_t e = run();
    if (e == hipSuccess && iters > 0 && avg_ms) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && e == hipSuccess; ++i) e = run();
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    // (no stream synchronisation outside the timing branch: like every other operator entry the work is only ENQUEUED on
    // `stream` -- the differentiable forward calls this three times per pass and must stay capturable)
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_warp failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

int eamm_op_warp_backward(int device, const float* feat, const float* deformation, const float* occlusion, const float* grad_out,
                          int n, int ns, int hf, int wf, int C, float* grad_feat, float* grad_deformation, float* grad_occlusion,
                          void* stream_) {
    if (!feat || !deformation || !grad_out || n < 1 || (ns != 1 && ns != n) || hf < 1 || wf < 1 || C < 4 || (C & 3) ||
        (!grad_feat && !grad_deformation && !grad_occlusion) || (grad_occlusion && !occlusion))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_warp_backward: bad argument");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    // the kernel accumulates: the gradients start from zero here, on the caller's stream
    hipError_t e = hipSuccess;
    if (grad_feat) e = hipMemsetAsync(grad_feat, 0, (size_t)ns * hf * wf * C * sizeof(float), s);
    if (e == hipSuccess && grad_deformation) e = hipMemsetAsync(grad_deformation, 0, (size_t)n * hf * wf * 2 * sizeof(float), s);
    if (e == hipSuccess && grad_occlusion) e = hipMemsetAsync(grad_occlusion, 0, (size_t)n * hf * wf * sizeof(float), s);
    if (e == hipSuccess)
        e = warp_features_backward_launch(feat, deformation, occlusion, grad_out, n, ns, hf, wf, C, grad_feat, grad_deformation,
                                          grad_occlusion, s);
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_warp_backward failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

// ---- convolution with DEVICE weights (training path): pack on the device into the caller's workspace, then the forward kernels
namespace {
struct ConvDevPlan {
    bool wino4 = false;
    int groups = 1;         // F(4x4): rows of transform points split over this many workgroups per tile (few tiles), as the engine does
    size_t zbuf = 0;        // floats of the split form's x-folded products
    int BN = 0, ntiles = 0;
    size_t packed = 0, bias = 0, aux = 0;   // floats: packed filter, padded bias, Winograd V or split-K partials
    ConvLayer L;
};
bool conv_dev_plan(int B, int H, int W, int Cin, int Cout, int kh, int kw, ConvDevPlan* P) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || Cin % CONV_BK || (Cout & 3)) return false;
    if (!((kh == 3 && kw == 3) || (kh == 7 && kw == 7))) return false;
    const long long Mq = (long long)B * (H / 4) * (W / 4);
    // F(4x4,3x3) where the bottleneck's kernel applies and the tile count fills the chip; else the register-staged implicit GEMM
    static const int wino_off = knob_int("EAMM_CONV_DEV_WINO4", 1) == 0;
    static const long long wino_min_tiles = knob_int("EAMM_CONV_DEV_WINO4_MIN_TILES", 2048);
    P->wino4 = !wino_off && kh == 3 && kw == 3 && Cin % (2 * CONV_BK) == 0 && !(H & 3) && !(W & 3) && Mq >= wino_min_tiles &&
               (unsigned long long)36 * Mq * Cin * sizeof(float) < 0xFFFFF000ull &&
               wino4_packed_elems(Cout, Cin, 64) * sizeof(float) < 0xFFFFF000ull;
    if (P->wino4) {
        P->BN = 64;
        P->ntiles = (Cout + 63) / 64;
        P->packed = wino4_packed_elems(Cout, Cin, 64);
        P->aux = (size_t)36 * Mq * Cin;
        // a single launch of <= 128 64-tile workgroups leaves half the chip idle (the engine runs two chains there): split
        // the transform-point rows over 2 / 3 / 6 workgroups per tile, as the engine's one-chain calls do
        const long long nb = ((Mq + 63) / 64) * P->ntiles;
        for (int gsel : {6, 3, 2})
            if (nb * gsel <= 256) {
                P->groups = gsel;
                break;
            }
        if (P->groups > 1) P->zbuf = (size_t)24 * Mq * Cout;
    } else {
        ConvLayer& L = P->L;
        L.kh = kh;
        L.kw = kw;
        L.C0 = Cin;
        L.Cout = Cout;
        L.BN = conv_tile_n(Cout);
        L.ntiles = (Cout + L.BN - 1) / L.BN;
        L.nchunks = kh * kw * (Cin / CONV_BK);
        P->BN = L.BN;
        P->ntiles = L.ntiles;
        P->packed = conv_packed_elems(kh * kw, Cin, Cout, L.BN, 1);
        P->aux = conv_plan(L, B * H * W, 0).partial_elems;
    }
    P->bias = (size_t)P->ntiles * P->BN;
    auto up = [](size_t n) { return (n + 63) / 64 * 64; };   // 256-byte aligned sections
    P->packed = up(P->packed);
    P->bias = up(P->bias);
    P->aux = up(P->aux);
    P->zbuf = up(P->zbuf);
    return true;
}
}  // namespace

namespace {
// A per-device vector of zeros (never written after its allocation): the bias of the convolutions that have none.
const float* device_zeros(int device, size_t n, hipStream_t stream) {
    constexpr size_t CAP = 16384;
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static float* zeros[MAXDEV] = {};
    if (n > CAP || device < 0 || device >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!zeros[device]) {
        // first use: hipMalloc + a synchronous hipMemset -- neither is legal while `stream` is being captured into a graph (they
        // would invalidate the capture): the caller then pads its bias with a kernel instead, and a later eager call allocates
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        void* p = nullptr;
        if (hipMalloc(&p, CAP * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, CAP * sizeof(float)) != hipSuccess) {   // synchronous: visible to every stream that follows
            (void)hipFree(p);
            return nullptr;
        }
        zeros[device] = static_cast<float*>(p);
    }
    return zeros[device];
}
}  // namespace

size_t eamm_op_conv_dev_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    ConvDevPlan P;
    return conv_dev_plan(B, H, W, Cin, Cout, kh, kw, &P) ? P.packed + P.bias + P.aux + P.zbuf : 0;
}

int eamm_op_conv_dev(int device, const float* x, int B, int H, int W, int Cin, const float* weight, const float* bias, int Cout, int kh,
                     int kw, int transposed, float* out, float* workspace, size_t workspace_floats, void* stream_) {
    ConvDevPlan P;
    if (!x || !weight || !out || !workspace || !conv_dev_plan(B, H, W, Cin, Cout, kh, kw, &P))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv_dev: bad argument (3x3 or 7x7, Cin a multiple of 32, Cout of 4)");
    if (workspace_floats < P.packed + P.bias + P.aux + P.zbuf || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv_dev: workspace too small or misaligned (%zu floats needed)", P.packed + P.bias + P.aux + P.zbuf);
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    float *wp = workspace, *bp = workspace + P.packed, *aux = workspace + P.packed + P.bias;
    // the kernels read ntiles * BN bias values: a bias of exactly that length is read where it lies, a missing one (every data
    // gradient) is the device's shared zero vector -- the padding kernel only runs for a Cout off the tile grid
    hipError_t e = hipSuccess;
    const int nbias = P.ntiles * P.BN;
    if (bias && nbias == Cout) {
        bp = const_cast<float*>(bias);
    } else if (const float* z = bias ? nullptr : device_zeros(device, (size_t)nbias, s)) {
        bp = const_cast<float*>(z);
    } else {
        e = bias_pad_dev_launch(bias, Cout, nbias, bp, s);
    }
    if (e == hipSuccess && P.wino4) {
        WinoLayer WL;
        WL.Cin = Cin;
        WL.Cout = Cout;
        WL.tile = 4;
        WL.BN = 64;
        WL.ntiles = P.ntiles;
        WL.u = wp;
        WL.bias = bp;
        e = wino4_pack_dev_launch(weight, Cout, Cin, 64, transposed, wp, s);
        if (e == hipSuccess) e = wino4_transform_launch(x, nullptr, nullptr, B, H, W, Cin, aux, s);
        if (e == hipSuccess)
            e = wino4_gemm_launch(WL, aux, B, H, W, ACT_NONE, nullptr, out, s, P.groups > 1 ? 0 : 3, P.groups,
                                  P.groups > 1 ? aux + P.aux : nullptr, 0);
    } else if (e == hipSuccess) {
        ConvLayer& L = P.L;
        L.w = wp;
        L.bias = bp;
        e = conv_pack_dev_launch(weight, Cout, Cin, kh * kw, L.BN, transposed, wp, s);
        ConvIO io{};
        io.in0 = x;
        io.B = B;
        io.Hin = H;
        io.Win = W;
        io.out = out;
        io.partial = P.aux ? aux : nullptr;
        io.partial_cap = P.aux;
        if (e == hipSuccess) e = conv_launch(L, io, s, 0);
    }
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_conv_dev failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

// ---- the 7x7 layers with three channels on one side (conv7_thin.hip)
size_t eamm_op_conv7_thin_workspace_floats(int B, int H, int W, int N) {
    if (B < 1 || H < 1 || W < 1 || (N != 32 && N != 64)) return 0;
    return conv7_thin_workspace_floats(B, H, W, N);
}

int eamm_op_conv7_thin(int device, const float* thin, const float* weight, const float* bias, int B, int H, int W, int N, int transposed,
                       float* out, float* workspace, size_t workspace_floats, void* stream_) {
    if (!thin || !weight || !out || !workspace || B < 1 || H < 1 || W < 1 || (N != 32 && N != 64) ||
        workspace_floats < conv7_thin_workspace_floats(B, H, W, N) || (long long)B * H * W * 64 >= (1ll << 31))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv7_thin: bad argument (N = 32 | 64, workspace of eamm_op_conv7_thin_workspace_floats)");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipError_t e = conv7_thin_in_launch(thin, weight, bias, B, H, W, N, transposed, out, workspace, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_conv7_thin failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

int eamm_op_conv7_thin_wgrad(int device, const float* thin, const float* wide, int B, int H, int W, int N, int thin_is_input,
                             float* grad_weight, float* workspace, size_t workspace_floats, void* stream_) {
    if (!thin || !wide || !grad_weight || !workspace || B < 1 || H < 1 || W < 1 || (N != 32 && N != 64) ||
        workspace_floats < conv7_thin_workspace_floats(B, H, W, N) || (long long)B * H * W * 64 >= (1ll << 31))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv7_thin_wgrad: bad argument");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipError_t e = conv7_thin_wgrad_launch(thin, wide, B, H, W, N, thin_is_input, grad_weight, workspace, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_conv7_thin_wgrad failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

int eamm_op_final_conv_sigmoid(int device, const float* x, const float* weight, const float* bias, int B, int H, int W, int C,
                               float* out_nchw, float* workspace, size_t workspace_floats, void* stream_) {
    if (!x || !weight || !bias || !out_nchw || !workspace || B < 1 || H < 1 || W < 1 || (C != 32 && C != 64) ||
        workspace_floats < (size_t)7 * C * 32 || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_final_conv_sigmoid: bad argument (C = 32 | 64, workspace of 7 * C * 32 floats)");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    hipError_t e = col7_pack_dev_launch(weight, C, workspace, s);
    if (e == hipSuccess) e = conv_col7_fused_launch(x, C, B, H, W, workspace, bias, out_nchw, s);
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_final_conv_sigmoid failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

size_t eamm_op_conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1) return 0;
    return conv_wgrad_workspace_floats(B, H, W, Cin, Cout, kh, kw);
}

int eamm_op_conv_wgrad(int device, const float* x, const float* grad_out, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                       float* grad_weight, float* grad_bias, float* workspace, size_t workspace_floats, void* stream_) {
    if (!x || !grad_out || !grad_weight || !workspace || B < 1 || H < 1 || W < 1 || Cin < 4 || Cout < 4 || (Cin & 3) || (Cout & 3) ||
        Cout > 1024 || kh < 1 || kw < 1 || !(kh & 1) || !(kw & 1) || kh > 7 || kw > 7)
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv_wgrad: bad argument (channels multiples of 4, Cout <= 1024, odd filter up to 7x7)");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipError_t e = conv_wgrad_launch(x, grad_out, B, H, W, Cin, Cout, kh, kw, grad_weight, grad_bias, workspace, workspace_floats,
                                     reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_conv_wgrad failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

size_t eamm_op_conv_saved_transform_offset(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    ConvDevPlan P;
    if (!conv_dev_plan(B, H, W, Cin, Cout, kh, kw, &P) || !P.wino4 || !conv_wgrad_takes_transformed(B, H, W, Cin, Cout, kh, kw))
        return (size_t)-1;
    return P.packed + P.bias;
}

int eamm_op_conv_wgrad_saved(int device, const float* x_transformed, const float* grad_out, int B, int H, int W, int Cin, int Cout,
                             float* grad_weight, float* grad_bias, float* workspace, size_t workspace_floats, void* stream_) {
    if (!x_transformed || !grad_out || !grad_weight || !workspace || eamm_op_conv_saved_transform_offset(B, H, W, Cin, Cout, 3, 3) == (size_t)-1 ||
        Cout > 1024)
        return fail(nullptr, EAMM_ERR_ARG, "eamm_op_conv_wgrad_saved: only where eamm_op_conv_saved_transform_offset names a saved transform");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");
    hipError_t e = conv_wgrad_launch(nullptr, grad_out, B, H, W, Cin, Cout, 3, 3, grad_weight, grad_bias, workspace, workspace_floats,
                                     reinterpret_cast<hipStream_t>(stream_), x_transformed);
    if (e != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "eamm_op_conv_wgrad_saved failed: %s", hipGetErrorString(e));
    return EAMM_OK;
}

// ---- differentiable dense-motion front end and flow head (motion.hip forward kernels, motion_backward.hip) -----------------------
#define MOTION_ENTRY(cond, msg)                                                                  \
    if (cond) return fail(nullptr, EAMM_ERR_ARG, msg);                                           \
    DeviceGuard guard(device);                                                                   \
    if (guard.status != hipSuccess) return fail(nullptr, EAMM_ERR_HIP, "hipSetDevice failed");   \
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_)
#define MOTION_DONE(e, name) \
    return (e) == hipSuccess ? EAMM_OK : fail(nullptr, EAMM_ERR_HIP, name " failed: %s", hipGetErrorString(e))

int eamm_op_antialias_down(int device, const float* source, const float* aa_weight, int B, int H, int W, int inv_scale, float* small,
                           void* stream_) {
    MOTION_ENTRY(!source || !small || B < 1 || H < 1 || W < 1 || (inv_scale != 1 && inv_scale != 4) || (inv_scale == 4 && !aa_weight) ||
                     H % inv_scale || W % inv_scale,
                 "eamm_op_antialias_down: bad argument (inv_scale 1 | 4: the 13x13 kernel of scale_factor 0.25)");
    hipError_t e = antialias_down_launch(source, aa_weight, B, H, W, inv_scale, 4, small, s);
    MOTION_DONE(e, "eamm_op_antialias_down");
}

int eamm_op_antialias_down_backward(int device, const float* grad_small, const float* aa_weight, int B, int H, int W, int inv_scale,
                                    float* grad_source, void* stream_) {
    MOTION_ENTRY(!grad_small || !grad_source || B < 1 || H < 1 || W < 1 || (inv_scale != 1 && inv_scale != 4) ||
                     (inv_scale == 4 && !aa_weight) || H % inv_scale || W % inv_scale,
                 "eamm_op_antialias_down_backward: bad argument");
    hipError_t e = antialias_down_backward_launch(grad_small, aa_weight, B, H, W, inv_scale, grad_source, s);
    MOTION_DONE(e, "eamm_op_antialias_down_backward");
}

int eamm_op_kp_records(int device, const float* kd_val, const float* kd_jac, const float* ks_val, const float* ks_jac, int n, int K,
                       float* records, int* singular_flag, void* stream_) {
    MOTION_ENTRY(!kd_val || !ks_val || !records || !singular_flag || n < 1 || K < 1 || K > 31 || ((kd_jac == nullptr) != (ks_jac == nullptr)),
                 "eamm_op_kp_records: bad argument");
    hipError_t e = kp_prepare_launch(kd_val, kd_jac, ks_val, ks_jac, n, n, K, records, singular_flag, s);
    MOTION_DONE(e, "eamm_op_kp_records");
}

int eamm_op_kp_records_backward(int device, const float* kd_jac, const float* ks_jac, const float* grad_records, int n, int K,
                                float* grad_kd_val, float* grad_ks_val, float* grad_kd_jac, float* grad_ks_jac, void* stream_) {
    MOTION_ENTRY(!grad_records || n < 1 || K < 1 || ((kd_jac == nullptr) != (ks_jac == nullptr)) ||
                     ((grad_kd_jac || grad_ks_jac) && !kd_jac),
                 "eamm_op_kp_records_backward: bad argument");
    hipError_t e = kp_records_backward_launch(kd_jac, ks_jac, grad_records, n, K, grad_kd_val, grad_ks_val, grad_kd_jac, grad_ks_jac, s);
    MOTION_DONE(e, "eamm_op_kp_records_backward");
}

size_t eamm_op_motion_workspace_floats(int n, int K, int h, int w) {
    if (n < 1 || K < 1 || h < 1 || w < 1) return 0;
    return motion_backward_workspace_floats(n, K, h, w);
}

int eamm_op_motion_front(int device, const float* records, const float* small, int n, int K, int h, int w, float kp_variance, int Cpad,
                         float* hourglass_in, float* sparse_deformed, void* stream_) {
    MOTION_ENTRY(!records || !small || !hourglass_in || n < 1 || K < 1 || K > 31 || h < 2 || w < 2 || Cpad < 4 * (K + 1) || (Cpad & 3) ||
                     !(kp_variance > 0.f),
                 "eamm_op_motion_front: bad argument");
    hipError_t e = motion_front_launch(records, small, n, n, K, h, w, kp_variance, Cpad, hourglass_in, sparse_deformed, s);
    MOTION_DONE(e, "eamm_op_motion_front");
}

int eamm_op_motion_front_backward(int device, const float* records, const float* small, int n, int K, int h, int w, float kp_variance,
                                  int Cpad, const float* grad_hourglass_in, const float* grad_sparse_deformed, float* grad_small,
                                  float* grad_records, float* workspace, size_t workspace_floats, void* stream_) {
    MOTION_ENTRY(!records || !small || !grad_records || !workspace || (!grad_hourglass_in && !grad_sparse_deformed) || n < 1 || K < 1 ||
                     K > 31 || h < 2 || w < 2 || Cpad < 4 * (K + 1) || (Cpad & 3) || !(kp_variance > 0.f) ||
                     workspace_floats < motion_backward_workspace_floats(n, K, h, w),
                 "eamm_op_motion_front_backward: bad argument");
    hipError_t e = motion_front_backward_launch(records, small, n, K, h, w, kp_variance, Cpad, grad_hourglass_in, grad_sparse_deformed,
                                                grad_small, grad_records, workspace, s);
    MOTION_DONE(e, "eamm_op_motion_front_backward");
}

int eamm_op_motion_head(int device, const float* mask_logits, int ld, const float* occlusion_logits, int ldo, const float* records, int n,
                        int K, int h, int w, float* mask, float* deformation, float* occlusion, void* stream_) {
    MOTION_ENTRY(!mask_logits || !records || !mask || !deformation || n < 1 || K < 1 || K > 31 || h < 2 || w < 2 || ld < K + 1 ||
                     (occlusion_logits && (ldo < 1 || !occlusion)),
                 "eamm_op_motion_head: bad argument");
    hipError_t e = motion_head_forward_launch(mask_logits, ld, occlusion_logits, ldo, records, n, K, h, w, mask, deformation, occlusion, s);
    MOTION_DONE(e, "eamm_op_motion_head");
}

int eamm_op_motion_head_backward(int device, const float* mask, const float* occlusion, const float* records, int n, int K, int h, int w,
                                 const float* grad_mask, const float* grad_deformation, const float* grad_occlusion,
                                 float* grad_mask_logits, int ld, float* grad_occlusion_logits, int ldo, float* grad_records,
                                 float* workspace, size_t workspace_floats, void* stream_) {
    MOTION_ENTRY(!mask || !records || !grad_mask_logits || !grad_records || !workspace || n < 1 || K < 1 || K > 31 || h < 2 || w < 2 ||
                     ld < K + 1 || (grad_occlusion_logits && (ldo < 1 || !occlusion)) ||
                     workspace_floats < motion_backward_workspace_floats(n, K, h, w),
                 "eamm_op_motion_head_backward: bad argument");
    hipError_t e = motion_head_backward_launch(mask, occlusion, records, n, K, h, w, grad_mask, grad_deformation, grad_occlusion,
                                               grad_mask_logits, ld, grad_occlusion_logits, ldo, grad_records, workspace, s);
    MOTION_DONE(e, "eamm_op_motion_head_backward");
}
#undef MOTION_ENTRY
#undef MOTION_DONE

}  // extern "C"
