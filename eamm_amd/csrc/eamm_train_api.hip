// Training-mode forward of OcclusionAwareGenerator (second slice): every BatchNorm of the
// generator uses the statistics of the batch -- reference sync_batchnorm/batchnorm.py:55-125 inside SameBlock2d / DownBlock2d /
// UpBlock2d / ResBlock2d (modules/util.py:858-938) when train.py:133 fine-tunes the generator -- so nothing can be folded
// into the convolutions: each of the 2 * num_down_blocks + 1 + 2 * dm_num_blocks + 2 * num_bottleneck_blocks sites runs
//     convolution with the RAW weights (the path's fp32-MFMA kernels, nothing fused behind them)
//     -> bn_nhwc_partial / bn_combine          per-channel sum, sum of squares, count of this replica's batch
//     -> [the caller all-reduces those 2C + 2 floats over the replicas: torch.distributed / RCCL]
//     -> bn_finalize                            mean, inverse standard deviation, running statistics (in the CALLER's tensors)
//     -> bn_nhwc_apply                          normalise, ReLU (+ the 2x2 average of DownBlock2d)
// The forward is therefore RESUMABLE: eamm_train_begin lays it out as a list of steps, eamm_train_next runs them up to the
// next statistics hand-over and returns 1 with the buffer to reduce, 0 when the pass is complete.  Everything else of the
// pass (anti-aliasing, heat-maps, sparse motions, flow head, warps, final convolution) is the evaluation path's kernels.
// This file is the graph-free FORWARD; the backward operators live in backward.hip, conv7_thin.hip, batchnorm*.hip and
// motion_backward.hip and are composed by eamm_amd/train_graph.py (which also offers this forward, `train_route = "operators"`).
#include "eamm_ctx.h"

namespace {

void train_site_names(const eamm_ctx* c, std::vector<std::string>* names) {
    const std::string dm = "dense_motion_network.hourglass.";
    names->clear();
    names->push_back("first.norm");
    for (int i = 0; i < c->nd; ++i) names->push_back("down_blocks." + std::to_string(i) + ".norm");
    for (int i = 0; i < c->nb; ++i) names->push_back(dm + "encoder.down_blocks." + std::to_string(i) + ".norm");
    for (int i = 0; i < c->nb; ++i) names->push_back(dm + "decoder.up_blocks." + std::to_string(i) + ".norm");
    for (int i = 0; i < c->cfg.num_bottleneck_blocks; ++i) {
        names->push_back("bottleneck.r" + std::to_string(i) + ".norm1");
        names->push_back("bottleneck.r" + std::to_string(i) + ".norm2");
    }
    for (int i = 0; i < c->nd; ++i) names->push_back("up_blocks." + std::to_string(i) + ".norm");
}

}  // namespace

extern "C" {

int eamm_set_training(eamm_ctx* c, int on) {
    if (!c) return EAMM_ERR_ARG;
    if (c->finalized) return fail(c, EAMM_ERR_STATE, "eamm_set_training must precede eamm_finalize_weights");
    c->train_mode = on != 0;
    if (c->train_mode) {   // nothing is folded, so none of the folded fast forms is packed
        c->wino_min_m = -1;
        c->enc_wino = 0;
        c->first7 = 0;
        c->patch_min_blocks = -1;
        c->col7 = 0;
        train_site_names(c, &c->site_names);
    }
    return EAMM_OK;
}

int eamm_train_num_sites(const eamm_ctx* c) { return (c && c->train_mode) ? (int)c->site_names.size() : 0; }

const char* eamm_train_site_name(const eamm_ctx* c, int i) {
    return (c && i >= 0 && i < (int)c->site_names.size()) ? c->site_names[i].c_str() : nullptr;
}

int eamm_train_begin(eamm_ctx* c, const float* source, int n, const float* kd_val, const float* kd_jac, const float* ks_val,
                     const float* ks_jac, const eamm_bn_site* sites, int nsites, float momentum, float eps, int sync,
                     float* sums_buffer, const eamm_outputs* o, void* stream_) {
    if (!c || !source || !sites || !sums_buffer || !o || !o->prediction) return fail(c, EAMM_ERR_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(sums_buffer) & 7) return fail(c, EAMM_ERR_ARG, "sums_buffer must be 8-byte aligned");
    if (!c->train_mode || !c->finalized) return fail(c, EAMM_ERR_STATE, "needs eamm_set_training(ctx, 1) before eamm_finalize_weights");
    if (n < 1 || n > c->cfg.max_frames || n > c->cfg.max_sources)
        return fail(c, EAMM_ERR_ARG, "n=%d outside [1, min(max_frames %d, max_sources %d)]", n, c->cfg.max_frames, c->cfg.max_sources);
    if (nsites != (int)c->site_names.size()) return fail(c, EAMM_ERR_ARG, "%d BatchNorm sites given, the generator has %d", nsites, (int)c->site_names.size());
    if (c->nb > 0 && (!kd_val || !ks_val)) return fail(c, EAMM_ERR_ARG, "key points missing");
    if (kd_jac != nullptr && ks_jac == nullptr) return fail(c, EAMM_ERR_ARG, "kp_driving jacobian given without kp_source jacobian");
    if (o->occlusion_map && !c->cfg.estimate_occlusion_map) return fail(c, EAMM_ERR_ARG, "occlusion_map requested but estimate_occlusion_map is off");
    if (c->nb == 0 && (o->mask || o->sparse_deformed || o->deformed || o->deformation))
        return fail(c, EAMM_ERR_ARG, "this generator has no motion network: only 'prediction' exists");
    for (int i = 0; i < nsites; ++i)
        if (!sites[i].weight || !sites[i].bias || !sites[i].running_mean || !sites[i].running_var)
            return fail(c, EAMM_ERR_ARG, "BatchNorm site %d (%s) has a null tensor", i, c->site_names[i].c_str());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const std::vector<eamm_bn_site> st(sites, sites + nsites);
    const eamm_outputs out = *o;
    const int H = c->H, W = c->W, h = c->h, w = c->w, hf = c->hf, wf = c->wf, K = c->K, nr = c->cfg.num_bottleneck_blocks;
    const bool occ = c->cfg.estimate_occlusion_map != 0;
    const int mode = sync ? EAMM_BN_SYNC : EAMM_BN_SINGLE;
    c->steps.clear();
    c->step_pc = 0;
    c->train_sums = sums_buffer;   // caller-owned: the tensor its all-reduce runs on
    int site = 0;
    auto step = [&](std::function<int()> f) { c->steps.push_back(std::move(f)); };
    // one BatchNorm site over x [B,Hx,Wx,C] (NHWC): statistics hand-over, then normalise + ReLU (+ 2x2 average) into `dst`
    auto bn_site = [&](const float* x, int B, int Hx, int Wx, int C, int pool, float* dst) {
        const eamm_bn_site sp = st[site++];
        step([=]() -> int {
            HIP_TRY(c, bn_nhwc_sums_launch(x, (long long)B * Hx * Wx, C, c->train_sums, c->bn_work, s));
            c->pending_c = C;
            return 1;
        });
        step([=]() -> int {
            HIP_TRY(c, bn_finalize_launch(c->train_sums, C, eps, momentum, mode, sp.weight, sp.running_mean, sp.running_var, c->train_stat,
                                          c->train_stat + C, nullptr, s));
            HIP_TRY(c, bn_nhwc_apply_launch(x, c->train_stat, c->train_stat + C, sp.bias, B, Hx, Wx, C, 1, pool, dst, s));
            return 0;
        });
    };
    auto conv = [&](const LayerSet* S, const ConvLayer* single, const float* in0, const float* in1, int Hin, int Win, const float* resid,
                    float* dst) {
        step([=]() -> int {
            ConvIO io{};
            io.in0 = in0;
            io.in1 = in1;
            io.B = n;
            io.Hin = Hin;
            io.Win = Win;
            io.act = ACT_NONE;
            io.resid = resid;
            io.out = dst;
            io.partial = c->partial;
            io.partial_cap = c->partial_elems;
            HIP_TRY(c, conv_launch(S ? pick(c, *S, (size_t)n * Hin * Win) : *single, io, s));
            return 0;
        });
    };

    // ---- encoder (generator.py:61-63): SameBlock2d, DownBlock2d x nd, on the n source images of the batch
    step([=]() -> int {
        HIP_TRY(c, hipMemcpyAsync(c->src_full, source, (size_t)n * 3 * H * W * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, source_prepare_launch(source, c->aa_w, n, H, W, c->cfg.dm_inv_scale, c->Csrc, c->src_nhwc, c->src_small, s));
        return 0;
    });
    conv(nullptr, &c->first, c->src_nhwc, nullptr, H, W, nullptr, c->raw);
    bn_site(c->raw, n, H, W, c->down_c[0], 0, c->enc_tmp[0]);
    for (int i = 0; i < c->nd; ++i) {
        conv(&c->down[i], nullptr, c->enc_tmp[i], nullptr, H >> i, W >> i, nullptr, c->raw);
        bn_site(c->raw, n, H >> i, W >> i, c->down_c[i + 1], 1, i == c->nd - 1 ? c->feat : c->enc_tmp[i + 1]);
    }
    // ---- dense motion (dense_motion.py:81-113)
    if (c->nb > 0) {
        step([=]() -> int {
            HIP_TRY(c, kp_prepare_launch(kd_val, kd_jac, ks_val, kd_jac ? ks_jac : nullptr, n, n, K, c->kp_rec, c->bad_flag, s));
            HIP_TRY(c, motion_front_launch(c->kp_rec, c->src_small, n, n, K, h, w, c->cfg.kp_variance, c->Cp0, c->hg_in, out.sparse_deformed, s));
            return 0;
        });
        for (int i = 0; i < c->nb; ++i) {   // Encoder: DownBlock2d (util.py:956-960)
            conv(&c->hg_enc[i], nullptr, i == 0 ? c->hg_in : c->e_buf[i - 1], nullptr, h >> i, w >> i, nullptr, c->raw);
            bn_site(c->raw, n, h >> i, w >> i, c->enc_c[i], 1, c->e_buf[i]);
        }
        for (int i = 0; i < c->nb; ++i) {   // Decoder: UpBlock2d on cat[u_{i-1}, e_{nb-i}] (util.py:981-987): nearest x2 + 3x3 collapsed
            const int Hin = h >> (c->nb - i), Win = w >> (c->nb - i);
            conv(&c->hg_dec[i], nullptr, i == 0 ? c->e_buf[c->nb - 1] : c->u_buf[i - 1], i == 0 ? nullptr : c->e_buf[c->nb - 1 - i], Hin,
                 Win, nullptr, c->raw);
            bn_site(c->raw, n, 2 * Hin, 2 * Win, c->dec_c[i], 0, c->u_buf[i]);
        }
        step([=]() -> int {   // mask / occlusion logits, softmax, flow, sigmoid (dense_motion.py:98-111): no BatchNorm here
            ConvIO io{};
            io.in0 = c->u_buf[c->nb - 1];
            io.in1 = c->hg_in;
            io.B = n;
            io.Hin = h;
            io.Win = w;
            io.act = ACT_NONE;
            io.out = c->logits;
            io.partial = c->partial;
            io.partial_cap = c->partial_elems;
            ConvLayer head = c->head;
            if (c->head_nc) {
                head.Cout = 128;
                HIP_TRY(c, conv_launch(head, io, s));
                HIP_TRY(c, motion_head_rowsplit_launch(c->logits, 128, c->head_nc, c->head_bias, c->kp_rec, n, K, h, w, occ ? 1 : 0,
                                                       c->deformation, c->occlusion, out.mask, out.occlusion_map, s));
            } else {
                head.Cout = 32;
                HIP_TRY(c, conv_launch(head, io, s));
                HIP_TRY(c, motion_head_launch(c->logits, c->kp_rec, n, K, h, w, occ ? 1 : 0, c->deformation, c->occlusion, out.mask,
                                              out.occlusion_map, s));
            }
            if (out.deformation)
                HIP_TRY(c, hipMemcpyAsync(out.deformation, c->deformation, (size_t)n * h * w * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
            // feature warp x occlusion (generator.py:79-84), 'deformed' (generator.py:86)
            HIP_TRY(c, warp_features_launch(c->feat, c->deformation, occ ? c->occlusion : nullptr, n, n, hf, wf, c->Cb, h, w, c->xa, nullptr,
                                            nullptr, nullptr, s));
            if (out.deformed) HIP_TRY(c, warp_image_launch(c->src_full, c->deformation, n, n, H, W, h, w, out.deformed, s));
            return 0;
        });
    } else {
        step([=]() -> int {
            HIP_TRY(c, hipMemcpyAsync(c->xa, c->feat, (size_t)n * hf * wf * c->Cb * sizeof(float), hipMemcpyDeviceToDevice, s));
            return 0;
        });
    }
    // ---- bottleneck: x + conv2(relu(norm2(conv1(relu(norm1(x))))))   (util.py:872-880)
    float *x = c->xa, *xn = c->xb;
    for (int i = 0; i < nr; ++i) {
        bn_site(x, n, hf, wf, c->Cb, 0, c->act);
        conv(&c->res1[i], nullptr, c->act, nullptr, hf, wf, nullptr, c->tmp);
        bn_site(c->tmp, n, hf, wf, c->Cb, 0, c->act);
        conv(&c->res2[i], nullptr, c->act, nullptr, hf, wf, x, xn);
        std::swap(x, xn);
    }
    // ---- up blocks (util.py:895-900) and the final 7x7 + sigmoid (generator.py:92-93)
    const float* cur = x;
    for (int i = 0; i < c->nd; ++i) {
        conv(&c->up[i], nullptr, cur, nullptr, hf << i, wf << i, nullptr, c->raw);
        bn_site(c->raw, n, hf << (i + 1), wf << (i + 1), c->up_c[i], 0, c->up_buf[i]);
        cur = c->up_buf[i];
    }
    step([=]() -> int {
        ConvIO io{};
        io.in0 = cur;
        io.B = n;
        io.Hin = H;
        io.Win = W;
        io.act = ACT_NONE;
        io.out = c->final_part;
        io.partial = c->partial;
        io.partial_cap = c->partial_elems;
        ConvLayer fl = c->final_conv;
        fl.Cout = 32;
        HIP_TRY(c, conv_launch(fl, io, s));
        HIP_TRY(c, final_shift_sum_launch(c->final_part, c->final_bias, n, H, W, out.prediction, s));
        if (out.frames_u8) HIP_TRY(c, to_u8_launch(out.prediction, n, H, W, out.frames_u8, s));
        return 0;
    });
    if (site != nsites) return fail(c, EAMM_ERR_STATE, "internal: %d sites laid out, %d expected", site, nsites);
    c->ns_cached = 0;   // the source cache now holds this batch's sources: not a clip's
    return EAMM_OK;
}

int eamm_train_max_channels(const eamm_ctx* c) { return (c && c->train_mode) ? c->train_cmax : 0; }

int eamm_train_next(eamm_ctx* c, int* nfloats) {
    if (!c || !nfloats) return fail(c, EAMM_ERR_ARG, "null argument");
    DeviceGuard guard(c->device);
    if (guard.status != hipSuccess) return fail(c, EAMM_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    while (c->step_pc < c->steps.size()) {
        const int r = c->steps[c->step_pc++]();
        if (r < 0) {
            c->steps.clear();
            c->step_pc = 0;
            return r;
        }
        if (r == 1) {
            *nfloats = 2 * c->pending_c + 2;
            return 1;
        }
    }
    c->steps.clear();
    c->step_pc = 0;
    *nfloats = 0;
    return 0;
}

}  // extern "C"
