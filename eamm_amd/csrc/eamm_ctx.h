// The generator handle (private to the C-ABI translation units eamm_api.hip and eamm_train_api.hip).
#pragma once
#include "../../include/eamm_hip.h"
#include "api_common.h"

#include <functional>

using namespace eamm;

struct eamm_ctx : eamm::CtxBase {
    eamm_config cfg{};
    int ns_cached = 0;

    // derived geometry
    int H = 0, W = 0, h = 0, w = 0, hf = 0, wf = 0, K = 0, nb = 0, nd = 0;
    int Cb = 0;                 // bottleneck channels
    int Cp0 = 0;                // hourglass input channels padded to a multiple of 32
    int Csrc = 32;              // RGB source padded for the 7x7 MFMA encoder conv
    std::vector<int> enc_c;     // hourglass encoder output channels e_1..e_nb
    std::vector<int> dec_c;     // hourglass decoder output channels u_0..u_{nb-1}
    std::vector<int> down_c;    // generator encoder channels [be, ...]
    std::vector<int> up_c;      // generator decoder output channels
    // The four lists above and Cb hold the widths the KERNELS run at: the reference's widths rounded up to the 32-channel granule.
    // The *_r lists are the reference's own (what the state_dict has): when they differ, eamm_finalize_weights first pads the
    // state_dict into the equivalent wider network (zero filters, identity BatchNorm on the extra channels: they carry exact zeros).
    std::vector<int> enc_r, dec_r, down_r, up_r;
    int Cb_r = 0;
    bool padded_widths = false;
    // Image channels (num_channels, reference modules/generator.py:14,25,46): the kernels are written for three (a float4 per pixel:
    // RGB + a zero; the hourglass input line is (heat, R, G, B) per motion).  One or two channels run as the equivalent
    // three-channel network -- zero filters on the missing input channels of `first` / the hourglass's first block / the flow
    // head, zero filters + bias on the missing outputs of `final` (pad_state_dict) -- on a zero-extended source; the three-channel
    // results land in the staging buffers below and only the real channels are copied to the caller's tensors.
    int Cimg = 3;
    int G = 1, Cpl = 3;         // four to six image channels: G = 2 groups of three, Cpl = 6 planes in the zero-extended source copy
    float *stage_pred = nullptr, *stage_deformed = nullptr, *stage_sparse = nullptr;

    // layers
    ConvLayer first, final_conv, head;
    float *first7_w = nullptr, *first7_bias = nullptr;   // the first block on its dedicated 3-channel kernel (conv_first.hip), or null
    int first7 = 1;                                       // EAMM_FIRST7: 0 = the generic 7x7 kernel on the 32-channel-padded source
    std::vector<LayerSet> down, hg_enc, hg_dec, res1, res2, up;
    std::vector<WinoLayer> wres1, wres2;   // Winograd F(2x2,3x3) packing of the bottleneck convolutions
    std::vector<WinoLayer> w4res1, w4res2; // Winograd F(4x4,3x3) packing (bottleneck maps with sides divisible by 4)
    std::vector<WinoLayer> w4down;         // ... and of the generator encoder's DownBlock2d levels (source encoding, once per clip / per module call without cache)
    std::vector<WinoLayer> w4enc;          // F(4x4,3x3) packing of the hourglass encoder convolutions (Cout == 0: level not eligible)
    int enc_wino = 1;                      // hourglass DownBlock2d levels in F(4x4,3x3) form + pooled output transform (EAMM_ENC_WINO)
    int enc_wino_min_mflop = 3000;         // ... for levels of at least this many direct-form MFLOP per call (EAMM_ENC_WINO_MIN_MFLOP): smaller ones are
                                           // launch-bound and one launch beats three (measured 256x256: 1 frame 810 vs 804, 4 frames equal,
                                           // 8 frames 2896 -> 2960, 12 frames 2849 -> 2968, 16 frames 3529 -> 3659; 512x512 x 8: 911 -> 948 frames/s)
    int enc_wino_min_tiles = 32;           // ... and of at least this many 4x4 tiles (EAMM_ENC_WINO_MIN_TILES; x4 for levels with > 100 MB of transformed
                                           // weights): the 4x4-map level has 16 tiles at 16 frames against 151 MB and stays direct
    int bneck_chains = 2;                  // bottleneck as this many chains of frames on as many streams (EAMM_BNECK_CHAINS; 1 = off)
    int pass_chains = 0;                   // the whole per-frame pass as this many chains (EAMM_PASS_CHAINS; 1 = off; 0 = the default: two chains
                                           // when each chain's F(4x4) GEMM keeps enough workgroups (pass_chains_min_blocks); otherwise off)
    int pass_chains_min_blocks = 80;       // automatic mode: fewest bottleneck-GEMM workgroups per chain (EAMM_PASS_CHAINS_MIN_BLOCKS)
    int pass_chains_min_frames = 8;        // ... from this many frames per call (EAMM_PASS_CHAINS_MIN_FRAMES)
    std::vector<hipStream_t> side_streams; // the other chains' streams of the call being enqueued: pool_streams or own_streams (StreamLease in eamm_api.hip)
    std::vector<hipStream_t> pool_streams; // ... the device's shared pool (chain_stream(), not owned)
    std::vector<hipStream_t> own_streams;  // ... this handle's private set: used while the caller's stream is being captured or another
                                           // thread holds the pool; created on first need, destroyed with the handle
    int private_streams = 0;               // EAMM_PRIVATE_STREAMS=1: always the private set
    int last_streams = 0;                  // which set the last call used: 0 pool, 1 private (pool taken), 2 private (capture)
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_stagger = nullptr;       // recorded by the first chain after its first bottleneck input transform
    hipEvent_t ev_warp = nullptr;          // recorded by the first chain behind the joint warp launch (EAMM_WARP_JOINT)
    int enc_cus_pct = 100;                 // EAMM_ENC_CUS_PCT: the hourglass encoder's F(4x4) point-row split is sized for this share of the chip's CUs
                                           // (measured hg_enc ms per step: 25 %: 0.342, 50 %: 0.256, 100 %: 0.242, 200 %: 0.252, 400 %: 0.251)
    int epi_v = 0;                         // EAMM_WINO4_EPI_V=1: timing experiment (wrong results): see conv_winograd4.hip DBG 30
    float* epi_scratch = nullptr;
    int warp_joint = 0;                    // one feature-warp launch for all chains' frames
    int bneck_stagger = 0;                 // EAMM_BNECK_STAGGER=1: the other whole-pass chains start their bottleneck stage behind that event, so the
                                           // chains' HBM-bound input transforms run beside the other chain's GEMM instead of beside each other
    std::vector<hipEvent_t> ev_join;
    int cur_pass_chains = 1;               // whole-pass chains of the call being enqueued
    int cur_call_frames = 0;               // ... and its frames (all chains)
    int wino_tile = 4;                     // preferred output tile (EAMM_WINO_TILE): 4 -> F(4x4) where it applies, 2 -> F(2x2)
    int wino4_variant = 6;                 // wino4_gemm_kernel pipeline variant (3: one DMA piece per 8 MFMAs; 2.062 -> 2.047 ms per step vs one per 4;
                                           // 6 = 3 with the V stream loaded non-temporally where every GEMM workgroup of the call has a CU
                                           // of its own, 3 for larger calls: wino4_variant_for, eamm_api.hip)
    bool wino4_variant_pinned = false;     // EAMM_WINO4_VARIANT set: that variant for every call size
    int cus = 256;                         // the device's CU count, read ONCE in eamm_create (the per-call launch rules size their splits by it)
    int wino4_groups_knob = 0;             // EAMM_WINO4_GROUPS read once in eamm_create (0: unset -> the rule in wino4_groups())
    float* wino_v = nullptr;               // [16][F*hf*wf/4][Cb] (F(2x2)) or [36][F*hf*wf/16][Cb] transformed activations
    float* wino_z = nullptr;               // [24][F*hf*wf/16][Cb] x-folded products of the split F(4x4) form (few tiles; [48] for the half-row split)
    size_t wino_z_elems = 0;               // its size in floats
    int wino_min_m = 49152;                // smallest pixel count for which the bottleneck runs in Winograd F(2x2) form
    int wino4_min_m = 0;                   // ... in F(4x4) form: with the transform-point rows split over workgroups it wins from one frame up
    int wino_variant = 0;                  // wino_gemm_kernel pipeline variant (see wino_gemm_launch); in-pipeline all are within noise
    std::vector<float*> pre_s, pre_t;  // res-block pre-activation scale/shift (norm1)
    float* aa_w = nullptr;
    float* head_bias = nullptr;    // mask / occlusion biases when the head runs row-split (applied by the head kernel)
    int head_nc = 0;               // > 0: head is a 7x1 convolution over (dx, co), co < head_nc
    float* head_w_swz = nullptr;   // its weights with a 96-wide N tile in LDS-DMA layout (conv_col7s_kernel), or null
    float* final_bias = nullptr;   // bias of the final conv, applied by the shift-sum kernel
    float* final_part = nullptr;   // [F,H,W,32] (dx,co) partial products of the final 7x7 conv
    float* final_w_swz = nullptr;  // the 7x1 weights in LDS-DMA layout for the column-patch kernel (conv_col7.hip)
    int col7 = 1;                  // EAMM_COL7: 0 = im2col-style kernel for the final convolution
    int final_fused = 1;           // EAMM_FINAL_FUSED: 0 = partial products to HBM + final_shift_sum_kernel (round 2)
    int final_fused_min_rows = 128; // ... from this many rows of 16x16 tiles per call (EAMM_FINAL_FUSED_MIN_ROWS)
    int head_col7_min_tiles = 128; // fewest 16x16 tiles for which the flow head uses the column-patch kernel (EAMM_HEAD_COL7_MIN_TILES)

    // source cache (exportable): feat [S,hf,wf,Cb], src_small [S,h,w,4], src_full [S,3,H,W]
    float *feat = nullptr, *src_small = nullptr, *src_full = nullptr;
    // encoder temporaries
    float* src_nhwc = nullptr;
    std::vector<float*> enc_tmp;  // first output, then each down-block output except the last (= feat)
    // per-frame workspace
    float* kp_rec = nullptr;
    int* bad_flag = nullptr;
    float* hg_in = nullptr;
    std::vector<float*> e_buf, u_buf;
    float *logits = nullptr, *deformation = nullptr, *occlusion = nullptr;
    float *xa = nullptr, *xb = nullptr, *act = nullptr, *tmp = nullptr;
    std::vector<float*> up_buf;
    float* partial = nullptr;
    size_t partial_elems = 0;

    double flops_frame = 0, flops_encode = 0;

    // optional stage timing with HIP events on the caller's stream (bench.py roofline leg)
    static constexpr int NSTAGE = 22;      // front, hg_enc, hg_dec, head, warp, bneck_transform, bneck_conv, up, final + bneck_gemm_kernel
                                           // + bneck_union_ms (wall time during which ANY chain is in its bottleneck stage), bneck_windows_ms (sum of
                                           // the chains' bottleneck windows), exec_gflop / bneck_exec_gflop (executed MFMA GFLOP of the recorded
                                           // calls, all chains: whole pass / bottleneck GEMMs) + 14..21 executed MFMA GFLOP of each of the eight
                                           // recorded stage intervals (front, hg_enc, hg_dec, head, warp, bottleneck, up, final), all chains
    static constexpr int MAXCHAIN = 4;     // chains whose bottleneck window is recorded per call
    static constexpr int NMARK = 8;        // stage boundaries recorded per call (bottleneck is split from sub-events)
    static constexpr int NSUB = 64;        // per-launch events inside the bottleneck (4 per res-block + 1)
    static constexpr int PROF_CALLS = 256; // event sets kept before the host must read them
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;   // PROF_CALLS * (NMARK+1 + NSUB)
    std::vector<hipEvent_t> prof_chain_ev; // PROF_CALLS * MAXCHAIN * 2: start / end of each whole-pass chain's bottleneck stage
    std::vector<int> prof_nchain;          // chains recorded by each call
    std::vector<double> prof_flops, prof_flops_bneck;   // executed MFMA flops of each recorded call
    double call_flops = 0, call_flops_bneck = 0;         // ... of the call being enqueued
    double call_stage_flops[8] = {0};                    // ... per stage interval (all chains)
    std::vector<double> prof_stage_flops;                // PROF_CALLS * NMARK
    std::vector<int> prof_sub;             // sub-events used by each recorded call (0: direct form)
    std::vector<int> prof_marks;           // stage marks each recorded call completed (NMARK + 1 unless it failed midway)
    int prof_used = 0;
    double prof_ms[NSTAGE] = {0};
    long prof_calls = 0, prof_frames = 0;
    std::vector<int> prof_n;

    // ---- training-mode forward (eamm_train_api.hip; SURVEY.md 8f row N4, second slice) ---------------------------------
    bool train_mode = false;               // eamm_set_training before finalisation: convolutions packed WITHOUT BatchNorm folded in
    float* raw = nullptr;                  // pre-BatchNorm convolution output of the site in flight (largest: F x H x W x down_c[1])
    float* train_sums = nullptr;           // the CALLER's [6 Cmax + 2] buffer: packed statistics of the site in flight (what the replicas all-reduce)
    float* train_stat = nullptr;           // [2 Cmax] mean, scale of the site in flight
    float* bn_work = nullptr;              // partial sums workspace
    int train_cmax = 0;
    std::vector<std::string> site_names;   // BatchNorm sites in execution order (state_dict prefixes)
    std::vector<std::function<int()>> steps;   // the forward as resumable steps; a step returns 1 when statistics await the
    size_t step_pc = 0;                        // caller's all-reduce, 0 to go on, < 0 on error
    int pending_c = 0;                     // channels of the site whose sums are pending
};

