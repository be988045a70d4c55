/*
 * eamm_hip.h -- C ABI of libeamm_hip.so: the MI355X (gfx950) implementation of EAMM's
 * dense-motion + OcclusionAwareGenerator forward path.
 *
 * The reference has no FFI layer for this path: its boundary is the Python nn.Module
 * `OcclusionAwareGenerator` (reference modules/generator.py:8-97).  The entry points below are what
 * a binding for that module needs (see INTEGRATION.md for the ctypes stub); every one cites the
 * reference interface it stands in for.
 *
 * Conventions
 *   - every function returns 0 on success, a negative eamm_status otherwise; the message is
 *     available from eamm_last_error(ctx) (ctx may be NULL for a failed eamm_create);
 *   - the CALLER owns every device buffer passed in or out; the library owns only its handle, the
 *     packed weights, the cached source tensors and its workspace;
 *   - all work is enqueued on the caller's stream (a hipStream_t passed as void*); no call
 *     synchronises the device except eamm_create / eamm_finalize_weights / eamm_destroy;
 *   - one handle per (device, stream); handles are independent, so different host threads may
 *     drive different handles concurrently (reference threading model: one module replica per
 *     device, train.py:53-63);
 *   - boundary tensors are float32, contiguous, NCHW -- the reference's layout -- unless stated.
 */
#ifndef EAMM_HIP_H_
#define EAMM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAMM_ABI_VERSION 5

typedef enum eamm_status {
    EAMM_OK = 0,
    EAMM_ERR_ARG = -1,        /* bad argument / unsupported configuration            */
    EAMM_ERR_STATE = -2,      /* call order (weights not finalised, no source cached) */
    EAMM_ERR_KEY = -3,        /* unknown / missing / mis-shaped state_dict entry      */
    EAMM_ERR_HIP = -4,        /* HIP runtime error                                    */
    EAMM_ERR_NUMERIC = -5     /* singular key-point jacobian                          */
} eamm_status;

typedef struct eamm_ctx eamm_ctx;

/*
 * Constructor arguments of OcclusionAwareGenerator (reference modules/generator.py:14-15) with
 * dense_motion_params (reference modules/dense_motion.py:12-13) flattened in, plus the two sizes
 * the workspace is allocated for.  Any positive channel widths: those that are not multiples of the kernels' 32-channel
 * granule are served by padding the state_dict at eamm_finalize_weights (the extra channels carry exact zeros); a
 * training-mode handle (eamm_set_training) needs multiples of 32 and three image channels.  num_channels: 1 .. 6 -- one
 * or two channels run as the equivalent three-channel network (zero filters on the channels that do not exist); four to six
 * (RGBA ...) run as TWO groups of three channels through the motion kernels (a pixel's image channels live in one float4
 * beside its heat-map value per group: motion.hip) with `final` on the generic 7x7 kernel; sources and outputs keep the
 * reference's shapes [.,C,H,W] throughout; seven and more are refused.
 */
typedef struct eamm_config {
    int32_t num_channels;            /* 3 (1 .. 6 accepted)                            */
    int32_t num_kp;                  /* 10                                             */
    int32_t block_expansion;         /* 64                                             */
    int32_t max_features;            /* 512                                            */
    int32_t num_down_blocks;         /* 2                                              */
    int32_t num_bottleneck_blocks;   /* 6                                              */
    int32_t estimate_occlusion_map;  /* 0 / 1                                          */
    int32_t dm_block_expansion;      /* dense_motion_params.block_expansion            */
    int32_t dm_max_features;         /* dense_motion_params.max_features               */
    int32_t dm_num_blocks;           /* dense_motion_params.num_blocks                 */
    int32_t dm_inv_scale;            /* 1 / dense_motion_params.scale_factor (1, 2, 4) */
    float   kp_variance;             /* dense_motion_params.kp_variance, default 0.01  */
    int32_t height, width;           /* frame size (multiple of 2^max(down, log2 inv_scale + dm_num_blocks)) */
    int32_t max_frames;              /* frames per eamm_forward_frames call the workspace holds */
    int32_t max_sources;             /* source images cached at once (1 for a clip)    */
} eamm_config;

/* Optional outputs of a forward call; NULL pointers are skipped.  Shapes for n frames, frame
 * HxW, motion grid hxw = H/inv_scale x W/inv_scale, K = num_kp (reference generator.py:70-95). */
typedef struct eamm_outputs {
    float* prediction;       /* [n,C,H,W]      'prediction'      -- required (C = num_channels) */
    float* mask;             /* [n,K+1,h,w]    'mask'                                   */
    float* sparse_deformed;  /* [n,K+1,C,h,w]  'sparse_deformed'                        */
    float* occlusion_map;    /* [n,1,h,w]      'occlusion_map'                          */
    float* deformed;         /* [n,C,H,W]      'deformed'                               */
    float* deformation;      /* [n,h,w,2]      dense_motion 'deformation' (internal to the reference) */
    uint8_t* frames_u8;      /* [n,H,W,3] uint8 = round(255*prediction), HWC: the layout demo.py:281,507 saves (C = 3 only) */
} eamm_outputs;

int eamm_abi_version(void);

/* Replaces OcclusionAwareGenerator.__init__ (generator.py:14-48) + .cuda() (demo.py:56-57). */
int eamm_create(const eamm_config* cfg, int device, eamm_ctx** out);
void eamm_destroy(eamm_ctx* ctx);
const char* eamm_last_error(const eamm_ctx* ctx);

/*
 * Replaces generator.load_state_dict(checkpoint['generator']) (demo.py:91, logger.py:58): call
 * once per state_dict entry with the reference's key (e.g. "bottleneck.r0.conv1.weight") and a
 * HOST float32 pointer (num_batches_tracked entries may be skipped), then eamm_finalize_weights,
 * which checks the key set strictly, folds eval-mode BatchNorm (sync_batchnorm/batchnorm.py:48-53)
 * into the adjacent convolutions and repacks OIHW weights into MFMA tile order on the device.
 */
int eamm_load_tensor(eamm_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
int eamm_finalize_weights(eamm_ctx* ctx);

/*
 * Frame-invariant part of OcclusionAwareGenerator.forward, hoisted out of the per-frame loop
 * (generator.py:61-63 encoder, dense_motion.py:83 anti-alias down-sampling): caches, for `ns`
 * source images [ns,C,H,W] on the device, the encoder feature map, the down-sampled source and the
 * full-resolution source.
 */
int eamm_encode_source(eamm_ctx* ctx, const float* source, int ns, void* stream);

/*
 * Per-frame part of OcclusionAwareGenerator.forward (generator.py:68-95, dense_motion.py:81-113)
 * for n driving key-point sets.  kp_*_value: [*,K,2]; kp_*_jacobian: [*,K,2,2] or NULL (both or
 * neither; NULL = the "'jacobian' not in kp_driving" branch, dense_motion.py:55).  kp_source_* hold
 * `ns` sets matching the cached sources; ns == 1 broadcasts one source to all frames (the clip
 * loop, demo.py:251-281), otherwise ns == n and frame i uses source i (the module's batch contract).
 */
int eamm_forward_frames(eamm_ctx* ctx, int n,
                        const float* kp_driving_value, const float* kp_driving_jacobian,
                        const float* kp_source_value, const float* kp_source_jacobian,
                        const eamm_outputs* outputs, void* stream);

/*
 * torch.inverse raises on a singular key-point jacobian (dense_motion.py:56) and synchronises to do
 * so.  The kernels only record the condition; this call synchronises `stream` and returns
 * EAMM_ERR_NUMERIC if any forward call since the last check met one (outputs are then inf/nan).
 */
int eamm_check_numeric(eamm_ctx* ctx, void* stream);

/*
 * Cached source tensors as one flat device blob, so that a clip's source can be encoded on one
 * GPU and broadcast to the others (RCCL) instead of re-encoded: export copies the cache of the
 * `ns` encoded sources into `dst`, import installs such a blob.  Both are stream-ordered copies.
 */
size_t eamm_source_cache_bytes(const eamm_ctx* ctx, int ns);
int eamm_export_source_cache(eamm_ctx* ctx, void* dst, int ns, void* stream);
int eamm_import_source_cache(eamm_ctx* ctx, const void* src, int ns, void* stream);

/* Algorithmic work of one eamm_forward_frames call of n frames (for roofline accounting). */
double eamm_flops_per_frame(const eamm_ctx* ctx);
double eamm_encode_flops(const eamm_ctx* ctx);
/* Form the bottleneck ResBlock2d convolutions take for a call of n frames: 0 = direct, 2 = Winograd F(2x2,3x3),
 * 4 = Winograd F(4x4,3x3) (executed multiplies = 1, 4/9, 1/4 of the reference's; negative on a bad handle). */
int eamm_bottleneck_form(const eamm_ctx* ctx, int n);
/* Chains the bottleneck of a call of n frames is split into (1 = one launch sequence on the caller's stream; K > 1 =
 * K groups of whole frames on K streams forked from and joined back into the caller's stream inside the call). */
int eamm_bottleneck_chains(const eamm_ctx* ctx, int n);
/* Of those, the chains that cover the WHOLE per-frame pass (independent launch sequences whose bottleneck launches
 * time-share the chip) rather than the bottleneck alone: 1 = none. */
int eamm_pass_chains(const eamm_ctx* ctx, int n);
/* Which side streams the last eamm_forward_frames call ran its chains on: 0 = the device's shared pool, 1 = the handle's private
 * streams because another host thread held the pool, 2 = private because the caller's stream was being captured into a graph
 * (side streams join the capture until it ends, so a captured call never touches the shared pool). */
int eamm_last_stream_set(const eamm_ctx* ctx);
/* The launch plan of a call of n frames as a JSON object (chains, frames per chain, bottleneck form, hourglass level forms,
 * final-layer kernel ...), written NUL-terminated into buf[cap]; returns the length needed.  For benchmark records. */
int eamm_describe_plan(const eamm_ctx* ctx, int n, char* buf, int cap);
/* Every EAMM_* knob the library has read so far in this process, as a JSON object {"NAME": {"value": v, "set": 0|1|2}}
 * (value in effect; set = 1: the environment supplied it; 2: present in the environment but IGNORED).  Same buffer contract as
 * eamm_describe_plan.
 * DOCUMENTED knobs -- selectors of a computation form or of the launch plan; every setting computes the same frames up to rounding:
 *   EAMM_PASS_CHAINS / EAMM_BNECK_CHAINS   chains of a call over the whole pass / inside the bottleneck (0 = automatic, 1 = off)
 *   EAMM_WINO_TILE (4 | 2), EAMM_WINO_MIN_M (< 0: direct), EAMM_WINO4_MIN_M    bottleneck form: F(4x4), F(2x2), direct
 *   EAMM_ENC_WINO, EAMM_PATCH_POLY, EAMM_HEAD_ROWSPLIT, EAMM_COL7, EAMM_FIRST7, EAMM_FINAL_FUSED, EAMM_FINAL_MFMA4, EAMM_KPA_THIN
 *                                          (0 | 1) the alternative kernel of a stage: hourglass encoder, up blocks, flow head, 7x7 layers,
 *                                          KPDetector_a's heads (wide + thin split | one padded convolution)
 *   EAMM_WGRAD_WINO4, EAMM_WGRAD_ROW, EAMM_CONV_DEV_WINO4   (0 | 1) forms of the training operators
 *   EAMM_PRIVATE_STREAMS (0 | 1)           never use the shared side-stream pool
 *   EAMM_WARP_JOINT, EAMM_BNECK_STAGGER    scheduling variants measured in round 4 (off)
 *   EAMM_WINO4_VARIANT (0-6)               software pipeline of the F(4x4) bottleneck GEMM, pinned for every call size; unset: 6 (variant 3 with
 *                                          the transformed input loaded non-temporally) while the call's GEMM workgroups number at most the CUs,
 *                                          3 above (bit-identical results)
 * TUNING aids (tile-size thresholds, pipeline variants, split sizes: everything else the sources read) are honoured only together
 * with EAMM_TUNING=1; without it they are ignored and reported with "set": 2.
 * Knobs that compute WRONG results (timing experiments: EAMM_WINO4_EPI_V, EAMM_COL7_DBG, EAMM_WINO4_VARIANT 10/16/17/50) exist
 * only in a build with -DEAMM_EXPERIMENTS (make EXPERIMENTS=1); the product library's eamm_create fails when one of them is set. */
int eamm_knobs_json(char* buf, int cap);
/* Executed matrix-core flops (what the launched grids really issue: padded tiles, Winograd / polyphase point counts -- not the
 * reference convolution's) of every kernel enqueued through the library in this PROCESS so far, all host threads (autograd runs
 * the backward operators on its own thread): a monotonic total; difference two readings around a step (bench.py
 * `train_step.roofline`).  eamm_forward_frames keeps its own per-call count for the profile keys. */
double eamm_total_mfma_flops(void);
/* 1 when this library was built with -DEAMM_EXPERIMENTS (never benchmark or ship such a build), else 0. */
int eamm_build_experiments(void);

/*
 * ---- key-point detectors -----------------------------------------------
 * Constructor arguments of KPDetector / KPDetector_a (reference modules/keypoint_detector.py:12-14, 115-117).
 * with_predictor = 1: KPDetector -- eamm_kp_detect(image) runs anti-alias down-sampling, the `predictor`
 * hourglass and the heads (keypoint_detector.py:77-105; once per clip on the source, demo.py:206).
 * with_predictor = 0: KPDetector_a -- eamm_kp_detect_features(feature_map) runs only the heads on a caller-
 * provided [B, block_expansion + in_features, H/inv_scale, W/inv_scale] map (keypoint_detector.py:180-205;
 * once per frame, demo.py:219); `predictor.*` / `down.weight` entries need not be loaded for it.
 */
typedef struct eamm_kp_ctx eamm_kp_ctx;
typedef struct eamm_kp_config {
    int32_t num_kp;                 /* 10                                                   */
    int32_t num_channels;           /* 3 (1 .. 8 accepted)                                   */
    int32_t in_features;            /* hourglass input channels: num_channels (KPDetector) or num_channels_a */
    int32_t block_expansion;        /* 32                                                   */
    int32_t max_features;           /* 1024                                                 */
    int32_t num_blocks;             /* 5                                                    */
    float   temperature;            /* 0.1                                                  */
    int32_t estimate_jacobian;      /* 0 / 1                                                */
    int32_t single_jacobian_map;    /* 0 / 1                                                */
    int32_t inv_scale;              /* 1 / scale_factor (1, 2, 4)                           */
    int32_t pad;                    /* padding of the 7x7 heads: 0 (reference default) or 3 */
    int32_t height, width;          /* image size; feature maps are height/inv_scale x width/inv_scale */
    int32_t max_batch;
    int32_t with_predictor;
} eamm_kp_config;

typedef struct eamm_kp_outputs {    /* 'value' required; NULL pointers are skipped           */
    float* value;                   /* [B,K,2]     soft-argmax key points, (x, y) in [-1,1] */
    float* jacobian;                /* [B,K,2,2]                                            */
    float* heatmap;                 /* [B,K,h-6+2*pad,w-6+2*pad]                            */
} eamm_kp_outputs;

int eamm_kp_create(const eamm_kp_config* cfg, int device, eamm_kp_ctx** out);
void eamm_kp_destroy(eamm_kp_ctx* ctx);
const char* eamm_kp_last_error(const eamm_kp_ctx* ctx);
int eamm_kp_load_tensor(eamm_kp_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
int eamm_kp_finalize_weights(eamm_kp_ctx* ctx);
int eamm_kp_detect(eamm_kp_ctx* ctx, const float* image /*[B,3,H,W]*/, int B, const eamm_kp_outputs* out, void* stream);
int eamm_kp_detect_features(eamm_kp_ctx* ctx, const float* feature_map, int B, const eamm_kp_outputs* out, void* stream);
/* Round 6 -- the private hand-over between the deconvolution tail and KPDetector_a (reference: util.py:604-607 produces
 * deco_out[:, t], keypoint_detector.py:180-205 consumes it; both sides live in this library, so the [B,35,64,64] NCHW tensor the
 * reference materialises between them is optional).  eamm_kp_split_channels: the `wide` channel count 32 m this handle's heads
 * read (feature map = 32 m + 3 channels, heads in the wide + thin form), or 0.  eamm_kp_detect_features_split: `wide` NHWC
 * [B,h,w,32 m] float32 + `thin` [B,h,w,4] (the last three channels and a zero), both caller-owned device memory, as written by
 * eamm_deconv_forward_split; same outputs as eamm_kp_detect_features on the equivalent NCHW map, bit for bit. */
int eamm_kp_split_channels(const eamm_kp_ctx* ctx);
int eamm_kp_detect_features_split(eamm_kp_ctx* ctx, const float* wide, const float* thin, int B, const eamm_kp_outputs* out, void* stream);

/*
 * ---- N3: audio-to-feature deconvolution tail ------------------------------------------------------------
 * Replaces the nn.Sequential `AT_net2.decon` (reference modules/util.py:559-576), which the reference evaluates
 * once per frame on the LSTM output (util.py:604-607) to produce the [35,64,64] maps KPDetector_a consumes
 * (demo.py:219): num_layers ConvTranspose2d layers -- the first 6x6 / stride 2 / padding 1 on the 1x1 input, the
 * rest 4x4 / stride 2 / padding 1 -- each but the last followed by eval-mode BatchNorm2d + ReLU.
 * State-dict keys are the Sequential's: "0.weight" [C0,C1,6,6], "0.bias", "1.weight", "1.bias", "1.running_mean",
 * "1.running_var", "3.weight" [C1,C2,4,4], ... (strict check in eamm_deconv_finalize_weights).
 */
typedef struct eamm_deconv_ctx eamm_deconv_ctx;

typedef struct eamm_deconv_config {
    int32_t num_layers;     /* reference: 5                                                          */
    int32_t channels[9];    /* [0] input features (256); [i+1] outputs of layer i (256,128,128,128,35);
                             * every layer INPUT width must be a multiple of 32                      */
    int32_t max_batch;      /* frames per eamm_deconv_forward call                                   */
} eamm_deconv_config;

int eamm_deconv_create(const eamm_deconv_config* cfg, int device, eamm_deconv_ctx** out);
void eamm_deconv_destroy(eamm_deconv_ctx* ctx);
const char* eamm_deconv_last_error(const eamm_deconv_ctx* ctx);
int eamm_deconv_load_tensor(eamm_deconv_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
int eamm_deconv_finalize_weights(eamm_deconv_ctx* ctx);
/* x: [B, channels[0]] (= the reference's [B,C,1,1]) device fp32; out: [B, channels[L], S, S] NCHW, S = 4 << (L-1). */
int eamm_deconv_forward(eamm_deconv_ctx* ctx, const float* x, int B, float* out, void* stream);
/* Round 6: the same layers with the last one writing the split NHWC form KPDetector_a's heads read (above) instead of NCHW:
 * wide [B,S,S,C_last - 3], thin [B,S,S,4].  eamm_deconv_split_channels: C_last - 3 when that is a positive multiple of 32, else 0
 * (eamm_deconv_forward_split then fails with EAMM_ERR_ARG). */
int eamm_deconv_split_channels(const eamm_deconv_ctx* ctx);
int eamm_deconv_forward_split(eamm_deconv_ctx* ctx, const float* x, int B, float* wide, float* thin, void* stream);

/*
 * ---- N4: training-mode BatchNorm, forward (round 2) and backward (round 3) --------------------------------------
 * Replaces the arithmetic of _SynchronizedBatchNorm.forward (reference sync_batchnorm/batchnorm.py:46-125) on NCHW
 * float32 device tensors; the reduction over replicas (ReduceAddCoalesced / Broadcast, batchnorm.py:102-105) is the
 * caller's all-reduce of `sums` between eamm_bn_local_sums and eamm_bn_finalize (torch.distributed / RCCL).
 * Stateless: the caller owns every buffer and makes the tensors' device current; work is enqueued on `stream`.
 *   eamm_bn_local_sums   sums[c] = sum over (N, HW) of x[:, c], sums[C + c] = sum of squares (batchnorm.py:61-64);
 *                        sums[2C] + 4096 * sums[2C+1] = N * HW as two exact floats, so that element counts add up
 *                        through the same float all-reduce -- all-reduce THESE 2C+2 floats.  `sums` holds 6C+2 floats:
 *                        behind the exchanged part sit the same 2C totals in double (8-byte aligned), which
 *                        EAMM_BN_SINGLE uses.  workspace: eamm_bn_workspace_floats(N, C, HW) floats, 8-byte aligned.
 *   eamm_bn_finalize     mean[c], scale[c] = inv_std * weight[c] and the running-statistics update (momentum, unbiased
 *                        variance) of _compute_mean_std (batchnorm.py:110-125).  mode EAMM_BN_SYNC: the replicas' path,
 *                        inv_std = clamp(biased var, eps)^-0.5; EAMM_BN_SINGLE: F.batch_norm(training=True)
 *                        (batchnorm.py:48-53), inv_std = 1/sqrt(biased var + eps); EAMM_BN_EVAL: running statistics,
 *                        nothing updated (sums may be NULL).  weight may be NULL (affine=False).
 *                        inv_std (optional, [C]): the inverse standard deviation alone -- what the backward needs.
 *   eamm_bn_apply        y = (x - mean[c]) * scale[c] + bias[c] (batchnorm.py:74-79); bias may be NULL.
 * Backward (round 3) -- the gradient autograd derives from those lines, with xhat = (x - mean) * inv_std:
 *   eamm_bn_backward_sums      sums[c] = sum dy, sums[C + c] = sum dy * (x - mean[c]) over this replica's shard, packed like
 *                              eamm_bn_local_sums' output (6C+2 floats): all-reduce the first 2C+2 floats INTO A COPY --
 *   eamm_bn_backward_finalize  takes both: dbias = local sum dy, dweight = local sum dy * xhat (this replica's shard; the
 *                              reduction of parameter gradients over replicas is the caller's, as under DistributedDataParallel),
 *                              and coef[3C] = (S1 / N, S2 * inv_std^2 / N, weight * inv_std) from the REDUCED sums (all replicas);
 *                              mode as eamm_bn_finalize (EAMM_BN_EVAL: statistics are constants, S-terms vanish; EAMM_BN_SYNC: a
 *                              channel whose variance was clamped to eps has no gradient through the variance).
 *   eamm_bn_backward_apply     dx = coef_s * (dy - coef_a - (x - mean) * coef_b).
 */
#define EAMM_BN_SYNC 0
#define EAMM_BN_SINGLE 1
#define EAMM_BN_EVAL 2
size_t eamm_bn_workspace_floats(int N, int C, int HW);
int eamm_bn_local_sums(const float* x, int N, int C, int HW, float* sums /*[6C+2]*/, float* workspace, void* stream);
int eamm_bn_finalize(const float* sums, int C, float eps, float momentum, int mode, const float* weight,
                     float* running_mean, float* running_var, float* mean /*[C]*/, float* scale /*[C]*/, float* inv_std /*[C] or NULL*/,
                     void* stream);
int eamm_bn_backward_sums(const float* x, const float* dy, const float* mean, int N, int C, int HW, float* sums /*[6C+2]*/,
                          float* workspace, void* stream);
int eamm_bn_backward_finalize(const float* local_sums, const float* reduced_sums, int C, const float* inv_std, const float* weight,
                              float eps, int mode, float* dweight /*[C] or NULL*/, float* dbias /*[C] or NULL*/, float* coef /*[3C]*/,
                              void* stream);
int eamm_bn_backward_apply(const float* x, const float* dy, const float* mean, const float* coef, int N, int C, int HW, float* dx,
                           void* stream);

/* The same BatchNorm on NHWC activations x [B,H,W,C] (= [M,C], M = B*H*W; what the path's convolutions produce), fused with the
 * tail of the reference's blocks (modules/util.py:858-938): y = [avgpool2x2](act((x - mean) * scale + bias)), relu = 1 | 0,
 * pool = 1 | 0 (DownBlock2d; H, W even; y is [B,H/2,W/2,C]).  Statistics / finalize / backward_finalize are the layout-free
 * entries above (same packed `sums`).  Backward: grad_out has y's shape; the ReLU mask is recomputed from x, so x is all the
 * forward keeps.  C a multiple of 4, at most 1024. */
size_t eamm_bn_nhwc_workspace_floats(long long M, int C);
int eamm_bn_nhwc_local_sums(const float* x, long long M, int C, float* sums, float* workspace, void* stream);
int eamm_bn_nhwc_apply(const float* x, const float* mean, const float* scale, const float* bias, int B, int H, int W, int C, int relu,
                       int pool, float* y, void* stream);
int eamm_bn_nhwc_backward_sums(const float* x, const float* grad_out, const float* mean, const float* scale, const float* bias, int B,
                               int H, int W, int C, int relu, int pool, float* sums, float* workspace, void* stream);
int eamm_bn_nhwc_backward_apply(const float* x, const float* grad_out, const float* mean, const float* scale, const float* bias,
                                const float* coef, int B, int H, int W, int C, int relu, int pool, float* grad_x, void* stream);
/* One replica (nothing to all-reduce between the sums and their finalize step): eamm_bn_nhwc_local_sums + eamm_bn_finalize, and
 * eamm_bn_nhwc_backward_sums + eamm_bn_backward_finalize (reduced_sums == local_sums), each as ONE call -- the finalize step runs
 * in the kernel that adds the slices up, on the values the separate entry would read back (bit-identical results, two launches
 * fewer per BatchNorm site and direction).  mode 0 | 1 as eamm_bn_finalize; `sums` is still written. */
int eamm_bn_nhwc_local_stats(const float* x, long long M, int C, float eps, float momentum, int mode, const float* weight,
                             float* running_mean, float* running_var, float* sums, float* mean, float* scale, float* inv_std,
                             float* workspace, void* stream);
int eamm_bn_nhwc_backward_local(const float* x, const float* grad_out, const float* mean, const float* scale, const float* bias, int B,
                                int H, int W, int C, int relu, int pool, const float* inv_std, const float* weight, float eps, int mode,
                                float* sums, float* grad_weight /*[C] or NULL*/, float* grad_bias /*[C] or NULL*/, float* coef /*[3C]*/,
                                float* workspace, void* stream);
int eamm_bn_apply(const float* x, const float* mean, const float* scale, const float* bias, int N, int C, int HW, float* y,
                  void* stream);
const char* eamm_bn_last_error(void);

/*
 * ---- N4, second slice (round 3): the generator's forward in TRAINING mode ------------------------------------------------
 * Replaces OcclusionAwareGenerator.forward when the module is in .train() (the fine-tuning loop, reference train.py:133):
 * every SynchronizedBatchNorm2d inside SameBlock2d / DownBlock2d / UpBlock2d / ResBlock2d (modules/util.py:858-938) then
 * normalises with the statistics of the batch (sync_batchnorm/batchnorm.py:55-125) and updates its running statistics.
 *   eamm_set_training(ctx, 1)  before eamm_finalize_weights: convolutions are packed with their RAW weights (no BatchNorm
 *                              folded in); eamm_encode_source / eamm_forward_frames of such a handle would run without
 *                              any normalisation and must not be used.
 *   eamm_train_num_sites / eamm_train_site_name   the BatchNorm sites of one forward in execution order, named by their
 *                              state_dict prefix ("first.norm", "bottleneck.r0.norm1", ...).
 *   eamm_train_begin           lays out one forward over n (source, key point) pairs -- source [n,3,H,W]; key points as for
 *                              eamm_forward_frames with one source set per frame -- and the caller's BatchNorm tensors
 *                              (device pointers, one eamm_bn_site per site: running_mean / running_var are UPDATED in place
 *                              with `momentum`, batchnorm.py:110-125).  sync = 0: F.batch_norm's formula (one replica,
 *                              batchnorm.py:48-53), 1: the replicas' formula.  sums_buffer: caller-owned device buffer of
 *                              6 * eamm_train_max_channels(ctx) + 2 floats, 8-byte aligned, that stays valid until the
 *                              last eamm_train_next.  Nothing is enqueued yet.
 *   eamm_train_next            enqueues the forward up to the next BatchNorm's statistics and returns 1 with
 *                              *nfloats = 2C + 2: the first 2C + 2 floats of sums_buffer hold per-channel sum, sum of
 *                              squares and the element count as in eamm_bn_local_sums -- ALL-REDUCE them over the replicas on the
 *                              same stream (nothing to do on one replica) and call again; returns 0 when the outputs given
 *                              to eamm_train_begin are complete (stream-ordered), < 0 on error.
 * Forward only: the outputs carry no gradient (the convolution and warp backward kernels are not built).
 */
typedef struct eamm_bn_site {
    const float* weight;     /* [C] device                                  */
    const float* bias;       /* [C] device                                  */
    float* running_mean;     /* [C] device, updated                         */
    float* running_var;      /* [C] device, updated (unbiased batch variance) */
} eamm_bn_site;
int eamm_set_training(eamm_ctx* ctx, int on);
int eamm_train_num_sites(const eamm_ctx* ctx);
const char* eamm_train_site_name(const eamm_ctx* ctx, int i);
int eamm_train_max_channels(const eamm_ctx* ctx);
int eamm_train_begin(eamm_ctx* ctx, const float* source, int n, const float* kp_driving_value, const float* kp_driving_jacobian,
                     const float* kp_source_value, const float* kp_source_jacobian, const eamm_bn_site* sites, int nsites,
                     float momentum, float eps, int sync, float* sums_buffer, const eamm_outputs* outputs, void* stream);
int eamm_train_next(eamm_ctx* ctx, int* nfloats);

/*
 * Stage timing for roofline accounting (bench.py): while enabled, every eamm_forward_frames call
 * records HIP events on the caller's stream at its stage boundaries and around every bottleneck launch
 * (up to 256 calls between reads).  eamm_profile_read waits for the recorded calls and returns
 * accumulated milliseconds per stage: 0 key points + motion front end, 1 hourglass encoder, 2 hourglass
 * decoder, 3 flow head, 4 feature warp, 5 bottleneck Winograd input transforms (0 in the direct form),
 * 6 rest of the bottleneck stage (the convolution kernels; with several chains: the stage's wall time minus the main
 * stream's transforms), 7 up blocks, 8 final 7x7 + sigmoid (+ uint8 packing) -- stages 0..8 add up to the call --
 * and 9 the bottleneck GEMM kernels' own durations on the main stream (2 x num_bottleneck_blocks launches per call;
 * with K chains each of those launches covers 1/K of the frames and runs beside the other chains' launches).
 * Chip-level accounting (all chains, not only the main stream): 10 the wall time during which ANY whole-pass chain is
 * inside its bottleneck stage (union of the chains' windows, events on every chain's stream), 11 the sum of those
 * windows, 12 executed matrix-core GFLOP of the recorded calls (what the grids really issue: padded tiles, Winograd /
 * polyphase point counts), 13 the bottleneck GEMMs' share of 12, 14..21 the same count per stage interval (key points +
 * front end, hourglass encoder, hourglass decoder, flow head, warp, bottleneck, up blocks, final layer) so that every stage's
 * fraction of the matrix peak can be printed, not only the bottleneck's.  (12..21 are GFLOP, not milliseconds.)
 */
#define EAMM_NSTAGE 22
int eamm_profile_enable(eamm_ctx* ctx, int on);
int eamm_profile_read(eamm_ctx* ctx, double* stage_ms, int nstage, int64_t* calls, int64_t* frames, int reset);

/*
 * Op-level entry points used by the parity tests (tests/test_gpu_ops.py): each runs ONE kernel of
 * the path on caller-provided device buffers so it can be compared with the oracle in isolation.
 */

/* KHxKW (3x3, 7x7 or 7x1) convolution on NHWC activations as an fp32-MFMA implicit GEMM.
 * in0/in1: [B,Hin,Win,C0|C1] (in1 optional: channel concatenation), weight: OIHW host pointer
 * [Cout,C0+C1,kh,kw], bias host [Cout]; up = 1 (3x3 only) computes conv3x3(nearest_x2(input)) in its
 * collapsed four-phase 2x2 form; act: 0 none, 1 relu, 2 sigmoid; pool = 1 applies avgpool2x2 after the
 * activation; resid: NHWC tensor added before the activation; splitk 0 = automatic; tile_n 0 =
 * automatic, 32/64/128 = register-staged 128 x tile_n kernel, 1000+id = LDS-DMA big-tile kernel (1: 256x256,
 * 2: 256x128, 3: 512x64), 4002 = the generator's final layer whole (7x7, Cout 3, act 2: out is NCHW [B,3,H,W]), 2000 = Winograd F(2x2,3x3) (input transform + GEMM; 3x3, single input, no pool); out NHWC [B,H(/2),W(/2),Cout] with H = Hin << up.  iters > 0 additionally times `iters`
 * back-to-back launches with HIP events on `stream` and stores the average milliseconds in *avg_ms. */
int eamm_op_conv(int device, const float* in0, int C0, const float* in1, int C1, int B, int Hin, int Win, int up,
                 const float* weight_host, const float* bias_host, int Cout, int kh, int kw,
                 int act, int pool, const float* resid, int splitk, int tile_n, float* out, int iters,
                 float* avg_ms, void* stream);

/* Temporal smoothing of a clip's key points: the reference's filter1.OneEuroFilter (filter1.py:13-47)
 * applied as `process(x * scale) / scale` frame after frame (demo.py:241-250; key points: mincutoff 0.05, beta 8, dcutoff 1,
 * freq 100, scale 10; emotion displacements demo.py:231-239: 1, 0.2, 1, 100, 100).  x, out: device [T,E] float32, the filter
 * runs along T independently per element (E = K*2 values or K*4 jacobian entries; out may alias x).  `state` (device [3,E] or
 * NULL) receives the filter's memory after the last frame; with `resume` != 0 it is read first, so a clip filtered in chunks gives
 * exactly the clip filtered whole (the filter is causal: the streaming clip pipeline).  Stream-ordered, no sync. */
int eamm_op_one_euro(int device, const float* x, int T, int E, float mincutoff, float beta, float dcutoff, float freq, float scale,
                     float* out, float* state, int resume, void* stream);

/* The HBM-bound kernel of the path on its own: out = grid_sample(feat, deformation, bilinear, zeros, align_corners=False)
 * * occlusion (reference modules/generator.py:50-57, 79-84).  feat: NHWC [ns,hf,wf,C] (ns = 1 broadcasts one source to
 * all frames, else ns = n), deformation [n,h,w,2], occlusion [n,h,w] or NULL, out NHWC [n,hf,wf,C]; when (h,w) != (hf,wf)
 * flow and occlusion are resized bilinearly first (generator.py:52-56, 82-83).  iters > 0 times `iters` back-to-back
 * launches with HIP events on `stream` (bench.py's isolated roofline_warp figure). */
int eamm_op_warp(int device, const float* feat, const float* deformation, const float* occlusion, int n, int ns, int hf,
                 int wf, int C, int h, int w, float* out, int iters, float* avg_ms, void* stream);

/*
 * Backward kernels of the path's two operator kinds, op level.  The reference gets these from
 * autograd (train.py:133 `loss.backward()` through modules/generator.py and modules/util.py); the BatchNorm backward is
 * eamm_bn_backward_* above.
 */

/* Gradient of eamm_op_warp at (h,w) == (hf,wf): grad_out NHWC [n,hf,wf,C] -> grad_feat [ns,hf,wf,C] (ns = 1: summed over the
 * frames), grad_deformation [n,hf,wf,2], grad_occlusion [n,hf,wf]; any of the three may be NULL (not wanted), occlusion NULL
 * means the forward had none.  What autograd derives for F.grid_sample(bilinear, zeros, align_corners=False) * occlusion
 * (generator.py:50-57, 79-84).  The entry zeroes the gradients on `stream`, then accumulates with float atomics (grad_feat is
 * therefore equal to the fixed-order sum up to fp32 rounding of the order, as ATen's own grid_sampler backward is). */
int eamm_op_warp_backward(int device, const float* feat, const float* deformation, const float* occlusion, const float* grad_out,
                          int n, int ns, int hf, int wf, int C, float* grad_feat, float* grad_deformation, float* grad_occlusion,
                          void* stream);

/* Weight and bias gradient of a stride-1 "same" KHxKW convolution (every Conv2d of modules/util.py:858-938): x NHWC
 * [B,H,W,Cin], grad_out NHWC [B,H,W,Cout] -> grad_weight OIHW [Cout,Cin,kh,kw] (device), grad_bias [Cout] (device, or NULL).
 * fp32-MFMA GEMMs with K = the B*H*W pixels split over workgroups, partial sums in `workspace`
 * (eamm_op_conv_wgrad_workspace_floats floats for this shape) and summed in a fixed order: deterministic.  Forms: Winograd
 * F(3x3,4x4) for 3x3 filters on maps with sides multiples of 4 and at least 512 tiles (36 GEMMs over the tiles on the forward's
 * transformed input: a quarter of the multiplies), one GEMM per filter ROW for maps with W % 32 == 0, else one per tap.
 * Cin, Cout multiples of 4, Cout <= 1024; kh, kw odd and at most 7.  The DATA gradient of such a convolution is eamm_op_conv on grad_out with the filter transposed and
 * flipped (weight.permute(1,0,2,3).flip(2,3)), which eamm_amd/autograd_ops.py does. */
/* Stride-1 "same" 3x3 / 7x7 convolution with DEVICE parameters (the training path: they change every optimiser step):
 * x NHWC [B,H,W,Cin], weight OIHW [Cout,Cin,kh,kw] and bias [Cout] (or NULL) in device memory, out NHWC [B,H,W,Cout].  The filter
 * is packed on the device into `workspace` (eamm_op_conv_dev_workspace_floats floats, 16-byte aligned) and the evaluation
 * path's kernels run on it: F(4x4,3x3) Winograd where the bottleneck's kernel applies (3x3, Cin % 64 == 0, H and W multiples of
 * 4, >= 2048 tiles), else the register-staged implicit GEMM.  transposed = 1 computes the DATA GRADIENT of such a convolution:
 * x is then grad_out [B,H,W,Cin = the forward's Cout], weight the FORWARD filter [Cin,Cout,kh,kw], read transposed over
 * (out, in) and flipped over (y, x) while packing.  Asynchronous on `stream`; Cin a multiple of 32, Cout of 4. */
size_t eamm_op_conv_dev_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw);
int eamm_op_conv_dev(int device, const float* x, int B, int H, int W, int Cin, const float* weight, const float* bias, int Cout, int kh,
                     int kw, int transposed, float* out, float* workspace, size_t workspace_floats, void* stream);

/* The generator's two 7x7 layers with THREE channels on one side (`first` 3 -> N, `final` N -> 3: reference
 * modules/generator.py:26, 48), training path, without padding the thin side to 32 channels.  Thin tensors are NHWC with FOUR
 * floats per pixel, [B,H,W,4], the fourth zero; N = 32 | 64; device pointers; `workspace` of eamm_op_conv7_thin_workspace_floats.
 *   eamm_op_conv7_thin        transposed = 0: out [B,H,W,N] = conv7x7_same(thin, weight [N,3,7,7]) + bias   (forward of `first`)
 *                             transposed = 1: out [B,H,W,N] = data gradient of y3 = conv7x7_same(x [.,N], weight [3,N,7,7]) from
 *                                             thin = d y3                                                  (backward of `final`)
 *   eamm_op_conv7_thin_wgrad  thin_is_input = 1: grad_weight [N,3,7,7] from thin = the layer's input, wide = d out [B,H,W,N]
 *                             thin_is_input = 0: grad_weight [3,N,7,7] from thin = d y3, wide = the layer's input [B,H,W,N]
 * (the forward of `final` is the evaluation path's column-patch kernel: eamm_op_conv tile 4002; bias gradients: eamm_op_conv_wgrad's
 * or a plain sum). */
size_t eamm_op_conv7_thin_workspace_floats(int B, int H, int W, int N);
int eamm_op_conv7_thin(int device, const float* thin, const float* weight, const float* bias, int B, int H, int W, int N, int transposed,
                       float* out, float* workspace, size_t workspace_floats, void* stream);
int eamm_op_conv7_thin_wgrad(int device, const float* thin, const float* wide, int B, int H, int W, int N, int thin_is_input,
                             float* grad_weight, float* workspace, size_t workspace_floats, void* stream);

/* The generator's last layer whole with DEVICE parameters (training path): out NCHW [B,3,H,W] = sigmoid(conv7x7_same(x [B,H,W,C],
 * weight [3,C,7,7]) + bias [3]) -- reference modules/generator.py:92-93 -- on the evaluation path's fused column-patch kernel;
 * the filter is packed on the device into `workspace` (7 * C * 32 floats, 16-byte aligned).  C = 32 | 64. */
int eamm_op_final_conv_sigmoid(int device, const float* x, const float* weight, const float* bias, int B, int H, int W, int C,
                               float* out_nchw, float* workspace, size_t workspace_floats, void* stream);

size_t eamm_op_conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw);
int eamm_op_conv_wgrad(int device, const float* x, const float* grad_out, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                       float* grad_weight, float* grad_bias, float* workspace, size_t workspace_floats, void* stream);
/* Where eamm_op_conv_dev ran the F(4x4,3x3) form AND the weight gradient takes its F(3x3,4x4) form, both start from the same
 * transformed input V = B^T x B: a caller that keeps the forward call's workspace alive until the backward (autograd's saved
 * tensors) passes workspace + eamm_op_conv_saved_transform_offset(...) as `x_transformed` and the transform is not run again
 * (2.25 x the activation kept instead of 1 x; the same sums in the same order: bit-identical to eamm_op_conv_wgrad).
 * The offset is (size_t)-1 for a shape without a shared transform.  Workspace: eamm_op_conv_wgrad_workspace_floats. */
size_t eamm_op_conv_saved_transform_offset(int B, int H, int W, int Cin, int Cout, int kh, int kw);
int eamm_op_conv_wgrad_saved(int device, const float* x_transformed, const float* grad_out, int B, int H, int W, int Cin, int Cout,
                             float* grad_weight, float* grad_bias, float* workspace, size_t workspace_floats, void* stream);

/*
 * The dense-motion front end and flow head as DIFFERENTIABLE operators (round 4): forward = the evaluation
 * path's kernels (motion.hip), backward = what autograd derives from the reference's torch ops (motion_backward.hip).  Device
 * pointers, per-pair sources (n sources for n driving frames, the training batch of train.py:133); asynchronous on `stream`.
 *   eamm_op_antialias_down            AntiAliasInterpolation2d (modules/util.py:1044-1052): source NCHW [B,3,H,W], the module's
 *                                     [3,1,13,13] buffer -> small NHWC [B,H/s,W/s,4] (RGB + a zero); inv_scale 4 (scale_factor 0.25)
 *                                     or 1 (a copy, util.py:1047-1048)
 *   eamm_op_antialias_down_backward   grad_small [B,h,w,4] -> grad_source [B,3,H,W]
 *   eamm_op_kp_records                records [n,K,8] = kp_driving.xy, kp_source.xy, J = jac_source inverse(jac_driving)
 *                                     (dense_motion.py:47-67; jacobians NULL: identity); *singular_flag (device int) is set when
 *                                     a driving jacobian is singular (torch.inverse raises)
 *   eamm_op_kp_records_backward       grad_records -> grad of the four key-point tensors (any may be NULL)
 *   eamm_op_motion_front              heat-maps (util.py:815-836, dense_motion.py:32-45), sparse motions and the K+1 warps of
 *                                     `small` (:69-79) -> hourglass_in NHWC [n,h,w,Cpad] (channel 4k heat-map k, 4k+1..3 RGB warped
 *                                     by motion k, zero padded; dense_motion.py:93-94) and sparse_deformed [n,K+1,3,h,w] (or NULL)
 *   eamm_op_motion_front_backward     grad of both outputs (either may be NULL) -> grad_small [n,h,w,4] (or NULL; float atomics like
 *                                     ATen's grid_sampler backward) and grad_records [n,K,8] (fixed-order sums: deterministic)
 *   eamm_op_motion_head               mask = softmax over the K+1 mask logits [n,h,w,ld], deformation = sum_k mask_k T_k,
 *                                     occlusion = sigmoid(logit [n,h,w,ldo] channel 0) (dense_motion.py:98-111; occlusion NULL: none)
 *   eamm_op_motion_head_backward      grad_mask [n,K+1,h,w], grad_deformation [n,h,w,2], grad_occlusion [n,h,w] (any may be NULL)
 *                                     -> grad of the logits (same layouts, unused channels zeroed) and grad_records.
 *                                     Stacked logits (ONE convolution producing mask and occlusion: occlusion_logits =
 *                                     mask_logits + K + 1, ldo = ld; the same for the two gradient pointers) are recognised: the
 *                                     occlusion logit's gradient then lands in channel K + 1 of the one gradient tensor
 * `workspace`: eamm_op_motion_workspace_floats(n, K, h, w) floats.
 */
int eamm_op_antialias_down(int device, const float* source, const float* aa_weight, int B, int H, int W, int inv_scale, float* small,
                           void* stream);
int eamm_op_antialias_down_backward(int device, const float* grad_small, const float* aa_weight, int B, int H, int W, int inv_scale,
                                    float* grad_source, void* stream);
int eamm_op_kp_records(int device, const float* kd_val, const float* kd_jac, const float* ks_val, const float* ks_jac, int n, int K,
                       float* records, int* singular_flag, void* stream);
int eamm_op_kp_records_backward(int device, const float* kd_jac, const float* ks_jac, const float* grad_records, int n, int K,
                                float* grad_kd_val, float* grad_ks_val, float* grad_kd_jac, float* grad_ks_jac, void* stream);
size_t eamm_op_motion_workspace_floats(int n, int K, int h, int w);
int eamm_op_motion_front(int device, const float* records, const float* small, int n, int K, int h, int w, float kp_variance, int Cpad,
                         float* hourglass_in, float* sparse_deformed, void* stream);
int eamm_op_motion_front_backward(int device, const float* records, const float* small, int n, int K, int h, int w, float kp_variance,
                                  int Cpad, const float* grad_hourglass_in, const float* grad_sparse_deformed, float* grad_small,
                                  float* grad_records, float* workspace, size_t workspace_floats, void* stream);
int eamm_op_motion_head(int device, const float* mask_logits, int ld, const float* occlusion_logits, int ldo, const float* records, int n,
                        int K, int h, int w, float* mask, float* deformation, float* occlusion, void* stream);
int eamm_op_motion_head_backward(int device, const float* mask, const float* occlusion, const float* records, int n, int K, int h, int w,
                                 const float* grad_mask, const float* grad_deformation, const float* grad_occlusion,
                                 float* grad_mask_logits, int ld, float* grad_occlusion_logits, int ldo, float* grad_records,
                                 float* workspace, size_t workspace_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAMM_HIP_H_ */
