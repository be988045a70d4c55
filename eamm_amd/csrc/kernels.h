// Internal (non-ABI) declarations shared by the HIP translation units of libeamm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace eamm {

constexpr int CONV_BK = 32;           // K-chunk: 32 input channels of one filter tap
constexpr int CONV_LDK = CONV_BK + 4; // LDS row stride (dwords): +16 B keeps ds_read_b128 conflict-free
constexpr int KP_STRIDE = 8;          // per (frame, keypoint) record: kd.xy, ks.xy, J00 J01 J10 J11

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

// Executed-MFMA accounting for bench.py's roofline: every launcher of a matrix-core kernel adds the multiply-adds its grid
// really issues (padded tile dimensions, Winograd / polyphase point counts -- not the reference convolution's) x 2 to a
// per-host-thread counter; eamm_forward_frames folds it into the profile totals while profiling is on.
void note_mfma_flops(double flops);
double take_mfma_flops();   // returns the counter and resets it
double total_mfma_flops();  // process-wide monotonic total (all host threads)

// Tuning knobs: every EAMM_* environment variable the library reads goes through knob_int(), which records the name, the value
// in effect and whether the environment set it -- eamm_knobs_json() reports the table, so a benchmark line can carry the
// configuration it was measured under (bench.py `knobs`).  DOCUMENTED knobs are listed in include/eamm_hip.h.
long long knob_int(const char* name, long long dflt);
int knobs_json(char* buf, int cap);   // {"EAMM_X": {"value": v, "set": 0|1}, ...}; returns the length needed (excl. NUL)

// One convolution launch. Activations are NHWC fp32; the GEMM view is
//   M = B*H*W pixels (2x2-quad order), N = Cout, K = taps * (C0 + C1).
struct ConvArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] second half of a channel concatenation, or null
    int C0, C1;            // multiples of CONV_BK
    unsigned in0_bytes, in1_bytes, w_bytes;  // buffer-descriptor ranges
    int H, W;              // input size (even); phase mode writes a 2H x 2W output
    int linear;            // 1: pixels enumerated in raster order (odd H or W; never with pool) instead of 2x2 quads
    int nphase;            // 1, or 4 = collapsed "nearest x2 + 3x3" (one 2x2 filter per output parity)
    int M;                 // B*H*W
    const float* w;        // packed [nphase][ntiles][nchunks][BN][BK], BatchNorm already folded in
    const float* bias;     // [ntiles*BN]
    int Cout;              // real output channels (= output stride)
    int nchunks;           // taps * (C0+C1)/BK
    int mtiles, ntiles;
    int chunks_per_split;  // K range of one split-K slice
    int Mpad, Npad;        // partial slab dims
    float* partial;        // non-null: split-K, raw accumulators to [split][phase][Mpad][Npad]
    int act;               // Act
    int pool;              // avgpool2x2 after the activation
    int nchw;              // write out as [B,Cout,H,W] instead of NHWC
    const float* resid;    // NHWC, added before the activation
    float* out;            // NHWC [B,H,W,Cout] (pooled: [B,H/2,W/2,Cout]; phase: [B,2H,2W,Cout])
    float* out2;           // optional second output relu(out*s2 + t2): the next res-block's pre-activation
    const float* s2;
    const float* t2;
    int split_n;           // > 0 (round 6, the private DeconvTail -> KPDetector_a hand-over): channels n < split_n go to `out` as NHWC with
                           // a split_n-float pixel stride, channels split_n <= n < Cout to `out2` as one float4 per pixel (Cout <= split_n + 4)
};

// Packed convolution weights resident on the device.
struct ConvLayer {
    int kh = 0, kw = 0, C0 = 0, C1 = 0, Cout = 0;
    bool phase = false;     // UpBlock2d: nearest x2 + 3x3 as four 2x2 phase filters
    int BM = 128, BN = 0, ntiles = 0, nchunks = 0;
    int dma_cfg = 0;        // 0: register-staged 128 x BN kernel; >0: LDS-DMA big-tile kernel (conv_mfma_dma.hip)
    float* w = nullptr;     // device
    float* bias = nullptr;  // device, [ntiles*BN]
};

int conv_tile_n(int Cout);
size_t conv_packed_elems(int taps, int cin_packed, int Cout, int BN, int nphase);
// Host-side repack: [Cout][Cin][kh*kw] (BatchNorm folded by the caller) -> [nphase][ntiles][nchunks][BN][BK].
// cin_map[c] = original input channel of packed channel c, or -1 for zero padding.  phase = true takes
// 3x3 weights and emits the four collapsed 2x2 filters.
void conv_pack_host(const float* w, int Cout, int Cin, int kh, int kw, const int* cin_map, int cin_packed, int BN,
                    bool phase, bool swizzle, float* dst);
// Same packing from four explicit 2x2 phase filters wph [4][Cout][Cin][4] (ConvTranspose2d k4 s2 p1, conv_mfma.hip).
void conv_pack_phases_host(const float* wph, int Cout, int Cin, const int* cin_map, int cin_packed, int BN, bool swizzle,
                           float* dst);
// LDS-DMA tile configurations: id -> (BM, BN); returns false for an unknown id.
bool conv_dma_tile(int dma_cfg, int* BM, int* BN);
struct ConvArgs;
hipError_t conv_dma_launch_kernel(const ConvLayer& L, const ConvArgs& a, int blocks, hipStream_t stream);

struct ConvPlan {           // launch geometry of one layer at one problem size
    int mtiles, ntiles, splits, chunks_per_split, Mpad, Npad;
    size_t partial_elems;   // 0 when splits == 1
};
ConvPlan conv_plan(const ConvLayer& L, int M, int force_splits = 0);

struct ConvIO {
    const float* in0; const float* in1;
    int B, Hin, Win;
    int act, pool, nchw;
    const float* resid;
    float* out; float* out2; const float* s2; const float* t2;
    float* partial;         // split-K workspace
    size_t partial_cap;     // its capacity in floats (the slice count is clamped to fit)
    int split_n;            // ConvArgs::split_n
};
hipError_t conv_launch(const ConvLayer& L, const ConvIO& io, hipStream_t stream, int force_splits = 0);

// ---- spatial-patch (halo in LDS) kernel for the collapsed UpBlock2d convolution (conv_mfma_patch.hip)
struct PatchLayer {
    int C0 = 0, C1 = 0, Cout = 0;
    float* w = nullptr;     // device, packed [ntile][cchunk][phase][tap][64][32]
    float* w_poly = nullptr;  // device, polyphase minimal-filtering form [ntile 32][cchunk][point 9][32][32] (conv_mfma_patch_poly.hip)
    float* bias = nullptr;  // device, [ntiles*64]
};
size_t patch_packed_elems(int Cin_packed, int Cout);
size_t patch_poly_packed_elems(int Cin_packed, int Cout);
void patch_poly_pack_host(const float* w_oihw_3x3, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst);
// splits > 1: the channel reduction split over that many workgroups per tile, raw sums into partial[splits][B,2H,2W,Cout]
// (partial_cap floats), finished by patch_poly_reduce_kernel; patch_poly_splits picks the split for a launch
hipError_t patch_poly_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                             float* out, hipStream_t stream, int splits = 1, float* partial = nullptr, size_t partial_cap = 0);
int patch_poly_splits(const PatchLayer& L, int B, int H, int W, int max_splits, int cus);
void patch_pack_host(const float* w_oihw_3x3, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst);
hipError_t patch_phase_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                              float* out, hipStream_t stream);

// ---- column-patch kernel for the 7x1 convolution of the final layer (conv_col7.hip): in [B,H,W,C] (C = 32 or 64),
// w = conv_pack_host(..., kh 7, kw 1, BN 32, swizzle) image [C/32][7][32][32], out [B,H,W,32]
hipError_t conv_col7_launch(const float* in, int C, int B, int H, int W, const float* w_swizzled, float* out,
                            hipStream_t stream);
// the same with the horizontal gather, bias and sigmoid in the tile epilogue: out_nchw [B,3,H,W] = sigmoid(conv7x7 + bias)
// (weights: the row-split image above built from a 3-output-channel 7x7 filter, bias [3] on the device)
hipError_t conv_col7_fused_launch(const float* in, int C, int B, int H, int W, const float* w_swizzled, const float* bias,
                                  float* out_nchw, hipStream_t stream);

// streamed-weight variant for the row-split flow head: two concatenated inputs, N = ntile32*32 (3 -> 96), weights packed with
// conv_pack_host(..., kh 7, kw 1, BN 96, swizzle); out [B,H,W,pixel_stride]
hipError_t conv_col7s_launch(const float* in0, int C0, const float* in1, int C1, int B, int H, int W, const float* w_swizzled,
                             int ntile32, float* out, int pixel_stride, hipStream_t stream);

// ---- Winograd paths for the bottleneck convolutions: F(2x2,3x3) (conv_winograd.hip), F(4x4,3x3) (conv_winograd4.hip)
struct WinoLayer {
    int Cin = 0, Cout = 0, BN = 128, ntiles = 0;
    int tile = 2;           // output tile side: 2 -> 16 transform points, BN 128; 4 -> 36 points, BN 64
    float* u = nullptr;     // device, packed [ntiles][(tile+2)^2*Cin/32][BN][32]
    float* bias = nullptr;  // device, [ntiles*BN]
};
size_t wino_packed_elems(int Cout, int Cin, int BN);
void wino_pack_host(const float* w_oihw, int Cout, int Cin, int BN, float* dst);
// V[16][B*H/2*W/2][C] = B^T d B of x (optionally of relu(x*s + t))
hipError_t wino_transform_launch(const float* x, const float* s, const float* t, int B, int H, int W, int C, float* V,
                                 hipStream_t stream);
hipError_t wino_gemm_launch(const WinoLayer& L, const float* V, int B, int H, int W, int act, const float* resid,
                            float* out, hipStream_t stream, int variant = 0);

size_t wino4_packed_elems(int Cout, int Cin, int BN);
void wino4_pack_host(const float* w_oihw, int Cout, int Cin, int BN, float* dst);
// V[36][B*H/4*W/4][C] = B^T d B of x (optionally of relu(x*s + t)); H, W multiples of 4
hipError_t wino4_transform_launch(const float* x, const float* s, const float* t, int B, int H, int W, int C, float* V,
                                  hipStream_t stream);
// groups in {1,2,3,6}: > 1 splits the six rows of transform points over that many workgroups per tile (few tiles) and
// finishes with wino4_output_transform_kernel; zbuf: [24][B*H/4*W/4][Cout] floats; pool (groups > 1 only): 2x2 average
// after the activation, out is [B,H/2,W/2,Cout]
hipError_t wino4_gemm_launch(const WinoLayer& L, const float* V, int B, int H, int W, int act, const float* resid,
                             float* out, hipStream_t stream, int variant = 0, int groups = 1, float* zbuf = nullptr, int pool = 0);

// ---- the generator's first block: 7x7 conv 3 -> Cout (+ folded BatchNorm, ReLU) from the NCHW source (conv_first.hip)
bool first7_supported(int Cout);
void first7_pack_host(const float* w_oihw_7x7_folded, int Cout, float* dst /*[196][Cout]*/);
hipError_t first7_launch(const float* src /*[ns,3,H,W]*/, const float* w_packed, const float* bias, int ns, int H, int W,
                         int Cout, float* out /*[ns,H,W,Cout]*/, hipStream_t stream);

// ---- motion / warp / image kernels (motion.hip) ---------------------------------------------
hipError_t kp_prepare_launch(const float* kd_val, const float* kd_jac, const float* ks_val, const float* ks_jac,
                             int n, int ns, int K, float* kp_rec, int* bad_flag, hipStream_t s);
hipError_t motion_front_launch(const float* kp_rec, const float* src_small /*[ns,h,w,4]*/, int n, int ns, int K,
                               int h, int w, float variance, int Cpad, float* hg_in /*[n,h,w,Cpad]*/,
                               float* sparse_deformed /*[n,K+1,channels,h,w] or null*/, hipStream_t s, int groups = 1,
                               int channels = 3, size_t group_stride = 0);   // groups of three image channels (motion.hip)
hipError_t motion_head_launch(const float* logits /*[n,h,w,32]*/, const float* kp_rec, int n, int K, int h, int w,
                              int has_occ, float* deformation /*[n,h,w,2]*/, float* occlusion /*[n,h,w]*/,
                              float* mask_out /*[n,K+1,h,w] or null*/, float* occ_out /*[n,1,h,w] or null*/,
                              hipStream_t s);
hipError_t motion_head_rowsplit_launch(const float* part /*[n,h,w,PS]: channel dx*NC+co*/, int PS, int NC,
                                       const float* bias /*[NC] dev*/, const float* kp_rec, int n, int K, int h, int w,
                                       int has_occ, float* deformation, float* occlusion, float* mask_out,
                                       float* occ_out, hipStream_t s);
hipError_t warp_features_launch(const float* feat /*[ns,hf,wf,C]*/, const float* deformation /*[n,h,w,2]*/,
                                const float* occlusion /*[n,h,w] or null*/, int n, int ns, int hf, int wf, int C,
                                int h, int w, float* out, float* out2, const float* s2, const float* t2,
                                hipStream_t s);
hipError_t broadcast_features_launch(const float* feat /*[ns,hf,wf,C]*/, int n, int ns, int hf, int wf, int C, float* out,
                                     float* out2, const float* s2, const float* t2, hipStream_t s);
hipError_t warp_image_launch(const float* src /*[ns,3,H,W]*/, const float* deformation /*[n,h,w,2]*/, int n, int ns,
                             int H, int W, int h, int w, float* out /*[n,channels,H,W]*/, hipStream_t s, int src_planes = 3,
                             int channels = 3);
hipError_t source_prepare_launch(const float* src /*[ns,3,H,W]*/, const float* aa_w /*[3,13,13] dev*/, int ns, int H,
                                 int W, int inv_scale, int Cpad, float* src_nhwc /*[ns,H,W,Cpad]*/,
                                 float* src_small /*[ns,h,w,4]*/, hipStream_t s);
hipError_t final_shift_sum_launch(const float* part /*[n,H,W,32]: channel dx*3+co*/, const float* bias /*[3] dev*/,
                                  int n, int H, int W, float* out /*[n,3,H,W]*/, hipStream_t s);
hipError_t antialias_down_launch(const float* src /*[ns,3,H,W]*/, const float* aa_w, int ns, int H, int W,
                                 int inv_scale, int Cpad, float* dst /*[ns,h,w,Cpad]: RGB + zeros*/, hipStream_t s,
                                 int src_planes = 3, int first_channel = 0, int channels = 3, int slot = 0);
hipError_t nchw_to_nhwc_pad_launch(const float* src /*[B,planes,H,W]*/, int B, int C, int H, int W, int Cpad,
                                   float* dst /*[B,H,W,Cpad]*/, hipStream_t s, int planes = 0 /*0: = C; else the first C of them*/);
hipError_t kp_head_launch(const float* logits /*[B,h,w,Cs]*/, int B, int K, int njm, int h, int w, int Cs, int pad,
                          float temperature, float* value /*[B,K,2]*/, float* jacobian /*[B,K,2,2] or null*/,
                          float* heatmap /*[B,K,h-6+2pad,w-6+2pad] or null*/, hipStream_t s,
                          float* ws = nullptr /* kp_head_workspace_floats(B, K) floats: batched calls run the pixel-sliced form */);
constexpr int KP_HEAD_SLICES = 8;
size_t kp_head_workspace_floats(int B, int K);
// ---- training-mode BatchNorm forward (batchnorm.hip; SURVEY.md 8f row N4, first slice)
size_t bn_workspace_floats(int N, int C, int HW);
hipError_t bn_local_sums_launch(const float* x /*[N,C,HW]*/, int N, int C, int HW, float* sums /*[6C+2]*/, float* workspace,
                                hipStream_t s);
hipError_t bn_finalize_launch(const float* sums, int C, float eps, float momentum, int mode, const float* weight,
                              float* running_mean, float* running_var, float* mean, float* scale, float* inv_std /*[C] or null*/,
                              hipStream_t s);
// backward (round 3): sums in the forward's packed layout [6C+2]; coef [3C] = S1/N, S2*inv_std^2/N, weight*inv_std
hipError_t bn_bwd_sums_launch(const float* x, const float* dy, const float* mean, int N, int C, int HW, float* sums, float* workspace,
                              hipStream_t s);
hipError_t bn_bwd_finalize_launch(const float* local, const float* reduced, int C, const float* inv_std, const float* weight, float eps,
                                  int mode, float* dweight, float* dbias, float* coef, hipStream_t s);
hipError_t bn_bwd_apply_launch(const float* x, const float* dy, const float* mean, const float* coef, int N, int C, int HW, float* dx,
                               hipStream_t s);
hipError_t bn_apply_launch(const float* x, const float* mean, const float* scale, const float* bias, int N, int C, int HW,
                           float* y, hipStream_t s);
// NHWC variants for the engine's training-mode forward (batchnorm_nhwc.hip); sums / finalize as above
// The finalize step of one replica folded into bn_combine_kernel (kind 1: bn_finalize_kernel's modes 0 / 1; kind 2:
// bn_bwd_finalize_kernel's with reduced == local): nothing is exchanged in between and the step reads only its own channel's totals.
struct BnFuse {
    int kind = 0, mode = 1;
    float eps = 0.f, momentum = 0.f;
    const float* weight = nullptr;
    float *running_mean = nullptr, *running_var = nullptr, *mean = nullptr, *scale = nullptr, *inv_std_out = nullptr;   // kind 1
    const float* inv_std = nullptr;                                                                                   // kind 2
    float *dweight = nullptr, *dbias = nullptr, *coef = nullptr;
};
hipError_t bn_combine_launch(const double* partial /*[C][P][2]*/, int C, int P, long long count, float* sums /*[6C+2]*/, hipStream_t s,
                             const BnFuse* fuse = nullptr);
size_t bn_nhwc_workspace_floats(long long M, int C);
hipError_t bn_nhwc_sums_launch(const float* x /*[M,C]*/, long long M, int C, float* sums /*[6C+2]*/, float* workspace, hipStream_t s,
                               const BnFuse* fuse = nullptr);
hipError_t bn_nhwc_apply_launch(const float* x /*[B,H,W,C]*/, const float* mean, const float* scale, const float* bias, int B, int H,
                                int W, int C, int relu, int pool, float* y, hipStream_t s);
// backward of the fused tail y = [avgpool2x2](act(BatchNorm(x))) on NHWC: packed sums like bn_bwd_sums_launch, then dx with
// bn_bwd_finalize_launch's coefficients (the ReLU mask is recomputed from x)
hipError_t bn_nhwc_bwd_sums_launch(const float* x, const float* dy, const float* mean, const float* scale, const float* bias, int B, int H,
                                   int W, int C, int relu, int pool, float* sums, float* workspace, hipStream_t s,
                                   const BnFuse* fuse = nullptr);
hipError_t bn_nhwc_bwd_apply_launch(const float* x, const float* dy, const float* mean, const float* scale, const float* bias,
                                    const float* coef, int B, int H, int W, int C, int relu, int pool, float* dx, hipStream_t s);
hipError_t to_u8_launch(const float* pred /*[n,3,H,W]*/, int n, int H, int W, uint8_t* out /*[n,H,W,3]*/,
                        hipStream_t s);

// backward.hip: gradients of the warp (features, flow, occlusion; ACCUMULATING into zeroed buffers) and of a stride-1 "same"
// convolution's weights and bias (NHWC activations, OIHW gradient)
hipError_t warp_features_backward_launch(const float* feat, const float* defo, const float* occ, const float* dout, int n, int ns,
                                         int hf, int wf, int C, float* dfeat, float* ddefo, float* docc, hipStream_t s);
size_t conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw);
hipError_t conv_wgrad_launch(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int kh, int kw, float* dweight,
                             float* dbias, float* workspace, size_t workspace_floats, hipStream_t s, const float* x_transformed = nullptr);
bool conv_wgrad_takes_transformed(int B, int H, int W, int Cin, int Cout, int kh, int kw);

// conv_pack_dev.hip: the packings of conv_pack_host (register-staged kernel) and wino4_pack_host from a DEVICE OIHW tensor;
// transposed = 1 reads the forward filter [Cin][Cout][kh][kw] transposed and flipped (the data gradient's filter)
hipError_t conv_pack_dev_launch(const float* w, int Cout, int Cin, int T, int BN, int transposed, float* dst, hipStream_t s);
hipError_t wino4_pack_dev_launch(const float* w, int Cout, int Cin, int BN, int transposed, float* dst, hipStream_t s);
hipError_t bias_pad_dev_launch(const float* b, int Cout, int n, float* dst, hipStream_t s);
hipError_t col7_pack_dev_launch(const float* w_3c77, int C, float* dst /*[7*C/32][32][32]*/, hipStream_t s);

// conv7_thin.hip: the 7x7 layers with three channels on one side (training path); thin tensors are [B,H,W,4] (4th channel 0)
size_t conv7_thin_workspace_floats(int B, int H, int W, int N);
hipError_t conv7_thin_in_launch(const float* thin, const float* w, const float* bias, int B, int H, int W, int N, int transposed,
                                float* out, float* workspace, hipStream_t s);
hipError_t conv7_thin_wgrad_launch(const float* thin, const float* wide, int B, int H, int W, int N, int thin_is_input, float* dw,
                                   float* workspace, hipStream_t s);

// ---- backward of the dense-motion front end / flow head (motion_backward.hip; forward kernels in motion.hip)
hipError_t antialias_down_backward_launch(const float* dsmall /*[B,h,w,4]*/, const float* aa_w, int B, int H, int W, int inv_scale,
                                          float* dsrc /*[B,3,H,W]*/, hipStream_t s);
hipError_t kp_records_backward_launch(const float* kd_jac, const float* ks_jac, const float* drec, int n, int K, float* dkd_val,
                                      float* dks_val, float* dkd_jac, float* dks_jac, hipStream_t s);
size_t motion_backward_workspace_floats(int n, int K, int h, int w);
hipError_t motion_front_backward_launch(const float* rec, const float* src_small, int n, int K, int h, int w, float variance, int Cpad,
                                        const float* dhg, const float* dsd, float* dsrc_small, float* drec, float* workspace,
                                        hipStream_t s);
hipError_t motion_head_forward_launch(const float* lm, int ld, const float* lo, int ldo, const float* rec, int n, int K, int h, int w,
                                      float* mask, float* deformation, float* occlusion, hipStream_t s);
hipError_t motion_head_backward_launch(const float* mask, const float* occlusion, const float* rec, int n, int K, int h, int w,
                                       const float* dmask, const float* ddef, const float* docc, float* dlm, int ld, float* dlo, int ldo,
                                       float* drec, float* workspace, hipStream_t s);

// One-Euro smoothing of [T,E] sequences along T (keypoints.hip; reference filter1.py:13-47 as driven by demo.py:241-250).
hipError_t one_euro_launch(const float* x, int T, int E, float mincutoff, float beta, float dcutoff, float freq, float scale,
                           float* out, hipStream_t stream, float* state = nullptr, int resume = 0);

}  // namespace eamm
