// KHxKW convolution on NHWC fp32 activations as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces every F.conv2d + batch_norm + relu (+ avg_pool2d / nearest-upsample / cat / residual add)
// group of the reference's DownBlock2d / UpBlock2d / ResBlock2d / SameBlock2d (modules/util.py:858-938)
// with ONE kernel: eval-mode BatchNorm is folded into the weights at load time, the activation,
// 2x2 average pool, residual add and the next block's pre-activation are epilogues, the channel
// concatenation of the hourglass decoder is address arithmetic in the operand loader, and
// UpBlock2d's "nearest x2 then 3x3" is evaluated in its exact collapsed form: output pixel
// (2y+py, 2x+px) only ever sees a 2x2 neighbourhood of the low-resolution input, with the 3x3 taps
// that land on the same input pixel pre-summed -- four 2x2 "phase" convolutions, 2.25x fewer MACs
// (PHASE mode; nothing is up-sampled or materialised).
//
// GEMM view:  M = B*H*W pixels, N = Cout, K = taps * Cin, exact fp32 via v_mfma_f32_32x32x2_f32
// (bitwise an fmaf chain; gfx950 has no TF32/xf32).
//   * block tile BM x BN = (WM*MT*32) x (WN*NT*32), 4 waves, each wave MT x NT MFMA tiles;
//   * K is walked in chunks of 32 input channels of one filter tap; channel-chunk outer, tap inner,
//     so the shifted re-reads of an activation line hit L2 back to back;
//   * both operands are staged HBM/L2 -> VGPR -> LDS with raw buffer loads (out-of-range offset =>
//     hardware zero fill = the zero padding and the M tail, no branches), double buffered, one
//     barrier per chunk, laid out [row][32+4] so that each lane fetches FOUR consecutive k with one
//     conflict-free ds_read_b128: lanes 0-31 take k = 8s+{0..3}, lanes 32-63 k = 8s+{4..7}; A and B use
//     the same permutation of K, which the sum does not care about;
//   * M is enumerated in 2x2-quad order (m = 4*quad + 2*jy + jx): the four accumulator registers
//     4g..4g+3 of a lane are then exactly one pooling window, so AvgPool2d(2) is in-register;
//   * small-M layers (the deep hourglass levels, small batches) use split-K over gridDim with an fp32
//     slab reduction kernel that applies the same epilogue;
//   * blockIdx is remapped so that consecutive logical tiles (same M tile, all N tiles) share an XCD
//     and therefore an L2.
#include "conv_common.h"
#include <map>
#include <mutex>
#include <string>
#include <cstdlib>
#include <algorithm>

#include <algorithm>
#include <cstring>
#include <vector>

namespace eamm {

template <int KH, int KW, int MT, int NT, int WM, int WN, bool PHASE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int BK = CONV_BK, LDK = CONV_LDK;
    constexpr int T = KH * KW;
    constexpr int A_PER = BM / 32, B_PER = BN / 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(!PHASE || (KH == 2 && KW == 2), "phase mode is the collapsed nearest-x2 + 3x3");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;  // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    L /= p.ntiles;
    const int mtile = L % p.mtiles;
    L /= p.mtiles;
    int phase = 0;
    if (PHASE) {
        phase = L & 3;
        L >>= 2;
    }
    const int split = L;
    const int mbase = mtile * BM;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    // tap (ty,tx) reads input pixel (y + ty + oy, x + tx + ox)
    const int oy = PHASE ? ((phase >> 1) ? 0 : -1) : -(KH / 2);
    const int ox = PHASE ? ((phase & 1) ? 0 : -1) : -(KW / 2);

    // ---- operand loader state: this thread stages rows arow+32j, 16 bytes at column acol
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    int ry[A_PER], rx[A_PER], rb[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int m = mbase + arow + 32 * j;
        if (m < p.M) {
            int b;
            pix_decode(p, m, b, ry[j], rx[j]);
            rb[j] = b * p.H * p.W;
        } else {
            ry[j] = -(1 << 20);  // never inside the image
            rx[j] = 0;
            rb[j] = 0;
        }
    }

    // Buffer descriptors (wave-uniform, built from kernel arguments): an out-of-range offset makes the
    // hardware return zeros, so zero padding and the M tail need no branch -- the loads stay in flight
    // under the MFMAs and are first waited for at the LDS store.
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int wtile = (phase * p.ntiles + ntile) * p.nchunks;

    u32x4 av[A_PER], bv[B_PER];
    auto load_chunk = [&](int ci) {
        const int cc = ci / T, tap = ci - cc * T;
        const int dy = tap / KW + oy, dx = tap % KW + ox;
        const int c0 = cc * BK;
        const bool first = c0 < p.C0;
        const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
        const int C = first ? p.C0 : p.C1;
        const int coff = (first ? c0 : c0 - p.C0) + acol;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int yy = ry[j] + dy, xx = rx[j] + dx;
            const bool ok = ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
            const unsigned off = ok ? (unsigned)((rb[j] + yy * p.W + xx) * C + coff) * 4u : OOB;
            av[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
        const unsigned woff = (unsigned)((wtile + ci) * (BN * BK) + tid * 4) * 4u;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) bv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + j * 4096u, 0, 0);
    };
    auto store_chunk = [&](int st) {
        float* a = As + st * BM * LDK + arow * LDK + acol;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) *reinterpret_cast<u32x4*>(a + j * 32 * LDK) = av[j];
        float* b = Bs + st * BN * LDK + arow * LDK + acol;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) *reinterpret_cast<u32x4*>(b + j * 32 * LDK) = bv[j];
    };

    f32x16 acc[MT][NT];
    static_for<MT>([&](auto ic) {
        static_for<NT>([&](auto jc) {
            static_for<16>([&](auto rc) { acc[decltype(ic)::value][decltype(jc)::value][decltype(rc)::value] = 0.f; });
        });
    });

    auto compute = [&](int st) {
        const float* a_base = As + st * BM * LDK + (wm * MT * 32 + l31) * LDK + half * 4;
        const float* b_base = Bs + st * BN * LDK + (wn * NT * 32 + l31) * LDK + half * 4;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDK + s * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDK + s * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: LDS double buffer, one barrier per K chunk
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        int st = 0;
        for (int ci = c_begin; ci < c_end; ++ci) {
            const bool more = ci + 1 < c_end;
            if (more) load_chunk(ci + 1);  // buffer loads in flight under the MFMAs
            compute(st);
            if (more) store_chunk(st ^ 1);
            __syncthreads();
            st ^= 1;
        }
    }

    // ---- epilogue
    conv_epilogue<MT, NT, BN>(p, acc, mbase, ntile, wm, wn, l31, half, phase, split);
}

// Split-K slab reduction + the same epilogue as the fused path.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs p, int splits) {
    const int rows = p.pool ? (p.M >> 2) : p.M;
    const size_t per_phase = (size_t)rows * p.Cout;
    const size_t total = per_phase * p.nphase;
    const float* __restrict__ partial = p.partial;
    const size_t slab = (size_t)p.Mpad * p.Npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int phase = (int)(idx / per_phase);
        const size_t rem = idx - (size_t)phase * per_phase;
        const int n = (int)(rem % p.Cout);
        const int r = (int)(rem / p.Cout);
        const float bias = p.bias[n];
        if (p.pool) {
            // the four pixels of the window as four independent sums (sixteen loads in flight per thread); each keeps its order
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int sp = 0; sp < splits; ++sp) {
                const float* q = partial + (size_t)(sp * p.nphase + phase) * slab + (size_t)(4 * r) * p.Npad + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] += q[(size_t)e * p.Npad];
            }
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) v += apply_act(s4[e] + bias, p.act);
            p.out[(size_t)r * p.Cout + n] = 0.25f * v;
        } else {
            float s = 0.f;
#pragma unroll 16   // independent loads in flight; the additions keep their order
            for (int sp = 0; sp < splits; ++sp)
                s += partial[(size_t)(sp * p.nphase + phase) * slab + (size_t)r * p.Npad + n];
            int b, y, x;
            pix_decode(p, r, b, y, x);
            float s2 = 0.f, t2 = 0.f;
            if (p.out2 != nullptr && p.split_n == 0) {
                s2 = p.s2[n];
                t2 = p.t2[n];
            }
            epilogue_store(p, phase, b, y, x, n, s + bias, s2, t2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int conv_tile_n(int Cout) {
    if (Cout > 64) return 128;
    if (Cout > 32) return 64;
    return 32;
}

size_t conv_packed_elems(int taps, int cin_packed, int Cout, int BN, int nphase) {
    const int ntiles = (Cout + BN - 1) / BN;
    return (size_t)nphase * ntiles * BN * taps * cin_packed;
}

// w: [Cout][Cin][taps] (taps row-major kh*KW+kw) -> dst [ntiles][nchunks][BN][BK]
static void pack_one(const float* w, int Cout, int Cin, int T, const int* cin_map, int cin_packed, int BN,
                     bool swizzle, float* dst) {
    const int BK = CONV_BK;
    const int ntiles = (Cout + BN - 1) / BN;
    const int nchunks = T * (cin_packed / BK);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ci = 0; ci < nchunks; ++ci) {
            const int cc = ci / T, tap = ci % T;
            float* tile = dst + ((size_t)nt * nchunks + ci) * BN * BK;
            for (int nl = 0; nl < BN; ++nl) {
                const int o = nt * BN + nl;
                if (o >= Cout) continue;
                for (int kl = 0; kl < BK; ++kl) {
                    const int c = cin_map ? cin_map[cc * BK + kl] : cc * BK + kl;
                    if (c < 0 || c >= Cin) continue;
                    // LDS-DMA kernels copy the tile linearly into an XOR-swizzled LDS image: 16-byte slot q of
                    // row r lives at slot q ^ ((r >> 1) & 7) (conv_mfma_dma.hip)
                    const int kk = swizzle ? ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3)) : kl;
                    tile[nl * BK + kk] = w[((size_t)o * Cin + c) * T + tap];
                }
            }
        }
}

bool conv_dma_tile(int dma_cfg, int* BM, int* BN) {
    switch (dma_cfg) {
        case 1: *BM = 256; *BN = 256; return true;   // 8 waves, 64x128 per wave
        case 2: *BM = 256; *BN = 128; return true;   // 8 waves, 64x64 per wave
        case 3: *BM = 512; *BN = 64; return true;    // 8 waves, 64x64 per wave
        case 4: *BM = 32; *BN = 128; return true;    // 4 waves, 32x32 per wave: skinny GEMMs (M <= 32: weights streamed once,
        case 5: *BM = 64; *BN = 128; return true;    //          64x32 per wave: M <= 64   no MFMA passes over padding rows)
        default: return false;
    }
}

void conv_pack_host(const float* w, int Cout, int Cin, int kh, int kw, const int* cin_map, int cin_packed, int BN,
                    bool phase, bool swizzle, float* dst) {
    if (!phase) {
        std::memset(dst, 0, sizeof(float) * conv_packed_elems(kh * kw, cin_packed, Cout, BN, 1));
        pack_one(w, Cout, Cin, kh * kw, cin_map, cin_packed, BN, swizzle, dst);
        return;
    }
    // nearest-x2 followed by 3x3 (pad 1): output row 2y+py reads up-sampled rows 2y+py-1..2y+py+1, i.e.
    // input rows {y-1: ky=0 | y: ky=1,2} for py=0 and {y: ky=0,1 | y+1: ky=2} for py=1 (same along x).
    // Taps that hit the same input pixel are summed (in double) into a 2x2 filter per phase.
    const size_t per_phase = conv_packed_elems(4, cin_packed, Cout, BN, 1);
    std::memset(dst, 0, sizeof(float) * per_phase * 4);
    std::vector<float> w4((size_t)Cout * Cin * 4);
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc) {
            const float* s = w + oc * 9;
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    double acc = 0.0;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int my = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
                        if (my != ty) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int mx = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                            if (mx == tx) acc += (double)s[ky * 3 + kx];
                        }
                    }
                    w4[oc * 4 + ty * 2 + tx] = (float)acc;
                }
        }
        pack_one(w4.data(), Cout, Cin, 4, cin_map, cin_packed, BN, swizzle, dst + per_phase * ph);
    }
}

// Four explicit 2x2 phase filters (not derived from a 3x3 kernel): wph [4][Cout][Cin][4], phase = 2*py + px, tap =
// 2*ty + tx reading input pixel (y - 1 + py + ty, x - 1 + px + tx) for output pixel (2y + py, 2x + px).  This is the
// footprint of ConvTranspose2d(kernel 4, stride 2, padding 1) as well as of the collapsed UpBlock2d.
void conv_pack_phases_host(const float* wph, int Cout, int Cin, const int* cin_map, int cin_packed, int BN, bool swizzle,
                           float* dst) {
    const size_t per_phase = conv_packed_elems(4, cin_packed, Cout, BN, 1);
    std::memset(dst, 0, sizeof(float) * per_phase * 4);
    for (int ph = 0; ph < 4; ++ph)
        pack_one(wph + (size_t)ph * Cout * Cin * 4, Cout, Cin, 4, cin_map, cin_packed, BN, swizzle, dst + per_phase * ph);
}

ConvPlan conv_plan(const ConvLayer& L, int M, int force_splits) {
    ConvPlan pl;
    const int BM = L.BM;
    const int nphase = L.phase ? 4 : 1;
    pl.mtiles = (M + BM - 1) / BM;
    pl.ntiles = L.ntiles;
    const int blocks = pl.mtiles * pl.ntiles * nphase;
    int splits = 1;
    if (force_splits > 0) {
        splits = force_splits;
    } else if (blocks < 192) {
        // deep hourglass levels / small batches: too few output tiles for 256 CUs -> slice K instead
        splits = (256 + blocks - 1) / blocks;
        splits = std::min(splits, std::max(1, L.nchunks / (BM <= 64 ? 4 : 8)));   // skinny tiles: short K slices are fine
    }
    splits = std::max(1, std::min(splits, L.nchunks));
    pl.chunks_per_split = (L.nchunks + splits - 1) / splits;
    pl.splits = (L.nchunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
    pl.Mpad = pl.mtiles * BM;
    pl.Npad = pl.ntiles * L.BN;
    pl.partial_elems = pl.splits > 1 ? (size_t)pl.splits * nphase * pl.Mpad * pl.Npad : 0;
    return pl;
}

template <int KH, int KW, int MT, int NT, int WM, int WN, bool PHASE>
static hipError_t launch_cfg(const ConvArgs& a, int blocks, hipStream_t stream) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t lds = sizeof(float) * 2 * (BM + BN) * CONV_LDK;
    auto kern = conv_mfma_kernel<KH, KW, MT, NT, WM, WN, PHASE>;
    static lds_once_mask configured{0};  // per-device bit mask
    if (hipError_t e = ensure_dynamic_lds(kern, lds, &configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int KH, int KW, bool PHASE>
static hipError_t launch_tile(int BN, const ConvArgs& a, int blocks, hipStream_t stream) {
    if (BN == 128) return launch_cfg<KH, KW, 2, 2, 2, 2, PHASE>(a, blocks, stream);
    if (BN == 64) return launch_cfg<KH, KW, 2, 1, 2, 2, PHASE>(a, blocks, stream);
    if (BN == 32) return launch_cfg<KH, KW, 1, 1, 4, 1, PHASE>(a, blocks, stream);
    return hipErrorInvalidValue;
}

namespace {
thread_local double g_mfma_flops = 0.0;
std::mutex g_mfma_total_mu;
double g_mfma_total = 0.0;        // process-wide, monotonic (autograd runs backward launches on its own thread)
}
void note_mfma_flops(double flops) {
    g_mfma_flops += flops;
    std::lock_guard<std::mutex> lock(g_mfma_total_mu);
    g_mfma_total += flops;
}
double total_mfma_flops() {
    std::lock_guard<std::mutex> lock(g_mfma_total_mu);
    return g_mfma_total;
}
double take_mfma_flops() {
    const double v = g_mfma_flops;
    g_mfma_flops = 0.0;
    return v;
}

namespace {
std::mutex g_knob_mu;
std::map<std::string, std::pair<long long, int>> g_knobs;   // value in effect; 0 default, 1 set, 2 set but ignored
}
// The DOCUMENTED knobs (include/eamm_hip.h): selectors of a computation form or of the launch plan; every setting computes the
// same frames up to rounding.  Everything else the sources read through knob_int() is a TUNING aid (tile thresholds, pipeline
// variants, split sizes): honoured only when EAMM_TUNING=1 is set as well, so that a stray variable in a deployment's environment
// cannot change the plan silently; such a variable is recorded with "set": 2 (present, ignored).
static bool knob_documented(const char* name) {
    static const char* const names[] = {"EAMM_PASS_CHAINS", "EAMM_BNECK_CHAINS", "EAMM_WINO_TILE", "EAMM_WINO_MIN_M", "EAMM_WINO4_MIN_M",
                                        "EAMM_ENC_WINO", "EAMM_FINAL_FUSED", "EAMM_FINAL_MFMA4", "EAMM_COL7", "EAMM_FIRST7",
                                        "EAMM_PRIVATE_STREAMS", "EAMM_WARP_JOINT", "EAMM_BNECK_STAGGER", "EAMM_HEAD_ROWSPLIT",
                                        "EAMM_PATCH_POLY", "EAMM_KPA_THIN", "EAMM_WGRAD_WINO4", "EAMM_WGRAD_ROW", "EAMM_CONV_DEV_WINO4",
                                        "EAMM_WINO4_EPI_V", "EAMM_COL7_DBG", "EAMM_WINO4_VARIANT"};   // (the last three: refused / experiments build)
    for (const char* n : names)
        if (!strcmp(n, name)) return true;
    return false;
}
long long knob_int(const char* name, long long dflt) {
    const char* v = getenv(name);
    static const bool tuning = [] { const char* t = getenv("EAMM_TUNING"); return t && atoi(t) != 0; }();
    const bool honoured = v != nullptr && (tuning || knob_documented(name));
    const long long val = honoured ? atoll(v) : dflt;
    std::lock_guard<std::mutex> lock(g_knob_mu);
    auto it = g_knobs.find(name);
    const int state = v == nullptr ? 0 : (honoured ? 1 : 2);
    if (it == g_knobs.end()) g_knobs.emplace(name, std::make_pair(val, state));
    else if (state) it->second = std::make_pair(val, state);   // (a per-handle default may differ between handles: the set value wins)
    return val;
}
int knobs_json(char* buf, int cap) {
    std::string out = "{";
    {
        std::lock_guard<std::mutex> lock(g_knob_mu);
        bool first = true;
        for (const auto& kv : g_knobs) {
            if (!first) out += ", ";
            first = false;
            out += "\"" + kv.first + "\": {\"value\": " + std::to_string(kv.second.first) + ", \"set\": " + std::to_string(kv.second.second) + "}";
        }
    }
    out += "}";
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(out.size(), (size_t)cap - 1);
        std::memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size();
}

hipError_t conv_launch(const ConvLayer& L, const ConvIO& io, hipStream_t stream, int force_splits) {
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in0 = io.in0;
    a.in1 = io.in1;
    a.C0 = L.C0;
    a.C1 = L.C1;
    a.H = io.Hin;
    a.W = io.Win;
    a.nphase = L.phase ? 4 : 1;
    a.M = io.B * a.H * a.W;
    a.w = L.w;
    {   // buffer-descriptor ranges (32-bit): every tensor the loader touches must stay below 4 GiB
        const size_t px = (size_t)io.B * io.Hin * io.Win;
        const size_t b0 = px * L.C0 * sizeof(float), b1 = px * L.C1 * sizeof(float);
        const size_t bw = (size_t)a.nphase * L.ntiles * L.nchunks * L.BN * CONV_BK * sizeof(float);
        if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
        a.in0_bytes = (unsigned)b0;
        a.in1_bytes = (unsigned)b1;
        a.w_bytes = (unsigned)bw;
    }
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.nchunks = L.nchunks;
    ConvPlan pl = conv_plan(L, a.M, force_splits);
    while (pl.splits > 1 && pl.partial_elems > io.partial_cap) pl = conv_plan(L, a.M, pl.splits - 1);
    a.mtiles = pl.mtiles;
    a.ntiles = pl.ntiles;
    a.chunks_per_split = pl.chunks_per_split;
    a.Mpad = pl.Mpad;
    a.Npad = pl.Npad;
    a.partial = pl.splits > 1 ? io.partial : nullptr;
    a.act = io.act;
    a.pool = io.pool;
    a.nchw = io.nchw;
    a.resid = io.resid;
    a.out = io.out;
    a.out2 = io.out2;
    a.s2 = io.s2;
    a.t2 = io.t2;
    a.split_n = io.split_n;
    if (io.split_n > 0 && (io.out2 == nullptr || io.resid != nullptr || io.pool || io.nchw || L.Cout > io.split_n + 4 || L.Cout <= io.split_n))
        return hipErrorInvalidValue;
    a.linear = ((a.H | a.W) & 1) ? 1 : 0;                   // odd side: raster order (2x2 quads need even sides)
    if (a.linear && io.pool) return hipErrorInvalidValue;   // AvgPool2d(2) windows are the quads
    if (pl.splits > 1 && io.partial == nullptr) return hipErrorInvalidValue;
    if (L.phase && io.pool) return hipErrorInvalidValue;
    const int blocks = pl.mtiles * pl.ntiles * a.nphase * pl.splits;
    note_mfma_flops(2.0 * pl.mtiles * L.BM * (double)pl.ntiles * L.BN * (double)L.nchunks * CONV_BK * a.nphase);
    hipError_t e = hipErrorInvalidValue;
    if (L.dma_cfg > 0) e = conv_dma_launch_kernel(L, a, blocks, stream);
    else if (L.phase) e = launch_tile<2, 2, true>(L.BN, a, blocks, stream);
    else if (L.kh == 3 && L.kw == 3) e = launch_tile<3, 3, false>(L.BN, a, blocks, stream);
    else if (L.kh == 7 && L.kw == 7) e = launch_tile<7, 7, false>(L.BN, a, blocks, stream);
    else if (L.kh == 7 && L.kw == 1) e = launch_tile<7, 1, false>(L.BN, a, blocks, stream);
    else if (L.kh == 1 && L.kw == 1) e = launch_tile<1, 1, false>(L.BN, a, blocks, stream);
    if (e != hipSuccess) return e;
    if (pl.splits > 1) {
        const int rows = io.pool ? (a.M >> 2) : a.M;
        const size_t total = (size_t)rows * L.Cout * a.nphase;
        const int rb = (int)std::min<size_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, stream, a, pl.splits);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace eamm
