// Spatial-patch ("halo in LDS") variant of the collapsed UpBlock2d convolution (reference modules/util.py:883-900:
// nearest x2 -> 3x3 -> BN -> ReLU, evaluated as four 2x2 phase filters, see conv_mfma.hip).
//
// The im2col-style kernels re-fetch the activation once per (phase, tap): 16 times per up-block.  With N = 64..128
// output channels that operand stream, not the matrix pipe, bounds them (rocprofv3: ~8 B/clk/CU through the LDS-DMA
// path, profiles/r01_convbench_*).  Here one workgroup owns a 16x16 tile of LOW-RESOLUTION pixels of one image and
// all four output phases of 64 output channels:
//   * per 32-channel chunk the (16+2)x(16+2) input patch is DMA'd into LDS ONCE (41 KiB, swizzled like
//     conv_mfma_dma.hip: slot q of patch pixel i at q ^ ((i>>1)&7)); every (phase, tap) window is a shifted view of
//     it -- the A fragment of lane (row r, col c) for input offset (dy,dx) is patch pixel (r+dy+1)*18 + (c+dx+1);
//   * the weights stream in groups of one phase (4 taps x [64][32] = 32 KiB, host pre-swizzled), double buffered;
//   * one barrier per (chunk, phase): 4 taps x 2 MFMA tiles x 16 = 128 MFMAs per wave between barriers; the DMA
//     pieces of the next interval (4 weight pieces + at most 2 patch pieces per wave) are interleaved into the MFMA
//     stream; 8 waves, each 2 rows x 16 pixels x 64 channels x 4 phases = 128 accumulator VGPRs;
//   * epilogue per phase through LDS: bias (folded BatchNorm), ReLU, 16-byte stores to pixel (2y+py, 2x+px).
// Operand traffic per chunk drops from 16 x 32 KiB (A) to 41 KiB; the kernel is matrix-pipe bound.
#include "conv_common.h"

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int PT = 16;                   // tile side in low-resolution pixels
constexpr int PW = PT + 2;               // patch side (halo 1)
constexpr int PPIX = PW * PW;            // 324 patch pixels
constexpr int PPAD = (PPIX + 7) / 8 * 8; // rounded to whole DMA instructions (8 pixels each)
constexpr int PBN = 64;                  // output channels per workgroup
constexpr int PNT = 2;                   // 32-wide MFMA tiles along N per wave
constexpr int PWAVES = 8;
}  // namespace

struct PatchArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] (hourglass skip concatenation) or null
    int C0, C1;
    unsigned in0_bytes, in1_bytes, w_bytes;
    int B, H, W;           // low-resolution input size; output is [B,2H,2W,Cout]
    int tiles_x, tiles_y, ntiles;
    const float* w;        // packed [ntile][cchunk][phase][tap][64][32], swizzled
    const float* bias;     // [ntiles*64]
    int Cout, act;
    float* out;
};

__global__ __launch_bounds__(PWAVES * 64) void conv_patch_phase_kernel(const PatchArgs p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = PPAD * BK;            // floats
    constexpr int B_STAGE = 4 * PBN * BK;         // one phase: 4 taps
    constexpr int A_INSTR = (PPAD / 8 + PWAVES - 1) / PWAVES;   // patch DMA instructions per wave per chunk (6)
    constexpr int B_INSTR = 4 * PBN / 8 / PWAVES;               // weight DMA instructions per wave per interval (4)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][A_STAGE] [2][B_STAGE]
    float* const As = smem;
    float* const Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % p.ntiles;
    L /= p.ntiles;
    const int tx0 = (L % p.tiles_x) * PT;
    L /= p.tiles_x;
    const int ty0 = (L % p.tiles_y) * PT;
    const int b = L / p.tiles_y;
    const int cchunks = (p.C0 + p.C1) / BK;

    // ---- patch loader: DMA instruction j of this wave covers patch pixels (wave + 8j)*8 .. +8
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int ppix[A_INSTR], pslot[A_INSTR];   // global pixel index (or -1) and channel slot of this lane's 16 bytes
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int i = (wave + PWAVES * j) * 8 + (lane >> 3);   // patch pixel
        const int py = i / PW, px = i - py * PW;
        const int y = ty0 + py - 1, x = tx0 + px - 1;
        const bool ok = i < PPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        ppix[j] = ok ? (b * p.H + y) * p.W + x : -1;
        pslot[j] = ((lane & 7) ^ ((i >> 1) & 7)) << 2;
    }
    auto dma_patch_piece = [&](auto jc, int cc, int st) {
        constexpr int j = decltype(jc)::value;
        if ((wave + PWAVES * j) * 8 < PPAD) {
            const int c0 = cc * BK;
            const bool first = c0 < p.C0;
            const int C = first ? p.C0 : p.C1;
            const int coff = first ? c0 : c0 - p.C0;
            const unsigned off = ppix[j] >= 0 ? (unsigned)(ppix[j] * C + coff + pslot[j]) * 4u : OOB;
            float* dst = As + st * A_STAGE + (wave + PWAVES * j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
    };
    // weights of interval it = cc*4 + phase: 32 KiB contiguous
    auto dma_weight_piece = [&](auto jc, int it, int st) {
        constexpr int j = decltype(jc)::value;
        const unsigned off = (unsigned)(((ntile * cchunks * 4 + it) * 4 * PBN + (wave * B_INSTR + j) * 8) * BK + lane * 4) * 4u;
        float* dst = Bs + st * B_STAGE + (wave * B_INSTR + j) * (8 * BK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)dst, 16, off, 0, 0, 0);
    };

    f32x16 acc[4][PNT];
    static_for<4>([&](auto pc) {
        static_for<PNT>([&](auto jc) {
            static_for<16>([&](auto rc) { acc[decltype(pc)::value][decltype(jc)::value][decltype(rc)::value] = 0.f; });
        });
    });

    // this lane's pixel inside the tile: wave w owns rows 2w, 2w+1
    const int trow = 2 * wave + (l31 >> 4), tcol = l31 & 15;
    const int idx0 = (trow + 1) * PW + tcol + 1;   // patch pixel of input offset (0,0)

    // One interval = one phase of one channel chunk: 4 taps x PNT tiles x 16 MFMAs.
    auto compute = [&](auto phc, int a_st, int b_st, int it_next, int cc_next, bool more_w, bool more_a) {
        constexpr int ph = decltype(phc)::value;
        constexpr int oy = (ph >> 1) ? 0 : -1, ox = (ph & 1) ? 0 : -1;
        const float* a_stage = As + a_st * A_STAGE;
        const float* b_stage = Bs + b_st * B_STAGE;
        // keep the 16 per-step fragment addresses from being hoisted out of the chunk loop (they would pin ~40 VGPRs
        // next to the 128 accumulators): the lane bases are opaque inside an interval and re-derived per step
        int idx0_l = idx0, l31_l = l31;
        asm volatile("" : "+v"(idx0_l), "+v"(l31_l));
        f32x4 a[2], bb[2][PNT];
        auto fetch = [&](auto stc, int buf) {   // step = tap*4 + s
            constexpr int step = decltype(stc)::value;
            constexpr int tap = step >> 2, s = step & 3;
            constexpr int shift = ((tap >> 1) + oy) * PW + (tap & 1) + ox;
            const int idx = idx0_l + shift;
            a[buf] = *reinterpret_cast<const f32x4*>(a_stage + idx * BK + ((((2 * s + half)) ^ ((idx >> 1) & 7)) << 2));
            const float* bt = b_stage + tap * (PBN * BK) + l31_l * BK + (((2 * s + half) ^ ((l31_l >> 1) & 7)) << 2);
#pragma unroll
            for (int j = 0; j < PNT; ++j) bb[buf][j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * BK);
        };
        fetch(std::integral_constant<int, 0>{}, 0);
        static_for<16>([&](auto stc) {
            constexpr int step = decltype(stc)::value;
            if constexpr (step + 1 < 16) fetch(std::integral_constant<int, step + 1>{}, (step + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4 * PNT>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / PNT, j = q % PNT;
                acc[ph][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][t], bb[step & 1][j][t], acc[ph][j], 0, 0, 0);
                constexpr int g = step * 4 * PNT + q;   // MFMA index within the interval (128 total)
                // DMA pieces for the next interval: 4 weight pieces, then this phase's share of the next patch
                if constexpr (g % 4 == 3 && g / 4 < B_INSTR) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more_w) dma_weight_piece(std::integral_constant<int, g / 4>{}, it_next, b_st ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (g % 4 == 3 && g / 4 >= B_INSTR && g / 4 < B_INSTR + 2) {
                    constexpr int jp = ph + 4 * (g / 4 - B_INSTR);   // patch pieces ph, ph+4 of this wave
                    if constexpr (jp < A_INSTR) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more_a) dma_patch_piece(std::integral_constant<int, jp>{}, cc_next, a_st ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
        });
    };

    // ---- main loop over (channel chunk, phase)
    static_for<A_INSTR>([&](auto jc) { dma_patch_piece(jc, 0, 0); });
    static_for<B_INSTR>([&](auto jc) { dma_weight_piece(jc, 0, 0); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int b_st = 0;
    for (int cc = 0; cc < cchunks; ++cc) {
        const int a_st = cc & 1;
        const bool more_a = cc + 1 < cchunks;
        static_for<4>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            const int it = cc * 4 + ph;
            const bool more_w = it + 1 < cchunks * 4;
            compute(phc, a_st, b_st, it + 1, cc + 1, more_w, more_a);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            b_st ^= 1;
        });
    }

    // ---- epilogue: per phase, stage the 256 x 64 tile through LDS and store 16-byte pieces
    constexpr int LDO = PBN + 4, C4 = PBN / 4, NTHR = PWAVES * 64, PER = PT * PT * C4 / NTHR;
    const int OH = 2 * p.H, OW = 2 * p.W;
    static_for<4>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        __syncthreads();
        static_for<PNT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int col = j * 32 + l31;
            const float bias = p.bias[ntile * PBN + col];
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // tile pixel: (row >> 4, row & 15)
                smem[row * LDO + col] = acc[ph][j][r] + bias;
            });
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR;
            const int row = idx / C4, c4 = idx - row * C4;
            const int y = ty0 + (row >> 4), x = tx0 + (row & 15);
            const int n = ntile * PBN + c4 * 4;
            if (y < p.H && x < p.W && n < p.Cout) {
                float4 v = *reinterpret_cast<const float4*>(smem + row * LDO + c4 * 4);
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
                v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                const size_t o = ((size_t)(b * OH + 2 * y + (ph >> 1)) * OW + 2 * x + (ph & 1)) * p.Cout + n;
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t patch_packed_elems(int Cin_packed, int Cout) {
    return (size_t)((Cout + PBN - 1) / PBN) * (Cin_packed / CONV_BK) * 16 * PBN * CONV_BK;
}

// w: 3x3 weights [Cout][Cin][3][3] (BatchNorm folded) -> four 2x2 phase filters (pre-summed in double, as
// conv_pack_host does) laid out [ntile][cchunk][phase][tap][64][32] with the LDS swizzle applied.
void patch_pack_host(const float* w, int Cout, int Cin, const int* cin_map, int cin_packed, float* dst) {
    const int BK = CONV_BK, cch = cin_packed / BK;
    const size_t total = patch_packed_elems(cin_packed, Cout);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < Cout; ++o) {
        const int nt = o / PBN, nl = o % PBN;
        for (int cp = 0; cp < cin_packed; ++cp) {
            const int c = cin_map ? cin_map[cp] : cp;
            if (c < 0 || c >= Cin) continue;
            const float* s = w + ((size_t)o * Cin + c) * 9;
            const int cc = cp / BK, kl = cp % BK;
            const int kk = ((((kl >> 2) ^ ((nl >> 1) & 7)) << 2) | (kl & 3));
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
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
                        const size_t tile = (((size_t)nt * cch + cc) * 4 + ph) * 4 + (ty * 2 + tx);
                        dst[(tile * PBN + nl) * BK + kk] = (float)acc;
                    }
            }
        }
    }
}

hipError_t patch_phase_launch(const PatchLayer& L, const float* in0, const float* in1, int B, int H, int W, int act,
                              float* out, hipStream_t stream) {
    if ((L.C0 % CONV_BK) || (L.C1 % CONV_BK) || (L.Cout & 3)) return hipErrorInvalidValue;
    PatchArgs a{};
    a.in0 = in0;
    a.in1 = L.C1 ? in1 : nullptr;
    a.C0 = L.C0;
    a.C1 = L.C1;
    const size_t px = (size_t)B * H * W;
    const size_t b0 = px * L.C0 * 4, b1 = px * L.C1 * 4, bw = patch_packed_elems(L.C0 + L.C1, L.Cout) * 4;
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.in0_bytes = (unsigned)b0;
    a.in1_bytes = (unsigned)b1;
    a.w_bytes = (unsigned)bw;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = (W + PT - 1) / PT;
    a.tiles_y = (H + PT - 1) / PT;
    a.ntiles = (L.Cout + PBN - 1) / PBN;
    a.w = L.w;
    a.bias = L.bias;
    a.Cout = L.Cout;
    a.act = act;
    a.out = out;
    constexpr size_t lds_loop = sizeof(float) * 2 * (PPAD * CONV_BK + 4 * PBN * CONV_BK);
    constexpr size_t lds_epi = sizeof(float) * (PT * PT) * (PBN + 4);
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static lds_once_mask configured{0};
    if (hipError_t e = ensure_dynamic_lds(conv_patch_phase_kernel, lds, &configured); e != hipSuccess) return e;
    const int blocks = a.tiles_x * a.tiles_y * B * a.ntiles;
    note_mfma_flops(2.0 * a.tiles_x * a.tiles_y * B * (PT * PT) * 16.0 * (a.ntiles * PBN) * (L.C0 + L.C1));
    hipLaunchKernelGGL(conv_patch_phase_kernel, dim3(blocks), dim3(PWAVES * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace eamm
