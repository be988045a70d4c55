// Big-tile variant of the fp32-MFMA implicit-GEMM convolution (see conv_mfma.hip for the GEMM view):
// 8 waves, block tile (WM*MT*32) x (WN*NT*32) up to 256 x 256, operands streamed HBM/L2 -> LDS by the
// DMA path (buffer_load ... lds), no VGPR staging and no ds_write.
//
// Why: with the 128x128 / 4-wave kernel every wave issues 64 MFMAs (4096 cycles) per K chunk and then
// pays the per-chunk hand-off (LDS store, wait, barrier, first fragment reads).  rocprofv3 on the
// 256->256 bottleneck convolution showed the matrix pipe busy 76 % of cycles with zero LDS bank
// conflicts and HBM traffic ~= algorithmic: the loss is that hand-off.  Here a wave owns 64x128
// outputs = 128 MFMAs (8192 cycles) per chunk, two waves share a SIMD inside ONE workgroup, and the
// hand-off shrinks to "s_waitcnt vmcnt(0); s_barrier".
//
// LDS image: [rows][32 floats] unpadded (a DMA instruction writes 64 lanes x 16 B = 8 rows
// contiguously), 16-byte slot q of row r stored at slot q ^ ((r >> 1) & 7).  A ds_read_b128 lane
// group covers 16 distinct rows at one logical slot; rows alternate between the two 128-B halves of
// the 256-B bank row and (r>>1)&7 spreads them over the 8 slots of a half: conflict-free.  The
// permutation is applied on the SOURCE side: activation lanes fetch channel slot q = p ^ swz(row) of
// their pixel (same 128-B line, so coalescing is unchanged), weight tiles are pre-swizzled on the host.
// Zero padding and the M tail are buffer-descriptor range misses (the DMA then writes zeros).
#include "conv_common.h"

#include <type_traits>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int KH, int KW, int MT, int NT, int WM, int WN, bool PHASE>
__global__ __launch_bounds__(WM* WN * 64) void conv_mfma_dma_kernel(const ConvArgs p) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, BK = CONV_BK;
    constexpr int T = KH * KW;
    constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK;       // floats per stage
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // DMA instructions per wave per chunk
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    static_assert(!PHASE || (KH == 2 && KW == 2), "phase mode is the collapsed nearest-x2 + 3x3");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][A_STAGE] [2][B_STAGE]
    float* const As = smem;
    float* const Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int oy = PHASE ? ((phase >> 1) ? 0 : -1) : -(KH / 2);
    const int ox = PHASE ? ((phase & 1) ? 0 : -1) : -(KW / 2);

    // ---- A loader: DMA instruction j of this wave fills rows (wave*A_INSTR + j)*8 .. +8; lane -> row
    // rbase + lane/8, physical slot lane%8, i.e. channel slot (lane%8) ^ swz(row) of that pixel.
    // Everything a lane contributes to a piece's address is computed ONCE: the byte offset of its (un-shifted)
    // pixel in either input and a bit mask of the filter taps that stay inside the image.  Per piece the MFMA
    // stream then carries one v_add (the chunk's wave-uniform tap / channel displacement), one bit test and
    // one select (zero padding and the M tail = an offset the buffer descriptor rejects) -- the first version
    // redid the 2-D bounds test and the pixel address arithmetic per piece.
    // (7x7 has 49 taps: a 64-bit mask there, round 6 -- the key-point heads' 7x7 convolution on this kernel)
    typedef typename std::conditional<(T > 32), unsigned long long, unsigned>::type tapmask_t;
    unsigned abase0[A_INSTR], abase1[A_INSTR];
    tapmask_t tapmask[A_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
        const int rq = (((lane & 7) ^ ((row >> 1) & 7)) << 2);
        const int m = mbase + row;
        abase0[j] = abase1[j] = 0;
        tapmask[j] = 0;
        if (m < p.M) {
            int b, y, x;
            pix_decode(p, m, b, y, x);
            const int pix = (b * p.H + y) * p.W + x;
            abase0[j] = (unsigned)(pix * p.C0 + rq) * 4u;
            abase1[j] = (unsigned)(pix * p.C1 + rq) * 4u;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int yy = y + t / KW + oy, xx = x + t % KW + ox;
                if (((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W)) tapmask[j] |= (tapmask_t)1 << t;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int wtile = (phase * p.ntiles + ntile) * p.nchunks;
    const unsigned w_lane = (unsigned)lane * 16u;

    // One chunk = NPIECE DMA instructions per wave (A_INSTR activation pieces, then B_INSTR weight pieces).
    constexpr int NPIECE = A_INSTR + B_INSTR;
    // wave-uniform description of the chunk being fetched (set by chunk_src, read by dma_piece)
    int n_first = 1, n_st = 0;
    unsigned n_delta = 0, n_woff = 0;
    tapmask_t n_tapbit = 0;
    auto chunk_src = [&](int ci, int st) {
        const int cc = ci / T, tap = ci - cc * T;
        const int dy = tap / KW + oy, dx = tap % KW + ox;
        const int c0 = cc * BK;
        n_first = c0 < p.C0;
        const int C = n_first ? p.C0 : p.C1;
        const int coff = n_first ? c0 : c0 - p.C0;
        n_delta = (unsigned)(((dy * p.W + dx) * C + coff) * 4);   // two's complement: the lane's add wraps to the right offset
        n_tapbit = (tapmask_t)1 << tap;
        n_st = st;
        n_woff = (unsigned)((wtile + ci) * (BN * BK) + wave * (B_INSTR * 8 * BK)) * 4u;
    };
    auto dma_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < A_INSTR) {
            const unsigned base = n_first ? abase0[k] : abase1[k];
            const unsigned off = (tapmask[k] & n_tapbit) ? base + n_delta : OOB;
            float* dst = As + n_st * A_STAGE + (wave * A_INSTR + k) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(n_first ? rs0 : rs1, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        } else {
            constexpr int j = k - A_INSTR;
            float* dst = Bs + n_st * B_STAGE + (wave * B_INSTR + j) * (8 * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)dst, 16, w_lane, n_woff + j * (8 * BK * 4), 0, 0);
        }
    };

    f32x16 acc[MT][NT];
    static_for<MT>([&](auto ic) {
        static_for<NT>([&](auto jc) {
            static_for<16>([&](auto rc) { acc[decltype(ic)::value][decltype(jc)::value][decltype(rc)::value] = 0.f; });
        });
    });

    // fragment rows wm*MT*32 + i*32 + l31: all offsets are multiples of 16, so the swizzle key is lane-only
    const int sw = (l31 >> 1) & 7;
    // MFMAs of chunk `st`; when `more`, the DMA pieces of the NEXT chunk are issued one every PIECE_EVERY MFMAs
    // from the start of the chunk: they hide in the matrix pipe's shadow instead of idling it after the
    // barrier, and the rest of the chunk's MFMAs cover their flight time before the vmcnt(0).
    // (the skinny tiles have few MFMAs per chunk: their pieces spread over the whole chunk)
    constexpr int HALF_MFMAS = 2 * 4 * MT * NT, ALL_MFMAS = 2 * HALF_MFMAS;
    constexpr int PIECE_EVERY = HALF_MFMAS / NPIECE >= 2 ? (HALF_MFMAS / NPIECE < 4 ? HALF_MFMAS / NPIECE : 4) : ALL_MFMAS / NPIECE;
    static_assert(PIECE_EVERY >= 1 && PIECE_EVERY * NPIECE <= ALL_MFMAS, "DMA pieces must fit in the chunk");
    auto compute = [&](int st, bool more) {
        const float* a_base = As + st * A_STAGE + (wm * MT * 32 + l31) * BK;
        const float* b_base = Bs + st * B_STAGE + (wn * NT * 32 + l31) * BK;
        f32x4 a[2][MT], b[2][NT];
        auto fetch = [&](int s, int buf) {
            const int slot = ((2 * s + half) ^ sw) << 2;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[buf][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + slot);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[buf][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + slot);
        };
        fetch(0, 0);
        static_for<BK / 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < BK / 8) fetch(s + 1, (s + 1) & 1);  // fragments of sub-step s+1 under the MFMAs of s
            __builtin_amdgcn_sched_barrier(0);                        // (hipcc would sink the prefetch otherwise)
            static_for<4 * MT * NT>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int t = q / (MT * NT), i = (q / NT) % MT, j = q % NT;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i][t], b[s & 1][j][t], acc[i][j], 0, 0, 0);
                constexpr int g = s * 4 * MT * NT + q;  // MFMA index within the chunk
                if constexpr (g % PIECE_EVERY == PIECE_EVERY - 1 && g / PIECE_EVERY < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) dma_piece(std::integral_constant<int, g / PIECE_EVERY>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };

    // ---- main loop: LDS double buffer; the DMA of chunk c+1 flies under the MFMAs of chunk c
    if (c_begin < c_end) {
        chunk_src(c_begin, 0);
        static_for<NPIECE>([&](auto kc) { dma_piece(kc); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int st = 0;
        for (int ci = c_begin; ci < c_end; ++ci) {
            const bool more = ci + 1 < c_end;
            if (more) chunk_src(ci + 1, st ^ 1);
            compute(st, more);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            st ^= 1;
        }
    }

    // ---- epilogue
    if (p.partial == nullptr && !p.pool && !p.nchw && (p.Cout & 3) == 0 && (p.split_n & 3) == 0)
        conv_epilogue_lds<MT, NT, WM, WN>(p, acc, smem, mbase, ntile, wm, wn, l31, half, phase, tid);
    else
        conv_epilogue<MT, NT, BN>(p, acc, mbase, ntile, wm, wn, l31, half, phase, split);
}

template <int KH, int KW, int MT, int NT, int WM, int WN, bool PHASE>
static hipError_t launch_dma_cfg(const ConvArgs& a, int blocks, hipStream_t stream) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr size_t lds_loop = sizeof(float) * 2 * (BM + BN) * CONV_BK;
    constexpr size_t lds_epi = sizeof(float) * (WM * 32) * (BN + 4);  // conv_epilogue_lds staging tile
    constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_mfma_dma_kernel<KH, KW, MT, NT, WM, WN, PHASE>;
    static lds_once_mask configured{0};  // per-device bit mask
    if (hipError_t e = ensure_dynamic_lds(kern, lds, &configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, stream, a);
    return hipGetLastError();
}

template <int KH, int KW, bool PHASE>
static hipError_t launch_dma_tile(int BM, int BN, const ConvArgs& a, int blocks, hipStream_t stream) {
    if constexpr (KH == 7) {   // the key-point heads (K + 4 K logits: N = 64 or 128 per tile): two tiles are instantiated
        if (BM == 512 && BN == 64) return launch_dma_cfg<KH, KW, 2, 2, 8, 1, PHASE>(a, blocks, stream);
        if (BM == 256 && BN == 128) return launch_dma_cfg<KH, KW, 2, 2, 4, 2, PHASE>(a, blocks, stream);
        return hipErrorInvalidValue;
    }
    if (BM == 256 && BN == 256) return launch_dma_cfg<KH, KW, 2, 4, 4, 2, PHASE>(a, blocks, stream);
    if (BM == 256 && BN == 128) return launch_dma_cfg<KH, KW, 2, 2, 4, 2, PHASE>(a, blocks, stream);
    if (BM == 512 && BN == 64) return launch_dma_cfg<KH, KW, 2, 2, 8, 1, PHASE>(a, blocks, stream);
    if (BM == 32 && BN == 128) return launch_dma_cfg<KH, KW, 1, 1, 1, 4, PHASE>(a, blocks, stream);
    if (BM == 64 && BN == 128) return launch_dma_cfg<KH, KW, 2, 1, 1, 4, PHASE>(a, blocks, stream);
    return hipErrorInvalidValue;  // keep in sync with conv_dma_tile()
}

hipError_t conv_dma_launch_kernel(const ConvLayer& L, const ConvArgs& a, int blocks, hipStream_t stream) {
    if (L.phase) return launch_dma_tile<2, 2, true>(L.BM, L.BN, a, blocks, stream);
    if (L.kh == 3 && L.kw == 3) return launch_dma_tile<3, 3, false>(L.BM, L.BN, a, blocks, stream);
    if (L.kh == 7 && L.kw == 7) return launch_dma_tile<7, 7, false>(L.BM, L.BN, a, blocks, stream);
    return hipErrorInvalidValue;
}

}  // namespace eamm
