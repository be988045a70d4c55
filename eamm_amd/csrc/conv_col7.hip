// Column-patch kernel for the 7x1 (vertical) MFMA convolution behind the generator's final 7x7 convolution
// (reference modules/generator.py:47,92: Conv2d(64 -> 3, 7x7, pad 3) + sigmoid).
//
// The final layer runs as a 7x1 convolution with N = (dx, co) = 21 "output channels" followed by a horizontal gather
// (final_shift_sum_kernel, motion.hip).  The im2col-style kernel re-fetched its activation operand
// once per vertical tap -- 7 x 268 MB through L2 per launch at 256^2 x 16 frames, which bounds it (5.6 TB/s of
// L2 -> LDS traffic for 89 TFLOP/s).  Here a persistent workgroup walks 16x16-pixel tiles:
//   * per 32-channel chunk the (16+6) x 16 input patch is DMA'd into LDS once (44 KiB); vertical tap t of output pixel
//     (r, c) is patch pixel (r + t, c) = the same swizzled LDS image 2 KiB further on, so the seven taps' A fragments
//     cost no address arithmetic at all (an immediate offset);
//   * the whole weight tensor ([chunks][7][32][32] = 56 KiB for 64 input channels) stays resident in LDS;
//   * the patch of the next (tile, chunk) unit streams in behind the 112 MFMAs of the current one (two stages).
// Operand traffic drops from 7x to 22/16 = 1.4x the activation; the kernel is matrix-pipe bound (N = 21 of a 32-wide
// tile is the remaining waste).  Output: the [B,H,W,32] partial products the gather kernel reads.
#include "conv_common.h"

#include <algorithm>
#include <cstdlib>

namespace eamm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int CT = 16;                   // tile side
constexpr int CPR = CT + 6;              // patch rows (halo 3 above and below)
constexpr int CPIX = CPR * CT;           // 352 patch pixels = 44 DMA instructions of 8 pixels
constexpr int CWAVES = 8;
constexpr int CMAXCH = 2;                // channel chunks whose weights fit beside the two patch stages
}  // namespace

struct Col7Args {
    const float* in;       // [B,H,W,C]
    unsigned in_bytes, w_bytes;
    int C, B, H, W;
    int tiles_x, tiles_y, tiles;   // tiles per row / column / in total (B * tiles_y * tiles_x)
    const float* w;        // packed [C/32][7][32][32], LDS-DMA swizzle
    float* out;            // [B,H,W,32]   (plain form)
    const float* bias;     // [3]          (fused form)
    float* final_out;      // [B,3,H,W]    (fused form): sigmoid(7x7 convolution + bias)
    int dbg;               // diagnostic (EAMM_COL7_DBG, tools/final_layer_bench.py; wrong results): 1 no patch DMA in the loop, 2 no tile epilogue
};

// FUSED (round 3): the horizontal gather of the seven dx taps, the bias and the sigmoid happen in the tile's epilogue instead
// of in final_shift_sum_kernel -- the [B,H,W,32] partial products (65.5 MB written + read back per 8 frames at 256x256)
// never leave the CU.  Output column x needs the partial columns x-3 .. x+3, so a workgroup owns whole ROWS of tiles and
// walks them left to right: when tile T (columns c0 .. c0+15) is finished its partial products go to LDS (the patch stage
// that has just been consumed), the outputs of columns c0-3 .. c0+12 are formed from them and from the last six partial
// columns of tile T-1 (a 2 x 8 KB carry ring in the last 16 KB of LDS), and columns 10..15 become the next carry.

namespace {
constexpr int PS = 21;                         // partial products per pixel kept: (dx, co); odd stride: conflict-free column walks
constexpr int CARRY = CT * 6 * PS;             // six partial columns of a tile
}  // namespace

// Epilogue of a finished tile of the fused final layer (both MFMA forms): `scratch` holds the tile's partial products
// P[row][column][dx*3 + co]; forms the outputs of columns tx0-3 .. tx0+12 (.. tx0+15 for the row's last tile) from them and from
// the carry of the tile to the left, + bias, sigmoid, NCHW store; then saves columns 10..15 as the next tile's carry.
__device__ __forceinline__ void col7_gather_tile(const Col7Args& p, const float* scratch, float* Carry, const float (&bias_r)[3],
                                                 int b, int ty0, int tx0, int tid) {
    const int tix = tx0 / CT;
    const bool last = tix == p.tiles_x - 1;
    const float* carry_in = Carry + (tix & 1) * CARRY;          // columns tx0-6 .. tx0-1 (from tile T-1)
    float* carry_out = Carry + ((tix + 1) & 1) * CARRY;
    const int ncol = last ? CT + 3 : CT;                          // the row's last tile also finishes its last three columns
    // One thread per (row, output column): 16 rows x 32 column slots (16 or 19 used) = the 512 threads, the three
    // output channels in registers -- no integer division, no divergent tap loop (round 3 ran `i % ncol`, `i / ncol`
    // per output and a branch per tap: ~5 us of the ~17 us a tile takes; the stage is matrix-pipe time otherwise).
    // Tap dx of output column xl reads partial column c = xl - 6 + dx of this tile, of the carry (c < 0) or nothing.
    {
        const int xl = tid & 31, row = tid >> 5;
        const int x = tx0 - 3 + xl, y = ty0 + row;
        float v0 = bias_r[0], v1 = bias_r[1], v2 = bias_r[2];
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
            const int c = xl - 6 + dx;
            const bool in_tile = (unsigned)c < (unsigned)CT, in_carry = (c < 0) & (tix > 0);
            // clamped (always valid) addresses, values selected afterwards: the loads issue back to back
            const float* src = in_tile ? scratch + (row * CT + c) * PS + dx * 3
                                       : carry_in + (row * 6 + max(c, -6) + 6) * PS + dx * 3;
            const float* safe = (in_tile | in_carry) ? src : scratch;
            const float p0 = safe[0], p1 = safe[1], p2 = safe[2];
            const bool use = in_tile | in_carry;
            v0 += use ? p0 : 0.f;
            v1 += use ? p1 : 0.f;
            v2 += use ? p2 : 0.f;
        }
        if (xl < ncol && x >= 0 && x < p.W && y < p.H) {
            float* o = p.final_out + ((size_t)b * 3 * p.H + y) * p.W + x;
            const size_t plane = (size_t)p.H * p.W;
            o[0] = 1.f / (1.f + __expf(-v0));
            o[plane] = 1.f / (1.f + __expf(-v1));
            o[2 * plane] = 1.f / (1.f + __expf(-v2));
        }
    }
    for (int i = tid; i < CARRY; i += CWAVES * 64) {                // columns 10..15 -> the next tile's carry
        const int n = i % PS, t2 = i / PS;
        const int c6 = t2 % 6, row = t2 / 6;
        carry_out[i] = scratch[(row * CT + 10 + c6) * PS + n];
    }
}


// Tile walk of a column-patch workgroup.  A unit is a (tile, 32-channel chunk) pair; ROWS: the workgroup's items are tile ROWS
// (first + k * gridDim.x), each walked tile by tile from the left; otherwise single tiles.  The position is advanced
// incrementally: round 3 re-derived it from the unit index with integer divisions inside every DMA piece -- ~180 vector
// instructions per unit and wave in the MFMA stream (the f32 MFMA shares the vector pipe: PMC SQ_INSTS_VALU 296 per unit and
// wave against 112 MFMAs), a fifth of the kernel.
struct Col7Pos {
    int cc, tix, t, b, ty0, tx0;
};
template <bool ROWS>
__device__ __forceinline__ void col7_pos_decode(Col7Pos& s, const Col7Args& p) {
    if constexpr (ROWS) {
        s.b = s.t / p.tiles_y;
        s.ty0 = (s.t - s.b * p.tiles_y) * CT;
        s.tx0 = s.tix * CT;
    } else {
        const int tt = s.t / p.tiles_x;
        s.tx0 = (s.t - tt * p.tiles_x) * CT;
        s.b = tt / p.tiles_y;
        s.ty0 = (tt - s.b * p.tiles_y) * CT;
    }
}
template <bool ROWS>
__device__ __forceinline__ void col7_pos_next(Col7Pos& s, const Col7Args& p, int cchunks, int grid) {
    if (++s.cc < cchunks) return;
    s.cc = 0;
    if constexpr (ROWS) {
        if (++s.tix < p.tiles_x) {
            s.tx0 = s.tix * CT;
            return;
        }
        s.tix = 0;
    }
    s.t += grid;
    col7_pos_decode<ROWS>(s, p);     // (the divisions: once per tile row / tile, not per DMA piece)
}

// The patch pieces a lane issues: piece j = wave + 8 i (i = 0..5, j < 44) covers patch pixels 8 j .. 8 j + 7, a lane's pixel
// q = 8 j + lane / 8 = (row q / 16 - 3, column q % 16) of the tile and its 16-byte slot never change -- decoded once.
struct Col7Lane {
    int qr[6], qc[6], rel[6];
};
__device__ __forceinline__ void col7_lane_init(Col7Lane& L, const Col7Args& p, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = (wave + CWAVES * i) * 8 + (lane >> 3);
        L.qr[i] = (q >> 4) - 3;
        L.qc[i] = q & 15;
        L.rel[i] = (L.qr[i] * p.W + L.qc[i]) * p.C + (((lane & 7) ^ ((q >> 1) & 7)) << 2);
    }
}

template <bool FUSED>
__global__ __launch_bounds__(CWAVES * 64) void conv_col7_kernel(const Col7Args p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = CPIX * BK;             // floats (44 KiB)
    constexpr int W_TAP = 32 * BK;                 // one (chunk, tap) weight tile
    constexpr int A_PIECES = CPIX / 8;             // 44
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [cchunks*7][32][32] weights, [2][A_STAGE] patches, FUSED: [2][CARRY]
    const int cchunks = p.C / BK;
    float* const Ws = smem;
    float* const As = smem + cchunks * 7 * W_TAP;
    float* const Carry = As + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // unit u of this workgroup = (tile first + (u / cchunks) * gridDim.x, chunk u % cchunks); FUSED: the workgroup's items are
    // tile ROWS (first + k * gridDim.x), each walked tile by tile from the left
    const int first = blockIdx.x;
    const int items = FUSED ? p.B * p.tiles_y : p.tiles;
    const int my_items = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int units = my_items * (FUSED ? p.tiles_x : 1) * cchunks;
    Col7Lane L;
    col7_lane_init(L, p, wave, lane);
    // patch piece i (j = wave + 8 i) of the unit at `pos` into stage st
    auto dma_patch_piece = [&](auto ic, const Col7Pos& pos, int st) {
        constexpr int i = decltype(ic)::value;
        const int j = wave + CWAVES * i;
        const int y = pos.ty0 + L.qr[i], x = pos.tx0 + L.qc[i];
        const bool ok = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        const int base = ((pos.b * p.H + pos.ty0) * p.W + pos.tx0) * p.C + pos.cc * BK;       // wave-uniform
        const unsigned off = ok ? (unsigned)(base + L.rel[i]) * 4u : OOB;
        float* dst = As + st * A_STAGE + j * (8 * BK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsi, (lds_ptr_t)dst, 16, off, 0, 0, 0);
    };

    if (units == 0) return;
    // ---- prologue: all weights + the first patch
    for (int j = wave; j < cchunks * 7 * 4; j += CWAVES) {   // 1 KiB pieces of the weight tensor
        const unsigned off = (unsigned)(j * (8 * BK) + lane * 4) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(Ws + j * (8 * BK)), 16, off, 0, 0, 0);
    }
    Col7Pos cur{0, 0, first, 0, 0, 0};
    col7_pos_decode<FUSED>(cur, p);
    static_for<6>([&](auto ic) {
        if (wave + CWAVES * decltype(ic)::value < A_PIECES) dma_patch_piece(ic, cur, 0);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc;
    static_for<16>([&](auto rc) { acc[decltype(rc)::value] = 0.f; });

    // this lane's pixel inside the tile: wave w owns rows 2w, 2w+1
    const int trow = 2 * wave + (l31 >> 4), tcol = l31 & 15;
    const int idx0 = trow * CT + tcol;                                 // patch pixel of tap 0 (dy = -3)
    const int a_off = idx0 * BK + ((half ^ ((idx0 >> 1) & 7)) << 2);   // + 8*s for K step s via XOR; + tap * CT * BK
    const int b_off = l31 * BK + ((half ^ ((l31 >> 1) & 7)) << 2);

    float bias_r[3] = {0.f, 0.f, 0.f};
    if constexpr (FUSED) {
        bias_r[0] = p.bias[0]; bias_r[1] = p.bias[1]; bias_r[2] = p.bias[2];
    }
    for (int u = 0; u < units; ++u) {
        const int st = u & 1;
        const bool more = u + 1 < units;
        const float* a_stage = As + st * A_STAGE;
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0, cc = cur.cc;
        Col7Pos nxt = cur;
        col7_pos_next<FUSED>(nxt, p, cchunks, (int)gridDim.x);
        const float* w_stage = Ws + cc * 7 * W_TAP;
        // 7 taps x 4 K steps x 4 MFMAs; the next unit's patch pieces (wave, wave+8, ...: 5 or 6 per wave) ride along
        f32x4 a[2], bb[2];
        auto fetch = [&](auto tc, auto sc_, int buf) {
            constexpr int t = decltype(tc)::value, s = decltype(sc_)::value;
            a[buf] = *reinterpret_cast<const f32x4*>(a_stage + ((a_off ^ (8 * s)) + t * CT * BK));
            bb[buf] = *reinterpret_cast<const f32x4*>(w_stage + ((b_off ^ (8 * s)) + t * W_TAP));
        };
        fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        static_for<28>([&](auto gc) {
            constexpr int g = decltype(gc)::value;       // (tap, K step) index
            if constexpr (g + 1 < 28)
                fetch(std::integral_constant<int, (g + 1) / 4>{}, std::integral_constant<int, (g + 1) % 4>{}, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], bb[g & 1][q], acc, 0, 0, 0);
            });
            if constexpr (g % 4 == 1 && g / 4 < 6) {     // one patch piece after every 16 MFMAs, 6 slots
                __builtin_amdgcn_sched_barrier(0);
                if (more && wave + CWAVES * (g / 4) < A_PIECES && !(p.dbg & 1)) dma_patch_piece(std::integral_constant<int, g / 4>{}, nxt, st ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (cc == cchunks - 1 && !(p.dbg & 2)) {
            if constexpr (!FUSED) {
                // epilogue of the tile: lanes 0..31 of a wave hold the 32 channels of one pixel -> 128-byte rows
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;   // pixel of the wave's 32: row m / 16, column m % 16
                    const int y = ty0 + 2 * wave + (m >> 4), x = tx0 + (m & 15);
                    if (y < p.H && x < p.W) p.out[((size_t)(b * p.H + y) * p.W + x) * 32 + l31] = acc[r];
                    acc[r] = 0.f;
                });
            } else {
                // every wave is done with this unit's patch stage (the next unit reads the other one; its refill starts inside
                // the next unit): it becomes the scratch for the tile's partial products P[row][column][dx*3 + co]
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                float* const scratch = As + st * A_STAGE;
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int row = 2 * wave + (m >> 4), col = m & 15;
                    if (l31 < PS) scratch[(row * CT + col) * PS + l31] = acc[r];
                    acc[r] = 0.f;
                });
                __syncthreads();
                col7_gather_tile(p, scratch, Carry, bias_r, b, ty0, tx0, tid);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 4: the fused final layer on the MULTI-BLOCK MFMA v_mfma_f32_4x4x1_16b_f32.  N = (dx, co) = 21 of a
// 32-wide tile multiplies zeros in a third of its matrix-pipe time; the 4x4x1 form has an N granule of FOUR: with the
// A-block broadcast (cbsz = 4, abid = g) one instruction is D[i][lane p] += W[n = 4 g + i][k] * X[pixel p][k] -- 64 pixels
// (the lanes) x 4 columns x one k in 8 cycles, the same 64 FLOP / cycle / SIMD as the 32x32x2 form -- so six groups g cover
// n = 0..23: 24 / 21 = 14 % padding instead of 52 %, 0.75 x the matrix-pipe cycles.  A wave owns 64 pixels (4 rows of the
// 16x16 tile) and three of the six groups (waves 0-3: n 0..11, waves 4-7: n 12..23); per (tap, 16-byte slot of the channel
// chunk) it reads ONE ds_read_b128 of the patch (lane = pixel: its 4 k values) and ONE of the weights (lane = n: W[n][4 k]),
// and issues 4 k x 3 groups = 12 MFMAs on three rotating accumulators.  Same LDS images (weights [chunk][tap][32 n][32 k]
// swizzled, patch (16+6) x 16 pixels x 32 k swizzled), same DMA pipeline, same tile walk, same gather epilogue as above.
template <int NH, int EVERY, typename DmaPiece>
__device__ __forceinline__ void col7q_unit(const float* a_stage, const float* w_stage, int a_off, int b_off, f32x4 (&acc)[3],
                                           bool more, int wave, const Col7Pos& nxt, int st, int dbg, const DmaPiece& dma_patch_piece) {
    constexpr int BK = CONV_BK;
    constexpr int W_TAP = 32 * BK;
    constexpr int A_PIECES = CPIX / 8;
    f32x4 xa[2], wb[2];
    auto fetch = [&](auto tc, auto sc_, int buf) {
        constexpr int t = decltype(tc)::value, sl = decltype(sc_)::value;
        xa[buf] = *reinterpret_cast<const f32x4*>(a_stage + ((a_off ^ (4 * sl)) + t * CT * BK));
        wb[buf] = *reinterpret_cast<const f32x4*>(w_stage + ((b_off ^ (4 * sl)) + t * W_TAP));
    };
    fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
    static_for<56>([&](auto gc) {
        constexpr int g = decltype(gc)::value;       // (tap, slot) index: 7 taps x 8 slots of 4 channels
        if constexpr (g + 1 < 56)
            fetch(std::integral_constant<int, (g + 1) / 8>{}, std::integral_constant<int, (g + 1) % 8>{}, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            // srcA = weights (block abid broadcast to all sixteen blocks), srcB = the 64 pixels
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[g & 1][q], xa[g & 1][q], acc[0], 4, 3 * NH + 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[g & 1][q], xa[g & 1][q], acc[1], 4, 3 * NH + 1, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[g & 1][q], xa[g & 1][q], acc[2], 4, 3 * NH + 2, 0);
        });
        if constexpr (g % EVERY == EVERY / 2 - 1 && g / EVERY < 6) {     // one patch piece of the next unit every EVERY groups, 6 slots
            __builtin_amdgcn_sched_barrier(0);
            if (more && wave + CWAVES * (g / EVERY) < A_PIECES && !(dbg & 1)) dma_patch_piece(std::integral_constant<int, g / EVERY>{}, nxt, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

template <int EVERY>
__global__ __launch_bounds__(CWAVES * 64) void conv_col7q_kernel(const Col7Args p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = CPIX * BK;             // floats (44 KiB)
    constexpr int W_TAP = 32 * BK;
    constexpr int A_PIECES = CPIX / 8;             // 44
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [cchunks*7][32][32] weights, [2][A_STAGE] patches, [2][CARRY]
    const int cchunks = p.C / BK;
    float* const Ws = smem;
    float* const As = smem + cchunks * 7 * W_TAP;
    float* const Carry = As + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, nh = wave >> 2;       // pixel group (rows 4 pg .. 4 pg + 3), half of the six n groups

    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    const int first = blockIdx.x;
    const int items = p.B * p.tiles_y;             // tile ROWS, each walked tile by tile from the left
    const int my_items = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int units = my_items * p.tiles_x * cchunks;
    Col7Lane L;
    col7_lane_init(L, p, wave, lane);
    // patch piece i (j = wave + 8 i) of the unit at `pos` into stage st
    auto dma_patch_piece = [&](auto ic, const Col7Pos& pos, int st) {
        constexpr int i = decltype(ic)::value;
        const int j = wave + CWAVES * i;
        const int y = pos.ty0 + L.qr[i], x = pos.tx0 + L.qc[i];
        const bool ok = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        const int base = ((pos.b * p.H + pos.ty0) * p.W + pos.tx0) * p.C + pos.cc * BK;       // wave-uniform
        const unsigned off = ok ? (unsigned)(base + L.rel[i]) * 4u : OOB;
        float* dst = As + st * A_STAGE + j * (8 * BK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsi, (lds_ptr_t)dst, 16, off, 0, 0, 0);
    };

    if (units == 0) return;
    for (int j = wave; j < cchunks * 7 * 4; j += CWAVES) {   // 1 KiB pieces of the weight tensor
        const unsigned off = (unsigned)(j * (8 * BK) + lane * 4) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(Ws + j * (8 * BK)), 16, off, 0, 0, 0);
    }
    Col7Pos cur{0, 0, first, 0, 0, 0};
    col7_pos_decode<true>(cur, p);
    static_for<6>([&](auto ic) {
        if (wave + CWAVES * decltype(ic)::value < A_PIECES) dma_patch_piece(ic, cur, 0);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[3];
    static_for<3>([&](auto gc) { acc[decltype(gc)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; });

    // this lane's pixel inside the tile (srcB): row 4 pg + lane / 16, column lane % 16; tap 0 (dy = -3) is the same patch pixel
    const int idx0 = (4 * pg + (lane >> 4)) * CT + (lane & 15);
    const int a_off = idx0 * BK + (((idx0 >> 1) & 7) << 2);            // ^ 4 * slot; + tap * CT * BK (keeps (idx >> 1) & 7)
    // this lane's weight row (srcA): n = lane (only lanes 4 g .. 4 g + 3 of the broadcast block are read: n < 24)
    const int nrow = lane & 31;
    const int b_off = nrow * BK + (((nrow >> 1) & 7) << 2);
    float bias_r[3] = {p.bias[0], p.bias[1], p.bias[2]};

    for (int u = 0; u < units; ++u) {
        const int st = u & 1;
        const bool more = u + 1 < units;
        const float* a_stage = As + st * A_STAGE;
        const int b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0, cc = cur.cc;
        Col7Pos nxt = cur;
        col7_pos_next<true>(nxt, p, cchunks, (int)gridDim.x);
        const float* w_stage = Ws + cc * 7 * W_TAP;
        if (nh == 0)
            col7q_unit<0, EVERY>(a_stage, w_stage, a_off, b_off, acc, more, wave, nxt, st, p.dbg, dma_patch_piece);
        else
            col7q_unit<1, EVERY>(a_stage, w_stage, a_off, b_off, acc, more, wave, nxt, st, p.dbg, dma_patch_piece);
        if (cc == cchunks - 1 && !(p.dbg & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            float* const scratch = As + st * A_STAGE;     // the patch stage every wave has just finished reading
            float* dst = scratch + idx0 * PS + 12 * nh;   // P[row][column][n], n = 12 nh + 4 g + i
            static_for<3>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                static_for<4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if (12 * nh + 4 * g + i < PS) dst[4 * g + i] = acc[g][i];
                    acc[g][i] = 0.f;
                });
            });
            __syncthreads();
            col7_gather_tile(p, scratch, Carry, bias_r, b, ty0, tx0, tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Same tile geometry for the row-split flow head (7x1 over N = 7 x (K+2) = 84 of 96, two concatenated inputs, 128 input
// channels): the weights ([4][7][96][32] = 336 KiB) do not fit, so they stream one (chunk, tap) tile per barrier interval
// (NT x 4 KiB, double buffered) next to the two patch stages; one workgroup per 16x16 tile.
struct Col7sArgs {
    const float* in0;      // [B,H,W,C0]
    const float* in1;      // [B,H,W,C1] or null
    unsigned in0_bytes, in1_bytes, w_bytes;
    int C0, C1, B, H, W;
    int tiles_x, tiles_y;
    const float* w;        // packed [(C0+C1)/32][7][NT*32][32], LDS-DMA swizzle
    float* out;            // [B,H,W,PS]
    int PS;                // pixel stride of `out` in floats (>= NT*32)
};

template <int NT>
__global__ __launch_bounds__(CWAVES * 64) void conv_col7s_kernel(const Col7sArgs p) {
    constexpr int BK = CONV_BK;
    constexpr int A_STAGE = CPIX * BK;             // floats (44 KiB)
    constexpr int W_STAGE = NT * 32 * BK;          // one (chunk, tap) weight tile
    constexpr int A_PIECES = CPIX / 8;             // 44
    constexpr int W_PIECES = NT * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][A_STAGE] patches, [2][W_STAGE] weights
    float* const As = smem;
    float* const Ws = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int cchunks = (p.C0 + p.C1) / BK;
    const int nint = cchunks * 7;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (t % p.tiles_x) * CT;
    t /= p.tiles_x;
    const int ty0 = (t % p.tiles_y) * CT;
    const int b = t / p.tiles_y;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, p.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 : p.in0), 0, p.in1 ? p.in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    auto dma_patch_piece = [&](int j, int cc, int st) {
        const int q = j * 8 + (lane >> 3);
        const int y = ty0 + (q >> 4) - 3, x = tx0 + (q & 15);
        const bool ok = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        const int slot = ((lane & 7) ^ ((q >> 1) & 7)) << 2;
        const int c0 = cc * BK;
        const bool first = c0 < p.C0;
        const int C = first ? p.C0 : p.C1;
        const int coff = first ? c0 : c0 - p.C0;
        const unsigned off = ok ? (unsigned)(((b * p.H + y) * p.W + x) * C + coff + slot) * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)(As + st * A_STAGE + j * (8 * BK)), 16, off, 0, 0, 0);
    };
    auto dma_weight_piece = [&](int j, int it, int st) {   // piece j (0..W_PIECES-1) of interval it
        const unsigned off = (unsigned)((it * NT * 32 + j * 8) * BK + lane * 4) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(Ws + st * W_STAGE + j * (8 * BK)), 16, off, 0, 0, 0);
    };

    for (int j = wave; j < A_PIECES; j += CWAVES) dma_patch_piece(j, 0, 0);
    for (int j = wave; j < W_PIECES; j += CWAVES) dma_weight_piece(j, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[NT];
    static_for<NT>([&](auto jc) { static_for<16>([&](auto rc) { acc[decltype(jc)::value][decltype(rc)::value] = 0.f; }); });

    const int trow = 2 * wave + (l31 >> 4), tcol = l31 & 15;
    const int idx0 = trow * CT + tcol;
    const int a_off = idx0 * BK + ((half ^ ((idx0 >> 1) & 7)) << 2);
    const int b_off = l31 * BK + ((half ^ ((l31 >> 1) & 7)) << 2);

    int cc = 0, tap = 0;
    for (int it = 0; it < nint; ++it) {
        const bool more_w = it + 1 < nint, more_a = cc + 1 < cchunks;
        const float* a_stage = As + (cc & 1) * A_STAGE + tap * (CT * BK);
        const float* w_stage = Ws + (it & 1) * W_STAGE;
        f32x4 a[2], bb[2][NT];
        auto fetch = [&](auto sc_, int buf) {
            constexpr int s = decltype(sc_)::value;
            a[buf] = *reinterpret_cast<const f32x4*>(a_stage + (a_off ^ (8 * s)));
#pragma unroll
            for (int j = 0; j < NT; ++j) bb[buf][j] = *reinterpret_cast<const f32x4*>(w_stage + (b_off ^ (8 * s)) + j * 32 * BK);
        };
        fetch(std::integral_constant<int, 0>{}, 0);
        static_for<4>([&](auto stc) {
            constexpr int step = decltype(stc)::value;
            if constexpr (step + 1 < 4) fetch(std::integral_constant<int, step + 1>{}, (step + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4 * NT>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int tq = q / NT, j = q % NT;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][tq], bb[step & 1][j][tq], acc[j], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (step == 0) {          // next interval's weights: pieces wave, wave + 8, ...
                if (more_w)
                    for (int j = wave; j < W_PIECES; j += CWAVES) dma_weight_piece(j, it + 1, (it + 1) & 1);
            } else if constexpr (step == 1) {   // next chunk's patch: piece tap*8 + wave (44 pieces over the 7 taps)
                const int j = tap * CWAVES + wave;
                if (more_a && j < A_PIECES) dma_patch_piece(j, cc + 1, (cc + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (++tap == 7) {
            tap = 0;
            ++cc;
        }
    }
    static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int y = ty0 + 2 * wave + (m >> 4), x = tx0 + (m & 15);
            if (y < p.H && x < p.W) p.out[((size_t)(b * p.H + y) * p.W + x) * p.PS + j * 32 + l31] = acc[j][r];
        });
    });
}

hipError_t conv_col7s_launch(const float* in0, int C0, const float* in1, int C1, int B, int H, int W, const float* w_swizzled,
                             int ntile32, float* out, int pixel_stride, hipStream_t stream) {
    if (C0 % CONV_BK || C1 % CONV_BK || C0 < CONV_BK || ntile32 != 3 || pixel_stride < ntile32 * 32)
        return hipErrorInvalidValue;
    Col7sArgs a{};
    a.in0 = in0;
    a.in1 = C1 ? in1 : nullptr;
    const size_t px = (size_t)B * H * W;
    const size_t b0 = px * C0 * 4, b1 = px * C1 * 4, wb = (size_t)((C0 + C1) / CONV_BK) * 7 * ntile32 * 32 * CONV_BK * 4;
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.in0_bytes = (unsigned)b0;
    a.in1_bytes = (unsigned)b1;
    a.w_bytes = (unsigned)wb;
    a.C0 = C0;
    a.C1 = C1;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = (W + CT - 1) / CT;
    a.tiles_y = (H + CT - 1) / CT;
    a.w = w_swizzled;
    a.out = out;
    a.PS = pixel_stride;
    constexpr size_t lds = sizeof(float) * (2 * CPIX * CONV_BK + 2 * 3 * 32 * CONV_BK);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static lds_once_mask configured{0};
    if (hipError_t e = ensure_dynamic_lds(conv_col7s_kernel<3>, lds, &configured); e != hipSuccess) return e;
    note_mfma_flops(2.0 * a.tiles_x * a.tiles_y * B * (CT * CT) * 7.0 * (ntile32 * 32) * (C0 + C1));
    hipLaunchKernelGGL(conv_col7s_kernel<3>, dim3(a.tiles_x * a.tiles_y * B), dim3(CWAVES * 64), lds, stream, a);
    return hipGetLastError();
}

static hipError_t col7_launch_impl(const float* in, int C, int B, int H, int W, const float* w_swizzled, float* out, const float* bias,
                                   float* final_out, hipStream_t stream) {
    const bool fused = final_out != nullptr;
    if (C % CONV_BK || C / CONV_BK > CMAXCH || C < CONV_BK) return hipErrorInvalidValue;
    Col7Args a{};
    a.in = in;
    const size_t ib = (size_t)B * H * W * C * 4, wb = (size_t)(C / CONV_BK) * 7 * 32 * CONV_BK * 4;
    if (ib >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    a.in_bytes = (unsigned)ib;
    a.w_bytes = (unsigned)wb;
    a.C = C;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = (W + CT - 1) / CT;
    a.tiles_y = (H + CT - 1) / CT;
    a.tiles = B * a.tiles_x * a.tiles_y;
    a.w = w_swizzled;
    a.out = out;
    a.bias = bias;
    a.final_out = final_out;
#ifdef EAMM_EXPERIMENTS   // diagnostic builds only (make EXPERIMENTS=1): the product library never drops the DMA / the epilogue
    static const int dbg = (int)knob_int("EAMM_COL7_DBG", 0);
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    const size_t lds = wb + sizeof(float) * (2 * CPIX * CONV_BK + (fused ? 2 * CT * 6 * 21 : 0));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static lds_once_mask configured{0}, configured_fused{0};
    if (hipError_t e = fused ? ensure_dynamic_lds(conv_col7_kernel<true>, 160 * 1024, &configured_fused)
                             : ensure_dynamic_lds(conv_col7_kernel<false>, 160 * 1024, &configured);
        e != hipSuccess)
        return e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = std::min(fused ? B * a.tiles_y : a.tiles, cus);
    // EAMM_FINAL_MFMA4 = 0: the fused form on the 32x32x2 MFMA with N padded to 32 (round 3); default: the 4x4x1 multi-block form
    static const bool mfma4 = knob_int("EAMM_FINAL_MFMA4", 1) != 0;
    if (fused && mfma4) {
        // (one patch piece of the next unit per tap, six of the seven taps: issuing them twice as densely at the start of the
        // unit measured the same -- 210.4 vs 213.4 us at 8 frames, 3915 vs 3912 frames/s in the pipeline)
        static lds_once_mask configured_q{0};
        note_mfma_flops(2.0 * a.tiles * (CT * CT) * 7.0 * 24 * C);      // N = 24 columns issued
        if (hipError_t e = ensure_dynamic_lds(conv_col7q_kernel<8>, 160 * 1024, &configured_q); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_col7q_kernel<8>, dim3(blocks), dim3(CWAVES * 64), lds, stream, a);
        return hipGetLastError();
    }
    note_mfma_flops(2.0 * a.tiles * (CT * CT) * 7.0 * 32 * C);
    if (fused)
        hipLaunchKernelGGL(conv_col7_kernel<true>, dim3(blocks), dim3(CWAVES * 64), lds, stream, a);
    else
        hipLaunchKernelGGL(conv_col7_kernel<false>, dim3(blocks), dim3(CWAVES * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t conv_col7_launch(const float* in, int C, int B, int H, int W, const float* w_swizzled, float* out,
                            hipStream_t stream) {
    return col7_launch_impl(in, C, B, H, W, w_swizzled, out, nullptr, nullptr, stream);
}

hipError_t conv_col7_fused_launch(const float* in, int C, int B, int H, int W, const float* w_swizzled, const float* bias,
                                  float* out_nchw, hipStream_t stream) {
    if (!bias || !out_nchw) return hipErrorInvalidValue;
    return col7_launch_impl(in, C, B, H, W, w_swizzled, nullptr, bias, out_nchw, stream);
}

}  // namespace eamm
