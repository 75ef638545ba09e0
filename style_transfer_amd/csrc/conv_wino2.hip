// 3x3 convolution through 2-D Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// Same job and interface as conv_mfma_kernel (forward + bias + ReLU,
// backward-to-data + ReLU mask + loss-gradient terms, split-K partials) with 16 multiplies per
// 2x2 output tile and input channel instead of 36 (Lavin & Gray):
//     V = Bt d B   (d = 4x4 input patch at rows 2ty-1.., columns 2tx-1..)
//     U = G g Gt   (g = 3x3 filter)                      -- formed once, when the bank is packed
//     M[xi][nu] = sum over input channels of U[xi][nu] * V[xi][nu]
//     out = At M A (2x2)
//     Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//     At = [1 1 1 0; 0 1 -1 -1]
// The sixteen (xi, nu) components are sixteen independent GEMMs D[m][t] = sum_k A[m][k] B[k][t]
// with m = output channel, t = tile, k = input channel.
//
// Work split.  A workgroup of eight waves computes 64 channels x (4 rows x 64 columns) = 64 tiles.
// Wave (xi, trow) owns transform row xi for the 32 tiles of tile row trow and both 32-channel
// blocks: 2 x 4 (nu) accumulators of one v_mfma_f32_32x32x2_f32 block each, 128 registers, the
// same budget as the other kernels.  The A and B operands of the four nu components sit next to
// each other in LDS, so one ds_read_b128 feeds four MFMAs.  After the reduction the waves turn
// their nu components into two output columns in registers, exchange those through LDS, and
// each wave combines the four xi rows of a quarter of the channels into the two output rows.
//
// Staging.  Per chunk of 8 input channels: the transformed filter bank comes as a ready LDS
// image (32 KB, four b128 loads per thread); the input patch is transformed on its way in, one
// (channel, tile) per thread: sixteen loads -> Bt d B in registers -> four b128 LDS writes.  The
// chunk is double buffered in LDS (2 x 64 KB), one barrier per chunk (inside it, before its last
// k-step); the workgroup has the CU to itself.  DESIGN.md section 3.3 has the cycle budget.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef STX_W2_SKIP
#define STX_W2_SKIP 0   // timing experiments (tools/ubench/wino2_bench.hip): 1 no filter loads, 2 no patch
#endif                 // loads in the main loop.  Wrong results when non-zero.

namespace stx {

#ifdef STX_WINO2_STAMPS   // tools/ubench/wino2_bench.hip: when does every wave of every workgroup start and end, and on which CU
struct Wino2Stamp { unsigned long long t0, t1; unsigned hw, xcc; };
__device__ Wino2Stamp g_wino2_stamps[8192][8];
#endif
#ifdef STX_WINO2_TIMING   // cycle counters for tools/ubench/wino2_bench.hip
__device__ long long g_wino2_timing[8][8];
__device__ unsigned long long g_wino2_sums[8];   // over ALL workgroups (wave 0): see the end of the kernel
#define STX_T(var) const long long var = clock64()
#define STX_TW(var) const long long var = wall_clock64()
#else
#define STX_TW(var) [[maybe_unused]] const long long var = 0
#define STX_T(var) [[maybe_unused]] const long long var = 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));

namespace {

constexpr int KC = 8, BM = 64, NT = 512;
// Pixel patch of a workgroup = 64 tiles.  TXW = tiles per tile row within a wave's 32: 32 -> 4 rows x
// 64 columns (long row segments for loads and stores), 16 -> 8 x 32 pixels, 8 -> 16 x 16 pixels (planes whose width is
// not near a multiple of 64, e.g. the 91 x 91 / 46 x 46 planes of a 724-pixel tile).
template <int TXW>
struct Geo {
    static constexpr int TYW = 32 / TXW;          // tile rows per wave
    static constexpr int PR = 4 * TYW, PC = 2 * TXW;
};
constexpr int U_FLOATS = 4 * KC * BM * 4;     // [xi][ci][m][nu]
constexpr int V_FLOATS = 4 * KC * 64 * 4;     // [xi][ci][tile][nu]
constexpr int STAGE = U_FLOATS + V_FLOATS;    // 64 KB
constexpr size_t kLdsBytes = 2 * STAGE * sizeof(float) + 64;     // + the waves' maxima (y_amax)

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic of this wave complete, then the workgroup barrier.  Unlike __syncthreads() this does
// not wait for the global loads in flight for the chunk after next (vmcnt is left alone).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

// BIG: plane sets of 2 GiB and more (a 64-channel blob of a tile larger than 2896 x 2896).  The
// kernel addresses memory through buffer descriptors with 32-bit offsets; the BIG variant moves
// the descriptor's 64-bit base instead of growing the offset -- to the chunk's 8 input planes for
// the loads, to the output channel for everything the epilogue touches -- so that every offset
// stays inside a few planes (8 planes must stay under 2 GiB: tiles up to 8192 x 8192).  Same
// arithmetic in the same order: results are bit-identical to the plain variant's.
// MK: ReLU sign nibbles (ConvProblem::in_codes / mask_codes).  Forward: the kernel writes those of
// its input; backward: the mask comes as nibbles (one byte per lane and channel) instead of the
// fp32 blob.  (A template parameter, not a run-time test: the forward kernels that do not emit --
// the 512-channel layers, whose backward pass is matrix-bound and gains nothing -- were 1 % slower
// with the untaken branch in their main loop.)
template <int EPI, int TXW, bool BIG = false, bool MK = false>
__global__ __launch_bounds__(NT) void conv_wino2_kernel(WinoArgs a) {
    constexpr int TYW = Geo<TXW>::TYW, PR = Geo<TXW>::PR, PC = Geo<TXW>::PC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef STX_WINO2_STAMPS
    const unsigned long long stamp0 = wall_clock64();
#endif
    STX_T(t_start);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int xi = wave & 3, trow = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_mfma.hip
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int kslice = sgpr(EPI == kEpiPartial ? L % a.ksplit : 0);
    const int Lt = sgpr(EPI == kEpiPartial ? a.item_base + L / a.ksplit : L);
    // (channel tile slowest, so that an XCD keeps one filter slice in L2, measured no different)
    // Eight channel tiles: the 32 workgroups an XCD runs at a time take 4 channel tiles x 8
    // patches instead of 8 x 4 -- per round 8.4 MB of filters + 6.5 MB of input through the L2
    // instead of 16.8 + 3.2 (512 -> 512 channels; tools/pmc_layers.py)  [common.h: wino2_item_tiles]
    int ptile, mtile;
    wino2_item_tiles(Lt, m_tiles, a.tiles_x * a.tiles_y, ptile, mtile);
    ptile = sgpr(ptile);
    const int c_begin = sgpr(EPI == kEpiPartial ? kslice * a.n_chunks / a.ksplit : 0);
    const int c_end = sgpr(EPI == kEpiPartial ? (kslice + 1) * a.n_chunks / a.ksplit : a.n_chunks);
    const int y0 = sgpr((ptile / a.tiles_x) * PR);
    const int x0 = sgpr((ptile % a.tiles_x) * PC);
    const int m0 = mtile * BM;
    const int HW = a.H * a.W;

    constexpr unsigned kOob = 0x80000000u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, BIG ? 0 : a.x_bytes, 0x00020000);
    // BIG: the 8 input planes of one chunk (fewer in the last one; what lies behind reads as zero)
    auto x_rebase = [&](int chunk) __attribute__((always_inline)) {
        const int k0 = sgpr(chunk * KC);
        const int nk = a.K - k0 < KC ? a.K - k0 : KC;
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + (size_t)k0 * (size_t)HW, 0,
                                               (nk > 0 ? nk : 0) * HW * 4, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- staging roles: this thread transforms the patch of channel `wave` of the chunk for
    // tile `lane` (tile row lane / 32, tile column lane % 32).  The four columns of a patch row
    // are one 16-byte load (dword aligned; neighbouring lanes overlap by half, which the texture
    // unit coalesces).  Rows outside the plane get an offset beyond the descriptor's range and
    // read as zero, so do channels past K.  Columns outside the plane are zeroed when the values
    // are consumed, by workgroups on the left / right edge only: x = -1 is the first patch column
    // of tile column 0, x >= W can only be one of the last two patch columns of a tile that
    // still has a column inside (tiles entirely outside compute garbage nobody stores).
    // The one patch row whose x = -1 would lie before the start of the tensor (channel 0, row 0
    // of the first workgroup) is loaded from x = 0 and shifted by one instead.
    const int st_x = x0 + 2 * (lane % TXW) - 1;           // first patch column
    const bool left = st_x < 0;
    const bool corner = wave == 0 && y0 == 0 && x0 == 0;  // uniform: lane 0, patch row 1
    const bool corner_lane = lane == 0;
    unsigned xvoff[4];
    {
        const int ty = lane / TXW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = y0 + 2 * ty - 1 + i;
            const int off = wave * HW + yy * a.W + st_x;
            xvoff[i] = (unsigned)yy < (unsigned)a.H && st_x < a.W
                           ? (unsigned)(off < 0 ? 0 : off) * 4u : kOob;
        }
    }
    // ReLU sign nibbles of the INPUT (forward, first channel tile only): this thread's patch holds
    // the 2x2 window at rows 2ty, 2ty+1 / columns 2tx, 2tx+1 in its middle -- patch rows 1, 2,
    // columns 1, 2 -- so the layer that consumes a rectified blob leaves, for the price of ten
    // vector instructions per chunk, what its own backward pass needs of that blob: one byte per
    // window instead of 16 bytes of fp32 mask per lane and channel (conv1_2's backward fetched
    // 908 MB, 268 of them conv1_1 read only for its signs).
    const int cph = (a.H + 1) >> 1, cpw = (a.W + 1) >> 1;
    const bool emit = EPI == kEpiForward && MK && mtile == 0;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        a.in_codes, 0, emit ? a.K * cph * cpw : 0, 0x00020000);
    unsigned cvoff = kOob;
    {
        const int ty = lane / TXW, tx = lane % TXW;
        if (y0 + 2 * ty < a.H && x0 + 2 * tx < a.W)
            cvoff = (unsigned)(wave * cph * cpw + ((y0 >> 1) + ty) * cpw + (x0 >> 1) + tx);
    }
    const bool edge_l = x0 == 0, edge_r = x0 + PC + 2 > a.W;      // workgroup-uniform
    const bool ok2 = st_x + 2 < a.W, ok3 = st_x + 3 < a.W;
    const unsigned w_base = (unsigned)(mtile * a.w_tile_stride) * 4u;
    constexpr unsigned w_chunk = (unsigned)U_FLOATS * 4u;
    const unsigned x_chunk = (unsigned)(KC * HW) * 4u;
    // LDS byte offsets of this thread's writes within a buffer
    const unsigned u_dst = (unsigned)tid * 16u;
    const unsigned v_dst = (unsigned)(U_FLOATS + (wave * 64 + lane) * 4) * 4u;

    u32x4 wreg[4];
    f32x4 xreg[4];
    f32x2 tq[4][2];    // Bt d, two columns at a time
    f32x4 vq[4];       // Bt d B, one transform row each

    // The hand-over of a chunk from the staging registers to LDS, cut into single-instruction
    // pieces.  Memory and LDS instructions ride in the 64-cycle shadow of an MFMA for free (one
    // per MFMA: tools/ubench/solo_issue.hip, coissue.hip), so those pieces are dealt out one per
    // MFMA; vector instructions are never hidden, the first one after an MFMA costs ~13 cycles
    // and every further one of the same burst ~4, so Bt d B is ONE burst: sixteen packed adds
    // (v_pk_add_f32 with operand-select / negate modifiers, which the compiler does not form by
    // itself); the results stay in their own registers until they are written.
#define STX_PK(dst, a_, b_, mods) asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(dst) : "v"(a_), "v"(b_))
    auto u_load = [&](int n, unsigned ws) {
        wreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(tid + n * NT) * 16u, ws, 0);
    };
    auto x_load = [&](int i, unsigned xs) {
        xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[i], xs, 0));
    };
    auto u_write = [&](int n, char *ldsb) {
        *reinterpret_cast<u32x4 *>(ldsb + u_dst + n * (NT * 16)) = wreg[n];
    };
    auto fix_edges = [&]() {
        // The asm keeps the loaded values opaque until here: otherwise the compiler forms the
        // differences right behind the loads, with a vmcnt wait in the wrong place.
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xreg[i]));
        // ONE uniform branch (workgroups on the left / right border of the plane), issued right
        // behind an MFMA so that the instruction-fetch bubble of the jump falls into its shadow;
        // inside, the selects are unconditional
        if (edge_l || edge_r) {
            asm volatile("");      // keeps this a (scalar) branch
            if (corner) {
                asm volatile("");  // one wave in the whole launch takes it
                // plain floats: element assignments through a select turn into dynamic indexing
                const float r0 = xreg[1].x, r1 = xreg[1].y, r2 = xreg[1].z, r3 = xreg[1].w;
                xreg[1].y = corner_lane ? r0 : r1;
                xreg[1].z = corner_lane ? r1 : r2;
                xreg[1].w = corner_lane ? r2 : r3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xreg[i].x = left ? 0.f : xreg[i].x;
                xreg[i].z = ok2 ? xreg[i].z : 0.f;
                xreg[i].w = ok3 ? xreg[i].w : 0.f;
            }
        }
    };
    auto emit_codes = [&](int chunk) {     // (after fix_edges: what lies outside the plane is zero)
        // x > 0 for a float is bits > 0 for its pattern read as a signed integer (-0.0 and the
        // negatives are negative integers): clamp(bits, 0, 1) is one v_med3_i32 per element,
        // seven vector instructions per nibble -- they are not hidden behind the fp32 MFMAs
        auto pos = [](float v) __attribute__((always_inline)) {
            const int b = __builtin_bit_cast(int, v);
            return (unsigned)min(max(b, 0), 1);
        };
        const unsigned nib = pos(xreg[1].y) | (pos(xreg[1].z) << 1) | (pos(xreg[2].y) << 2) |
                             (pos(xreg[2].z) << 3);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, rin, cvoff,
                                             (unsigned)sgpr(chunk * KC * cph * cpw), 0);
    };
    // rows: t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3 on column pair h
    auto row_op = [&](int q) {
        const int h = q >> 2, which = q & 3;
        const f32x2 d0 = h ? xreg[0].zw : xreg[0].xy, d1 = h ? xreg[1].zw : xreg[1].xy;
        const f32x2 d2 = h ? xreg[2].zw : xreg[2].xy, d3 = h ? xreg[3].zw : xreg[3].xy;
        if (which == 0) STX_PK(tq[0][h], d0, d2, "neg_lo:[0,1] neg_hi:[0,1]");
        if (which == 1) STX_PK(tq[1][h], d1, d2, "");
        if (which == 2) STX_PK(tq[2][h], d2, d1, "neg_lo:[0,1] neg_hi:[0,1]");
        if (which == 3) STX_PK(tq[3][h], d1, d3, "neg_lo:[0,1] neg_hi:[0,1]");
    };
    // columns, with P = (t[.][0], t[.][1]) and Q = (t[.][2], t[.][3]) of transform row x:
    //   (v0, v1) = (P.x - Q.x, P.y + Q.x)      (v2, v3) = (Q.x - P.y, P.y - Q.y)
    auto col_op = [&](int q) {
        const int x = q >> 1;
        f32x2 r;
        if ((q & 1) == 0) {
            STX_PK(r, tq[x][0], tq[x][1], "op_sel_hi:[1,0] neg_lo:[0,1]");
            vq[x].xy = r;
        } else {
            STX_PK(r, tq[x][0], tq[x][1], "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]");
            vq[x].zw = r;
        }
    };
    auto v_write = [&](int x, char *ldsb) {
        *reinterpret_cast<f32x4 *>(ldsb + v_dst + x * (KC * 64 * 16)) = vq[x];
    };
    auto load_stage = [&](int chunk) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * w_chunk));
        const unsigned xs = BIG ? 0u : (unsigned)sgpr((int)((unsigned)chunk * x_chunk));
        if (BIG) x_rebase(chunk);
#pragma unroll
        for (int n = 0; n < 4; ++n) u_load(n, ws);
#pragma unroll
        for (int i = 0; i < 4; ++i) x_load(i, xs);
    };
    auto store_stage = [&](int buf) {
        char *ldsb = reinterpret_cast<char *>(lds) + buf * (STAGE * 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) u_write(n, ldsb);
        fix_edges();
        if (emit) emit_codes(c_begin);
#pragma unroll
        for (int q = 0; q < 8; ++q) row_op(q);
#pragma unroll
        for (int q = 0; q < 8; ++q) col_op(q);
#pragma unroll
        for (int x = 0; x < 4; ++x) v_write(x, ldsb);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // operand addresses of k-step q (channels 2q and 2q+1; lane half h supplies channel 2q + h)
    const int a_off = ((xi * KC + half) * BM + l31) * 4;
    const int b_off = U_FLOATS + ((xi * KC + half) * 64 + trow * 32 + l31) * 4;
    constexpr int NS = KC / 2;

    // One chunk of matrix work out of LDS buffer `cur`, with (STORE) the hand-over of the next
    // chunk into the other buffer and (LOAD) the loads of the chunk after that dealt out between
    // the MFMAs (one piece behind each of MFMA 0..31): 0-3 filter-bank writes, 4-7 filter-bank
    // loads, 8 border fix-up, 9 Bt d B (one burst), 10-13 patch writes, 16-19 patch loads; the
    // operand reads of the next k-step ride behind the first three MFMAs of a k-step.  The
    // hand-over barrier sits before MFMA 24 (start of the last k-step).  sched_barrier pins the
    // order.
    f32x4 av[2][2], bv[2];
    auto run_chunk = [&](int cur, int chunk, auto store_c, auto load_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
        const float *base = lds + cur * STAGE;
        char *ldsb = reinterpret_cast<char *>(lds) + (cur ^ 1) * (STAGE * 4);
        unsigned ws = 0, xs = 0;
        if (LOAD) {
            ws = (unsigned)sgpr((int)(w_base + (unsigned)(chunk + 2) * w_chunk));
            xs = BIG ? 0u : (unsigned)sgpr((int)((unsigned)(chunk + 2) * x_chunk));
            if (BIG) x_rebase(chunk + 2);
        }
        const float *next = lds + (cur ^ 1) * STAGE;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = c * 2 + i, p = s * 8 + m;
                    if (STORE && p == 24) {
                        // every wave has written its share of the next chunk (pieces < 14) and
                        // issued its last reads of this one (during k-step 2)
                        __builtin_amdgcn_sched_barrier(0);
                        lds_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i][c], bv[s & 1][c],
                                                                      acc[i][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 3 && (s + 1 < NS || STORE)) {
                        // operands of the next k-step -- after the last one, of the next chunk's
                        // first, out of the other buffer (behind the barrier above)
                        const float *src = s + 1 < NS ? base : next;
                        const int o = s + 1 < NS ? (2 * (s + 1)) * 64 * 4 : 0;   // BM == 64 tiles: same stride
                        if (m == 0) av[(s + 1) & 1][0] = *reinterpret_cast<const f32x4 *>(src + a_off + o);
                        if (m == 1) av[(s + 1) & 1][1] = *reinterpret_cast<const f32x4 *>(src + a_off + o + 32 * 4);
                        if (m == 2) bv[(s + 1) & 1] = *reinterpret_cast<const f32x4 *>(src + b_off + o);
                    }
                    if (STORE) {
                        if (p < 4) u_write(p, ldsb);
                        // Bt d B as ONE burst of vector work: an isolated vector instruction
                        // between two MFMAs costs the matrix pipe ~13 cycles, the members of a
                        // burst ~4 each (tools/ubench/solo_issue.hip)
                        if (p == 8) {
                            fix_edges();
                            if (emit) emit_codes(chunk + 1);
                        }
                        if (p == 9) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) row_op(q);
#pragma unroll
                            for (int q = 0; q < 8; ++q) col_op(q);
                        }
                        if (p >= 10 && p < 14) v_write(p - 10, ldsb);
                    }
                    if (LOAD && !(STX_W2_SKIP & 1)) {
                        if (p >= 4 && p < 8) u_load(p - 4, ws);
                    }
                    if (LOAD && !(STX_W2_SKIP & 2)) {
                        if (p >= 16 && p < 20) x_load(p - 16, xs);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    STX_T(t_loads);       // index setup done, first loads about to be issued
    load_stage(c_begin);
    // clear the accumulators while the first loads are in flight (left alone the compiler sinks the
    // 128 moves to just before the first MFMA, behind the barrier: ~1000 cycles of an idle pipe)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(acc[i][c]));
    store_stage(0);
    STX_T(t_stored);      // first chunk landed, transformed and written to LDS
    if (c_begin + 1 < c_end) load_stage(c_begin + 1);
    lds_barrier();        // (not __syncthreads(): that would wait for the loads just issued)

    // Cycle counters (tools/ubench/wino2_bench.hip -DSTX_WINO2_TIMING): a chunk takes ~4360 cycles
    // of a SIMD against 4096 of pure matrix work (two waves x 32 MFMAs); the difference is the
    // vector work of the two waves (never hidden behind fp32 MFMAs, tools/ubench/solo_issue.hip)
    // and the hand-over.  Measured in round 1 and no faster: s_setprio (only swaps which wave of
    // a pair runs ahead), padding with s_nop to one MFMA per 128 cycles, alternating priority
    // per k-step, strict ping-pong with two barriers, a persistent variant (work items from a
    // ticket counter, first loads of the next item issued before the epilogue of the current
    // one: the ~4 us a workgroup spends outside its chunk loop are its own prologue and
    // epilogue, not the turnaround of a CU: 0.2-0.4 us, tools/ubench/wg_turnaround.hip).
    int cur = 0;
    int chunk = c_begin;
    [[maybe_unused]] long long t_work = 0;
    STX_T(t_begin);
    STX_TW(w_begin);
    // stx_clock_marks: the workgroup in the middle of the launch times its chunk loop with both
    // counters (a uniform branch; nothing is read when the pointer is null)
    const bool mark = a.clock_out != nullptr && (int)blockIdx.x == (int)(gridDim.x >> 1);
    long long mark_c = 0, mark_w = 0;
    if (mark) {
        mark_c = clock64();
        mark_w = wall_clock64();
    }
    // two chunks per trip: the LDS buffer index is a constant in each half, so every LDS address
    // of the hand-over and of the operand reads is a register plus an immediate
    // k-step 0 operands of the first chunk; from then on every chunk leaves those of its
    // successor behind (the hand-over barrier sits inside the chunk, before k-step 3)
    {
        const float *base = lds;
        av[0][0] = *reinterpret_cast<const f32x4 *>(base + a_off);
        av[0][1] = *reinterpret_cast<const f32x4 *>(base + a_off + 32 * 4);
        bv[0] = *reinterpret_cast<const f32x4 *>(base + b_off);
    }
    for (; chunk + 3 < c_end; chunk += 2) {
        STX_T(t0);
        run_chunk(0, chunk, yes{}, yes{});
        run_chunk(1, chunk + 1, yes{}, yes{});
        STX_T(t1);
        t_work += t1 - t0;
    }
    for (; chunk + 2 < c_end; ++chunk) {
        run_chunk(cur, chunk, yes{}, yes{});
        cur ^= 1;
    }
    STX_T(t_main_end);
    STX_TW(w_main_end);
    if (mark) {
        const long long dc = clock64() - mark_c, dw = wall_clock64() - mark_w;
        if (tid == 0) {       // (loops shorter than a microsecond say nothing about the clock)
            a.clock_out[0] = dw >= 100 ? dc : 0;
            a.clock_out[1] = dw >= 100 ? dw : 0;
        }
    }
    if (chunk + 1 < c_end) {
        run_chunk(cur, chunk, yes{}, no{});
        cur ^= 1;
        ++chunk;
    }
    // The backward epilogues read up to two arrays of the output's size (ReLU mask, style term):
    // 128 KB per workgroup, and the 256 workgroups of a round ask for theirs at the same moment.
    // Those reads are requested BEFORE the last chunk of matrix work and land during it (the
    // staging registers are free by then; with the pointer-arithmetic epilogue of round 1 this
    // spilled).  The forward epilogue only reads a bias vector and is set up after the chunk
    // (measured: 3-5 us slower per layer the other way round).
    constexpr bool kEarly = EPI == kEpiDgrad || EPI == kEpiDgradInject;
    if (!kEarly) {
        run_chunk(cur, chunk, no{}, no{});
        lds_barrier();
    }

    float s_scale = 0.f, c_scale = 0.f;
    if (EPI == kEpiDgradInject) {
        const float n = (float)((size_t)a.M * HW);
        if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / n + kEps));
        if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / n + kEps));
    }
    // ---- epilogue.  nu -> two output columns in registers; xi -> two output rows across waves:
    //   nu:  (c0 + c1 + c2,  c1 - c2 - c3)          xi:  (p0 + p1 + p2,  p1 - p2 - p3)
    // After the exchange this wave finishes accumulator registers 4*xi .. 4*xi+3 of both channel
    // blocks (D register r of a block is channel (r & 3) + 8 * (r >> 2) + 4 * half) for the 32
    // tiles of its tile row.  Both stages work on PAIRS of neighbouring registers (two channels)
    // with v_pk_add_f32, and -- on even plane widths -- everything the epilogue touches in
    // memory goes through buffer descriptors whose range check does the predication (a lane
    // outside the plane, or a channel past M, carries an out-of-range offset: loads return 0,
    // stores are dropped): no divergent branch, no 64-bit address arithmetic, one vector offset
    // per output row plus one scalar offset per channel.  (The form with pointer arithmetic and
    // a branch per store issued ~2.5 x as many vector instructions; tools/asm_mix.py.)
    const bool weven = (a.W & 1) == 0;       // pairs never straddle the end of a row
    const int yy = y0 + 2 * (trow * TYW + l31 / TXW), xx0 = x0 + 2 * (l31 % TXW);
    const unsigned plane_bytes = (unsigned)a.M * (unsigned)HW * 4u;
    const unsigned HW4 = (unsigned)HW * 4u;
    unsigned vo[2][2];                        // [row][column] of the lane's 2 x 2 outputs
    {
        const unsigned lane_base = (unsigned)((4 * half) * HW + yy * a.W + xx0) * 4u;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                vo[y][e] = (yy + y < a.H && xx0 + e < a.W) ? lane_base + (unsigned)(y * a.W + e) * 4u
                                                           : kOob;
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.y + (EPI == kEpiPartial ? (size_t)kslice * a.M * HW : 0), 0, BIG ? 0 : (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.mask), 0, a.mask && !BIG ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.inj.sgrad), 0, a.inj.sgrad && !BIG ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rft = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.inj.feat), 0, a.inj.feat && !BIG ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.bias), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.mask_codes), 0, MK ? a.M * cph * cpw : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t roc = __builtin_amdgcn_make_buffer_rsrc(
        a.out_codes, 0, a.out_codes ? a.M * cph * cpw : 0, 0x00020000);
    const unsigned vmc = (yy < a.H && xx0 < a.W)
                             ? (unsigned)((4 * half) * cph * cpw + (yy >> 1) * cpw + (xx0 >> 1))
                             : kOob;
    // BIG: a descriptor on channel c of an [M][H][W] array -- the lane's own channel is c or c + 4
    // (its half), so five planes are in reach; what lies past channel M - 1 is out of range
    auto at_channel = [&](const float *base, const __amdgpu_buffer_rsrc_t &whole, int c)
                          __attribute__((always_inline)) {
        if (!BIG) return whole;
        const int left = a.M - c < 5 ? a.M - c : 5;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base) + (size_t)c * (size_t)HW, 0,
                                                 base && left > 0 ? left * HW * 4 : 0, 0x00020000);
    };
    const int ph = (a.H + 1) >> 1, pw = a.W >> 1;
    const __amdgpu_buffer_rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc(
        a.pool_out, 0, a.pool_out ? a.M * ph * pw * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcodes = __builtin_amdgcn_make_buffer_rsrc(
        a.pool_codes, 0, a.pool_codes ? a.M * ph * pw : 0, 0x00020000);
    const unsigned vpool = (yy < a.H && xx0 < a.W)
                               ? (unsigned)((4 * half) * ph * pw + (yy >> 1) * pw + (xx0 >> 1)) * 4u
                               : kOob;
    // channel of output (i, rr) on the lower lane half, clamped to M: a scalar offset must not
    // exceed the descriptor's range (the check is offset >= num_records - soffset)
    const int M_ = a.M;
    auto chan = [&](int i, int rr) __attribute__((always_inline)) {
        const int c = m0 + i * 32 + rr + 8 * xi;
        return sgpr(c < M_ ? c : M_);
    };
    // even plane widths: a lane's two columns are one aligned 8-byte access; odd widths: two
    // dword accesses (a pair may straddle the end of a row)
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t &rs, int y, unsigned so, auto even_c) __attribute__((always_inline)) {
        if (decltype(even_c)::value)
            return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo[y][0], so, 0));
        f32x2 v;
        v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[y][0], so, 0));
        v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[y][1], so, 0));
        return v;
    };
    auto st2 = [&](const __amdgpu_buffer_rsrc_t &rs, int y, unsigned so, f32x2 v, auto even_c)
                   __attribute__((always_inline)) {
        if (decltype(even_c)::value) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), rs, vo[y][0], so, 0);
        } else {
            const float v0 = v.x, v1 = v.y;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs, vo[y][0], so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs, vo[y][1], so, 0);
        }
    };
    // The ReLU mask and the style term, for all sixteen outputs of the lane (see kEarly above).
    f32x2 mk[MK ? 1 : 16], sg[16];
    unsigned mkb[MK ? 8 : 1];      // MK: the 2x2 window's sign nibble per channel
    float bs[8];
    if (EPI == kEpiForward) {
        if (a.bias) {
#pragma unroll
            for (int n = 0; n < 8; ++n)
                bs[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rbias, (unsigned)half * 16u,
                                                      (unsigned)chan(n >> 2, n & 3) * 4u, 0));
        }
    } else if (EPI != kEpiPartial) {
        auto prefetch = [&](auto even_c) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int c = chan(n >> 3, (n >> 1) & 3);
                const unsigned so = BIG ? 0u : (unsigned)c * HW4;
                if (MK) {
                    if ((n & 1) == 0)
                        mkb[n >> 1] = __builtin_amdgcn_raw_buffer_load_b8(rmc, vmc, (unsigned)(c * cph * cpw), 0);
                } else if (a.mask) {
                    mk[n] = ld2(at_channel(a.mask, rmask, c), n & 1, so, even_c);
                }
                if (EPI == kEpiDgradInject) {
                    if (a.inj.sgrad) sg[n] = ld2(at_channel(a.inj.sgrad, rsg, c), n & 1, so, even_c);
                }
            }
        };
        if (weven) prefetch(yes{});
        else prefetch(no{});
    }

    if (kEarly) {
        run_chunk(cur, chunk, no{}, no{});
        lds_barrier();
    }

    // exchange: [wave][i][register pair q][lane] x (col 0 of r, col 0 of r + 1, col 1 of r, col 1
    // of r + 1), 128 KB in all
    f32x4 *ex = reinterpret_cast<f32x4 *>(lds);
#define STX_PK_ADD(dst, a_, b_) STX_PK(dst, a_, b_, "")
#define STX_PK_SUB(dst, a_, b_) STX_PK(dst, a_, b_, "neg_lo:[0,1] neg_hi:[0,1]")
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x2 c[4], t, u, o0, o1;
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = f32x2{acc[i][k][2 * q], acc[i][k][2 * q + 1]};
            STX_PK_ADD(t, c[0], c[1]);
            STX_PK_ADD(o0, t, c[2]);
            STX_PK_SUB(u, c[1], c[2]);
            STX_PK_SUB(o1, u, c[3]);
            ex[(wave * 16 + i * 8 + q) * 64 + lane] = f32x4{o0.x, o0.y, o1.x, o1.y};
        }
    __syncthreads();

    {
        const float *const content = a.inj.content;
        const int cw_ch = a.inj.win.ch, cw_cw = a.inj.win.cw,
                  cw_oy = a.inj.win.oy - a.inj.win.sy,
                  cw_ox = a.inj.win.ox - a.inj.win.sx;
        // common.h: content_index -- the wrapped row / column of the lane's 2 x 2 outputs in the
        // full-image content map, once per lane (two integer divisions per OUTPUT were ~1300
        // vector instructions of this epilogue on the content layer)
        int crow[2] = {0, 0}, ccol[2] = {0, 0};
        if (EPI == kEpiDgradInject && content) {
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const bool ok = yy + y < a.H && xx0 < a.W;
                int r = (cw_oy + (ok ? yy + y : 0)) % cw_ch;
                crow[y] = (r < 0 ? r + cw_ch : r) * cw_cw;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool ok = yy < a.H && xx0 < a.W;
                const int x = ok ? (xx0 + e < a.W ? xx0 + e : xx0) : (e && 1 < a.W ? 1 : 0);
                int r = (cw_ox + x) % cw_cw;
                ccol[e] = r < 0 ? r + cw_cw : r;
            }
        }
        // max |y| of what this lane stores, for an fp16-split convolution that reads the blob next
        // (ConvProblem::y_amax; conv_h2.hip)
        const bool track = EPI != kEpiPartial && a.y_amax != nullptr;
        float amax = 0.f;
        auto tail = [&](auto even_c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                f32x4 p[4];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    p[x] = ex[((trow * 4 + x) * 16 + i * 8 + 2 * xi + qq) * 64 + lane];
                // rows[y][e] = (channel a, channel b) of output row y, column e
                f32x2 rows[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x2 p0 = e ? p[0].zw : p[0].xy, p1 = e ? p[1].zw : p[1].xy;
                    const f32x2 p2 = e ? p[2].zw : p[2].xy, p3 = e ? p[3].zw : p[3].xy;
                    f32x2 t, u;
                    STX_PK_ADD(t, p0, p1);
                    STX_PK_ADD(rows[0][e], t, p2);
                    STX_PK_SUB(u, p1, p2);
                    STX_PK_SUB(rows[1][e], u, p3);
                }
                if (EPI == kEpiForward && a.bias) {
                    const f32x2 bb = {bs[i * 4 + 2 * qq], bs[i * 4 + 2 * qq + 1]};
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int e = 0; e < 2; ++e) STX_PK_ADD(rows[y][e], rows[y][e], bb);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {           // the two channels of the register pair
                    const int rr = 2 * qq + h, c = chan(i, rr);
                    const unsigned so = BIG ? 0u : (unsigned)c * HW4;
                    const __amdgpu_buffer_rsrc_t ry_c = at_channel(a.y, ry, c);
                    f32x2 o[2];
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        const int n = (i * 4 + rr) * 2 + y;
                        f32x2 v = {h ? rows[y][0].y : rows[y][0].x, h ? rows[y][1].y : rows[y][1].x};
                        if (EPI == kEpiForward) {
                            if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                        } else if (EPI != kEpiPartial) {
                            if (MK) {
                                const unsigned nib = mkb[n >> 1] >> (2 * y);
                                v.x = (nib & 1u) ? v.x : 0.f;
                                v.y = (nib & 2u) ? v.y : 0.f;
                            } else if (a.mask) {
                                v.x = mk[n].x > 0.f ? v.x : 0.f;
                                v.y = mk[n].y > 0.f ? v.y : 0.f;
                            }
                            if (EPI == kEpiDgradInject) {
                                if (content) {
                                    // (one layer per tile evaluation takes this: the content map
                                    // is read where it is used, not ahead of time)
                                    const f32x2 ft = ld2(at_channel(a.inj.feat, rft, c), y, so, even_c);
                                    const int mm = c + 4 * half;
                                    const int cm = mm < a.M ? mm : 0;     // (lanes past M store nothing)
                                    const float *cp = content + (size_t)cm * cw_ch * cw_cw + crow[y];
                                    v.x += c_scale * (ft.x - cp[ccol[0]]);
                                    v.y += c_scale * (ft.y - cp[ccol[1]]);
                                }
                                if (a.inj.sgrad) {
                                    v.x += s_scale * sg[n].x;
                                    v.y += s_scale * sg[n].y;
                                }
                            }
                        }
                        o[y] = v;
                        if (track)
                            amax = fmaxf(amax, fmaxf(vo[y][0] != kOob ? fabsf(v.x) : 0.f,
                                                     vo[y][1] != kOob ? fabsf(v.y) : 0.f));
                        if (EPI != kEpiForward || !a.skip_y) st2(ry_c, y, so, v, even_c);
                    }
                    // the ReLU sign nibble of the lane's window, for the backward pass of the layer that
                    // reads this blob (ConvProblem::out_codes; never with the BIG variants)
                    if (EPI == kEpiForward && !BIG && a.out_codes) {
                        const unsigned nib = (vo[0][0] != kOob && o[0].x > 0.f ? 1u : 0u) |
                                             (vo[0][1] != kOob && o[0].y > 0.f ? 2u : 0u) |
                                             (vo[1][0] != kOob && o[1].x > 0.f ? 4u : 0u) |
                                             (vo[1][1] != kOob && o[1].y > 0.f ? 8u : 0u);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, roc, vmc, (unsigned)(c * cph * cpw), 0);
                    }
                    // the lane's 2x2 outputs are exactly one window of the 2x2/2 pooling layer
                    // that follows (ceil mode: the second row may be missing): pool.hip's
                    // arithmetic (the divisor is 4 or 2: the product with its reciprocal is the
                    // same float)
                    if (EPI == kEpiForward && a.pool_out) {
                        const bool hy = yy + 1 < a.H;
                        float pr;
                        if (a.pool_mode == STX_POOL_MAX) {
                            pr = fmaxf(o[0].x, o[0].y);
                            pr = hy ? fmaxf(fmaxf(pr, o[1].x), o[1].y) : pr;
                        } else {
                            pr = (o[0].x + o[0].y + (hy ? o[1].x : 0.f) + (hy ? o[1].y : 0.f)) *
                                 (hy ? 0.25f : 0.5f);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pr), rpool, vpool,
                                                              (unsigned)c * (unsigned)(ph * pw * 4), 0);
                        if (a.pool_codes) {       // what the backward pass needs of this window
                            const unsigned code =
                                a.pool_mode == STX_POOL_MAX
                                    ? pool_max_code(o[0].x, o[0].y, o[1].x, o[1].y, true, hy)
                                    : pool_ave_code(o[0].x, o[0].y, o[1].x, o[1].y, true, hy);
                            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)code, rcodes, vpool >> 2,
                                                                 (unsigned)c * (unsigned)(ph * pw), 0);
                        }
                    }
                }
            }
        };
        if (weven) tail(yes{});
        else tail(no{});
        if (track) {          // one atomic per workgroup
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
            float *wmax = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + kLdsBytes - 64);
            if (lane == 0) wmax[wave] = amax;
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 8; ++w) amax = fmaxf(amax, wmax[w]);
                atomicMax(a.y_amax + (blockIdx.x & (kAmaxSlots - 1)), __builtin_bit_cast(unsigned, amax));
            }
        }
    }
#undef STX_PK_ADD
#undef STX_PK_SUB
#ifdef STX_WINO2_STAMPS
    if (lane == 0 && blockIdx.x < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wino2_stamps[blockIdx.x][wave] = Wino2Stamp{stamp0, (unsigned long long)wall_clock64(), hw, xcc};
    }
#endif
#ifdef STX_WINO2_TIMING
    if (blockIdx.x == gridDim.x - 3 && lane == 0) {     // a workgroup of the last round
        g_wino2_timing[wave][0] = t_work, g_wino2_timing[wave][2] = w_main_end - w_begin;   // 100 MHz ticks
        g_wino2_timing[wave][3] = t_main_end - t_begin;
        g_wino2_timing[wave][4] = t_begin - t_start;
        g_wino2_timing[wave][6] = t_loads - t_start, g_wino2_timing[wave][7] = t_stored - t_loads;
        g_wino2_timing[wave][5] = clock64() - t_main_end;
    }
    if (wave == 0 && lane == 0) {      // sums over every workgroup of the launch
        const long long t_end = clock64();
        atomicAdd(&g_wino2_sums[0], 1ull);
        atomicAdd(&g_wino2_sums[1], (unsigned long long)(t_loads - t_start));     // index setup
        atomicAdd(&g_wino2_sums[2], (unsigned long long)(t_stored - t_loads));    // first chunk: loads -> LDS
        atomicAdd(&g_wino2_sums[3], (unsigned long long)(t_begin - t_stored));    // hand-over before the loop
        atomicAdd(&g_wino2_sums[4], (unsigned long long)(t_main_end - t_begin));  // chunk loop (all but the last two)
        atomicAdd(&g_wino2_sums[5], (unsigned long long)(t_end - t_main_end));    // last two chunks + epilogue
        atomicAdd(&g_wino2_sums[6], (unsigned long long)(t_end - t_start));
        atomicAdd(&g_wino2_sums[7], (unsigned long long)(wall_clock64() - w_begin));   // 100 MHz ticks from loop start
    }
#endif
}

// ------------------------------------------------------------------------------------------------
ConvConfig wino2_config(int geometry) {
    ConvConfig c;
    c.id = 200 + geometry;            // ids >= 200 mark the 2-D Winograd configurations
    c.bm = BM;
    c.kc = KC;
    c.pr = geometry == 1 ? Geo<8>::PR : geometry == 2 ? Geo<16>::PR : Geo<32>::PR;
    c.pc = geometry == 1 ? Geo<8>::PC : geometry == 2 ? Geo<16>::PC : Geo<32>::PC;
    c.threads = NT;
    c.lds_bytes = kLdsBytes;
    return c;
}

// 16 x 16 patches for narrow planes, where a 64-pixel-wide patch is mostly padding (measured:
// a 256 x 256 tile 1.50 -> 1.36 ms); from 46 pixels up the wide patch is as fast or faster
// (its loads and stores are long row segments, and the smaller workgroup count of the square
// geometry buys nothing once the K split has evened out the last round).  Both geometries
// compute every output with the same arithmetic in the same order: the choice never changes
// a result.
int wino2_pick_geometry(int H, int W) {
    if (W <= 40) return 1;
    // 8 x 32 patches where they need at least a tenth fewer workgroups than 4 x 64 ones (the 91 x 91
    // planes of a 724-pixel tile: 36 instead of 46 per channel tile) -- the third geometry the
    // four-wave kernel has had since round 2, now also for the layers only this kernel runs (the
    // loss-injecting ones).  STX_WINO2_GEO8X32=0 keeps the two-geometry rule.
    const char *env = sw_env("STX_WINO2_GEO8X32");
    if (env && atoi(env) == 0) return 0;
    const long wide = (long)ceil_div(H, Geo<32>::PR) * ceil_div(W, Geo<32>::PC);
    const long mid = (long)ceil_div(H, Geo<16>::PR) * ceil_div(W, Geo<16>::PC);
    return mid * 10 <= wide * 9 ? 2 : 0;
}

// K split of a launch.  One workgroup per CU and ~6 us of prologue + epilogue per workgroup make
// the small-tile rule of conv_mfma.hip wrong here (it sliced the 8 chunks of a 64-channel layer
// in two: 0.167 ms where the unsplit launch takes 0.11).  The factor minimises a cost model of
// whole rounds of 256 workgroups -- chunks x 2.05 us + 6 us each -- plus the reduce pass over
// the slices; it depends on the shape only.
static double wino2_round_us(int chunks) { return chunks * 2.05 + 6.0; }

// (cost in model microseconds, 0 slices = not applicable)
static Wino2Tail wino2_tail_plan(const ConvConfig &cfg, const ConvProblem &p, double *cost, bool any_epilogue = false) {
    const Wino2Tail none{0, 1};
    if (cfg.id < 200 || cfg.id >= 210) return none;      // (the eight-wave kernel only)
    if (p.epilogue != kEpiForward && p.epilogue != kEpiDgrad) return none;
    // the fused pooling and the ReLU nibbles belong to the unsplit epilogue
    if (!any_epilogue && (p.pool_out || p.wants_codes || p.in_codes || p.mask_codes)) return none;
    const char *env = sw_env("STX_WINO2_TAIL");      // (=0: off)
    if (env && atoi(env) == 0) return none;
    const int n_chunks = ceil_div(p.K, KC);
    const long n = (long)ceil_div(p.M, BM) * ceil_div(p.H, cfg.pr) * ceil_div(p.W, cfg.pc);
    const int q = (int)(n / 256), r = (int)(n % 256);
    if (q < 1 || r == 0) return none;
    const int f = std::min({8, 256 / r, n_chunks / 4});
    if (f < 2) return none;
    const double patch_mb = 4e-6 * BM * cfg.pr * cfg.pc;
    // whole rounds + the short round + the reduce pass over r patches + two more launches behind a
    // drained GPU: 3 us each (fitted to the layers of a 724-pixel tile: 175 / 178 / 106 / 98 us measured
    // where this gives 176 / 179 / 103 / 101)
    *cost = q * wino2_round_us(n_chunks) + wino2_round_us(ceil_div(n_chunks, f)) +
            (f + 1) * r * patch_mb / 3.0 + 5.0 + 2 * 3.0;
    return Wino2Tail{r, f};
}

// Uniform K split, or none: {factor, model cost}
static int wino2_uniform_plan(const ConvConfig &cfg, const ConvProblem &p, double *cost) {
    *cost = 0;
    if (p.epilogue != kEpiForward && p.epilogue != kEpiDgrad) return 1;
    const int n_chunks = ceil_div(p.K, KC);
    const long n = (long)ceil_div(p.M, BM) * ceil_div(p.H, cfg.pr) * ceil_div(p.W, cfg.pc);
    const double out_mb = 4e-6 * p.M * (double)p.H * p.W;
    double best_cost = 0;
    int best = 1;
    for (int f = 1; f <= 8 && n_chunks / f >= 4; ++f) {
        const double rounds = (double)ceil_div((int)std::min<long>(n * f, 1 << 30), 256);
        double c = rounds * ((double)n_chunks / f * 2.05 + 6.0);
        if (f > 1) c += (f + 1) * out_mb / 3.0 + 5.0;      // reduce pass at ~3 TB/s + its launch
        if (f == 1 || c < best_cost) {
            best_cost = c;
            best = f;
        }
    }
    *cost = best_cost;
    return best;
}

static Wino2Tail wino2_tail_choice(const ConvConfig &cfg, const ConvProblem &p, bool any_epilogue) {
    double tail_cost = 0, uniform_cost = 0;
    const Wino2Tail t = wino2_tail_plan(cfg, p, &tail_cost, any_epilogue);
    if (t.items == 0) return t;
    wino2_uniform_plan(cfg, p, &uniform_cost);
    return tail_cost < uniform_cost ? t : Wino2Tail{0, 1};
}

Wino2Tail wino2_tail_split(const ConvConfig &cfg, const ConvProblem &p) { return wino2_tail_choice(cfg, p, false); }

int wino2_splitk_factor(const ConvConfig &cfg, const ConvProblem &p) {
    if (wino2_tail_split(cfg, p).items) return 1;      // (the launch slices its last items itself)
    double c;
    return wino2_uniform_plan(cfg, p, &c);
}

// The most plane-sized slices either split of this shape can ask for (the scratch buffer is sized
// before the caller has attached pooling or ReLU-nibble outputs to the problem).
int wino2_max_slices(const ConvConfig &cfg, const ConvProblem &p) {
    double c;
    return std::max(wino2_uniform_plan(cfg, p, &c), wino2_tail_choice(cfg, p, true).slices);
}

size_t wino2_packed_floats(int K, int M) {
    return (size_t)ceil_div(M, BM) * ceil_div(K, KC) * U_FLOATS;
}

// packed[mt][chunk][xi][ci][mm][nu] = (G g Gt)[xi][nu] of the filter W(m = mt*64 + mm, k = chunk*8 + ci)
__global__ void wino2_pack_kernel(const float *__restrict__ w, int Mo, int Ko, int transpose_flip,
                                  int M, int K, int n_chunks, float *__restrict__ packed,
                                  size_t total) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int nu = idx % 4;
        const int mm = (idx / 4) % BM;
        const int ci = (idx / (4 * BM)) % KC;
        const int x = (idx / (4 * BM * KC)) % 4;
        const int chunk = (idx / U_FLOATS) % n_chunks;
        const int mt = idx / ((size_t)U_FLOATS * n_chunks);
        const int m = mt * BM + mm, k = chunk * KC + ci;
        float v = 0.f;
        if (m < M && k < K) {
            // rows of G: [1 0 0], [.5 .5 .5], [.5 -.5 .5], [0 0 1]
            float row[3];   // (G g)[x][b]
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float g[3];
#pragma unroll
                for (int aa = 0; aa < 3; ++aa) {
                    const int t = aa * 3 + b;
                    g[aa] = transpose_flip ? w[((size_t)k * Ko + m) * 9 + (8 - t)]
                                           : w[((size_t)m * Ko + k) * 9 + t];
                }
                row[b] = x == 0 ? g[0]
                       : x == 1 ? (g[0] + g[1] + g[2]) * 0.5f
                       : x == 2 ? (g[0] - g[1] + g[2]) * 0.5f
                                : g[2];
            }
            v = nu == 0 ? row[0]
              : nu == 1 ? (row[0] + row[1] + row[2]) * 0.5f
              : nu == 2 ? (row[0] - row[1] + row[2]) * 0.5f
                        : row[2];
        }
        packed[idx] = v;
    }
}

int wino2_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                       float *packed) {
    const int M = transpose_flip ? Ko : Mo;
    const int K = transpose_flip ? Mo : Ko;
    const size_t total = wino2_packed_floats(K, M);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    wino2_pack_kernel<<<blocks, 256, 0, s>>>(w_caffe, Mo, Ko, transpose_flip, M, K,
                                             ceil_div(K, KC), packed, total);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// The eight-wave kernel writes / reads ReLU sign nibbles unless the launch is sliced along K.
// STX_RELU_CODES=0 (stx_reread_env after a change) keeps the fp32 masks, for A/B measurements and tests.
bool conv_uses_relu_codes(const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    const char *env = sw_env("STX_RELU_CODES");
    if (env && atoi(env) == 0) return false;
    if (cfg.id < 200 || cfg.id >= 210 || ksplit > 1) return false;
    const double xb = 4.0 * p.K * (double)p.H * p.W, yb = 4.0 * p.M * (double)p.H * p.W;
    if (xb >= 2147483648.0 || yb >= 2147483648.0) return false;
    const char *force_big = sw_env("STX_WINO_BIG");
    if (force_big && atoi(force_big) == 1) return false;
    return p.epilogue == kEpiForward ? p.in_codes != nullptr
                                     : p.epilogue == kEpiDgrad && p.mask_codes != nullptr;
}

// The unsplit eight-wave fp32 kernel and the unsplit fp16-split kernel leave the sign nibbles of their
// (rectified) output.  Mirrors wino2_launch / h2_launch: no K slices, no tail split, no re-based addressing.
bool conv_writes_out_codes(const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    const char *env = sw_env("STX_RELU_CODES");
    if (env && atoi(env) == 0) return false;
    if (p.epilogue != kEpiForward || !p.relu || !p.out_codes || ksplit > 1) return false;
    if (cfg.id >= 300) return true;
    if (cfg.id < 200 || cfg.id >= 210) return false;
    const double xb = 4.0 * p.K * (double)p.H * p.W, yb = 4.0 * p.M * (double)p.H * p.W;
    if (xb >= 2147483648.0 || yb >= 2147483648.0) return false;
    const char *force_big = sw_env("STX_WINO_BIG");
    if (force_big && atoi(force_big) == 1) return false;
    const bool mk = conv_uses_relu_codes(cfg, p, 1);
    const Wino2Tail tail = wino2_tail_split(cfg, p);
    const bool tail_taken = tail.items && p.splitk_ws &&
                            p.splitk_ws_floats >= (size_t)tail.slices * p.M * p.H * p.W && !mk && !wino2_fuses_pool(p);
    return !tail_taken;
}

// The forward epilogue pools only on its float2 path (even rows, 8-byte aligned arrays).
bool wino2_fuses_pool(const ConvProblem &p) {
    return p.pool_out && p.epilogue == kEpiForward && (p.W & 1) == 0 &&
           (((size_t)p.y | (size_t)p.pool_out) & 7) == 0;
}

template <int EPI, int TXW, bool BIG = false, bool MK = false>
static int wino2_launch_epi(hipStream_t s, const WinoArgs &args, int n_wg) {
    auto kern = conv_wino2_kernel<EPI, TXW, BIG, MK>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(lds=%zu): %s", kLdsBytes, hipGetErrorString(e));
        return STX_ERR_HIP;
    }
    kern<<<n_wg, NT, kLdsBytes, s>>>(args);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

static int wino2_launch_tail(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, const WinoArgs &whole,
                             int epi, const Wino2Tail &tail);

int wino2_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    WinoArgs a;
    a.x = p.x;
    a.w = p.w;
    a.y = p.y;
    a.bias = p.bias;
    a.mask = p.mask;
    a.K = p.K;
    a.M = p.M;
    a.H = p.H;
    a.W = p.W;
    a.n_chunks = ceil_div(p.K, KC);
    a.tiles_x = ceil_div(p.W, cfg.pc);
    a.tiles_y = ceil_div(p.H, cfg.pr);
    a.m_tiles = ceil_div(p.M, BM);
    a.ksplit = 1;
    a.w_tile_stride = a.n_chunks * U_FLOATS;
    a.relu = p.relu;
    a.inj = p.inject;
    a.pool_out = nullptr;
    a.pool_codes = nullptr;
    a.pool_mode = p.pool_mode;
    const double xb = 4.0 * p.K * (double)p.H * p.W;
    const double wb = 4.0 * (double)wino2_packed_floats(p.K, p.M);
    const double yb = 4.0 * p.M * (double)p.H * p.W;       // the epilogue addresses the output planes
    // through descriptors too.  2 GiB and more: the BIG variant, which re-bases its descriptors per
    // chunk / per channel and needs 8 planes (and the pooled output, if fused) under 2 GiB.
    // STX_WINO_BIG=1 forces it on every plane (tests: results must not change).
    const char *force_big = sw_env("STX_WINO_BIG");
    const bool huge = xb >= 2147483648.0 || yb >= 2147483648.0;
    if (wb >= 2147483648.0 || 32.0 * (double)p.H * p.W >= 2147483648.0) {
        set_error("wino2_launch: a %d x %d plane is beyond the buffer-addressing limit", p.H, p.W);
        return STX_ERR_UNSUPPORTED;
    }
    a.w_bytes = (int)wb;
    const bool inject = p.epilogue == kEpiDgrad && (p.inject.sgrad || p.inject.content);
    int n_wg = a.m_tiles * a.tiles_x * a.tiles_y;
    // (planes that large never need a K split; the slices' kernel has no BIG form)
    const bool split = ksplit > 1 && p.splitk_ws && !huge &&
                       p.splitk_ws_floats >= (size_t)ksplit * p.M * p.H * p.W;
    // the tail split (below) likewise: its slices run the same kernel
    const Wino2Tail tail = !split && !huge && ksplit == 1 ? wino2_tail_split(cfg, p) : Wino2Tail{0, 1};
    const bool tail_ok = tail.items && p.splitk_ws &&
                         p.splitk_ws_floats >= (size_t)tail.slices * p.M * p.H * p.W;
    const bool big = huge || (force_big && atoi(force_big) == 1 && !split && !tail_ok);
    a.x_bytes = big ? 0 : (int)xb;
    // ReLU sign nibbles (see ConvProblem): written by the plain forward kernel, read by the plain
    // backward kernels; K slices and the BIG variants keep the fp32 mask
    const bool codes = conv_uses_relu_codes(cfg, p, split ? ksplit : 1) && !big;
    a.clock_out = p.clock_out;
    a.y_amax = p.y_amax;              // (K slices: the reduce pass leaves it, splitk_reduce_args)
    a.in_codes = codes && p.epilogue == kEpiForward ? p.in_codes : nullptr;
    a.mask_codes = codes && p.epilogue == kEpiDgrad ? p.mask_codes : nullptr;
    const bool mk = a.mask_codes != nullptr || a.in_codes != nullptr;
    // nibbles of the output: by the unsplit kernel only (conv_writes_out_codes below says the same)
    a.out_codes = !split && !big && !(tail_ok && !mk && !wino2_fuses_pool(p)) && p.epilogue == kEpiForward && p.relu
                      ? p.out_codes : nullptr;
    if (split) {
        a.ksplit = ksplit;
        a.y = p.splitk_ws;
        a.y_amax = nullptr;
        n_wg *= ksplit;
    } else if (p.epilogue == kEpiForward && wino2_fuses_pool(p)) {
        a.pool_out = p.pool_out;
        a.pool_codes = p.pool_codes;
        a.skip_y = p.skip_y && p.pool_codes != nullptr;
    }
    const int epi = split ? kEpiPartial : inject ? kEpiDgradInject : p.epilogue;
    if (tail_ok && !big && !mk && !a.pool_out) return wino2_launch_tail(s, cfg, p, a, epi, tail);
#define STX_W2_CASE(E)                                                                            \
    case E:                                                                                       \
        if (big && E != kEpiPartial)                                                              \
            STX_TRY(cfg.id == 201   ? (wino2_launch_epi<E, 8, true>(s, a, n_wg))                  \
                    : cfg.id == 202 ? (wino2_launch_epi<E, 16, true>(s, a, n_wg))                 \
                                    : (wino2_launch_epi<E, 32, true>(s, a, n_wg)));               \
        else if (mk && (E == kEpiForward || E == kEpiDgrad || E == kEpiDgradInject))              \
            STX_TRY(cfg.id == 201   ? (wino2_launch_epi<E, 8, false, true>(s, a, n_wg))           \
                    : cfg.id == 202 ? (wino2_launch_epi<E, 16, false, true>(s, a, n_wg))          \
                                    : (wino2_launch_epi<E, 32, false, true>(s, a, n_wg)));        \
        else                                                                                      \
            STX_TRY(cfg.id == 201   ? (wino2_launch_epi<E, 8>(s, a, n_wg))                        \
                    : cfg.id == 202 ? (wino2_launch_epi<E, 16>(s, a, n_wg))                       \
                                    : (wino2_launch_epi<E, 32>(s, a, n_wg)));                     \
        break;
    switch (epi) {
        STX_W2_CASE(kEpiForward)
        STX_W2_CASE(kEpiDgrad)
        STX_W2_CASE(kEpiDgradInject)
        STX_W2_CASE(kEpiPartial)
        default:
            set_error("wino2_launch: no kernel for epilogue %d", p.epilogue);
            return STX_ERR_UNSUPPORTED;
    }
#undef STX_W2_CASE
    return split ? splitk_reduce_launch(s, p, ksplit) : STX_OK;
}

// The tail split of a launch (common.h: wino2_tail_split): whole rounds as they are, then the
// remaining items as K slices in ONE short round, then the slices of those items' patches added
// up with the epilogue the unsplit kernel would have applied.
static int wino2_launch_tail(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, const WinoArgs &whole,
                             int epi, const Wino2Tail &tail) {
    const int n_items = whole.m_tiles * whole.tiles_x * whole.tiles_y;
    const int n_full = n_items - tail.items;
    WinoArgs part = whole;
    part.ksplit = tail.slices;
    part.item_base = n_full;
    part.y = p.splitk_ws;
    part.clock_out = nullptr;
    part.y_amax = nullptr;
#define STX_W2_TAIL(E)                                                                            \
    case E:                                                                                       \
        STX_TRY(cfg.id == 201   ? (wino2_launch_epi<E, 8>(s, whole, n_full))                      \
                : cfg.id == 202 ? (wino2_launch_epi<E, 16>(s, whole, n_full))                     \
                                : (wino2_launch_epi<E, 32>(s, whole, n_full)));                   \
        break;
    switch (epi) {
        STX_W2_TAIL(kEpiForward)
        STX_W2_TAIL(kEpiDgrad)
        STX_W2_TAIL(kEpiDgradInject)
        default:
            set_error("wino2_launch: no tail split for epilogue %d", epi);
            return STX_ERR_UNSUPPORTED;
    }
#undef STX_W2_TAIL
    const int n_part = tail.items * tail.slices;
    STX_TRY(cfg.id == 201   ? (wino2_launch_epi<kEpiPartial, 8>(s, part, n_part))
            : cfg.id == 202 ? (wino2_launch_epi<kEpiPartial, 16>(s, part, n_part))
                            : (wino2_launch_epi<kEpiPartial, 32>(s, part, n_part)));
    return splitk_reduce_items_launch(s, p, cfg, tail.slices, n_full, tail.items);
}

// ------------------------------------------------------------------------------------------------
// The Winograd configurations of the library (cfg.id >= 200: this kernel; >= 300: conv_h2.hip), by id.
// (Until round 6 two more fp32 families lived here -- 1-D F(2,3), ids 100+, and the four-wave form of this
// kernel, ids 210+: reachable only through switches once conv_h2 took the 3x3 layers; they are kept under
// tools/experiments/ with their tests' history, and this kernel is the library's one fp32 Winograd family.)
size_t wino_packed_floats(const ConvConfig &cfg, int K, int M) {
    return cfg.id >= 300 ? h2_packed_floats(K, M) : wino2_packed_floats(K, M);
}

int wino_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                      const ConvConfig &cfg, float *packed) {
    if (cfg.id >= 300) return h2_pack_weights(s, w_caffe, Mo, Ko, transpose_flip, packed);
    return wino2_pack_weights(s, w_caffe, Mo, Ko, transpose_flip, packed);
}

// Launches a Winograd configuration; `w` must come from wino_pack_weights.
int wino_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    if (cfg.id >= 300) return h2_launch(s, cfg, p, ksplit);
    if (cfg.id >= 200 && cfg.id < 210) return wino2_launch(s, cfg, p, ksplit);
    set_error("wino_launch: no kernel for config %d", cfg.id);
    return STX_ERR_UNSUPPORTED;
}

}  // namespace stx
