// EXPERIMENT, NOT PART OF THE LIBRARY (not compiled by style_transfer_amd/build.py).
// Round 2: the half-tile form of conv_wino2.hip -- four waves, 64 channels x 32 tiles, 64 KB of
// LDS, <= 256 registers, so that two workgroups share a CU and one's prologue / epilogue could
// run under the other's matrix work.  It was wired in as config ids 220-222 (STX_CONV_ALGO=wino2h),
// passed the kernel parity tests on 15 of 16 shapes (one small-plane case left undebugged) and
// measured, layer by layer, within -3 .. +1 % of the eight-wave kernel:
// a wave that streams MFMAs leaves the other wave of its SIMD about one issue slot per MFMA, so a
// new workgroup's prologue took 30-45 k cycles under its neighbour's chunk loop (7-12 k with
// s_setprio 3), and raising its priority takes the cycles from the neighbour.  Kept for the record.
//
// conv_wino2.hip's kernel cut in half so that TWO workgroups share a CU.
//
// Same arithmetic (bit-identical results), packed filter bank and epilogues as conv_wino2.hip; a
// workgroup is four waves and computes 64 channels x 32 tiles.  Wave xi owns transform row xi for
// all 32 tiles and both 32-channel blocks (8 accumulators, as there).  With 64 KB of LDS and at
// most 256 registers per wave two workgroups fit a CU, one wave of each per SIMD: they are not
// coupled by a barrier, and while one is in its prologue or epilogue the other has the matrix
// pipe to itself (tools/ubench/coresident.hip priced that at 6-10 % on this instruction mix).
//
// LDS: the transformed patches of a chunk (16 KB) are double buffered; the filter image (32 KB,
// every workgroup now copies the whole of it for half the tiles) exists ONCE and is replaced in
// two halves: input channels 0-3 of the next chunk go in once k-steps 0 and 1 have been read
// (barrier before MFMA 8), channels 4-7 once k-steps 2 and 3 have (barrier before MFMA 24).

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef STX_W2H_STAGGER
#define STX_W2H_STAGGER 1
#endif
#ifndef STX_W2H_PRIO
#define STX_W2H_PRIO 3
#endif

#ifndef STX_W2_SKIP
#define STX_W2_SKIP 0   // timing experiments (tools/ubench/wino2_bench.hip): 1 no filter loads, 2 no patch
#endif                 // loads in the main loop.  Wrong results when non-zero.

namespace stx {

#ifdef STX_WINO2_TIMING   // cycle counters for tools/ubench/wino2_bench.hip
__device__ long long g_wino2h_timing[8][8];
#define STX_T(var) const long long var = clock64()
#else
#define STX_T(var) [[maybe_unused]] const long long var = 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));

namespace {

constexpr int KC = 8, BM = 64, NT = 256;
// Pixel patch of a workgroup = 32 tiles.  TXW = tiles per tile row: 32 -> 2 rows x 64 columns,
// 16 -> 4 x 32, 8 -> 8 x 16 pixels.
template <int TXW>
struct Geo {
    static constexpr int TYW = 32 / TXW;          // tile rows
    static constexpr int PR = 2 * TYW, PC = 2 * TXW;
};
constexpr int U_FLOATS = 4 * KC * BM * 4;     // [xi][ci][m][nu], one copy
constexpr int V_FLOATS = 4 * KC * 32 * 4;     // [xi][ci][tile][nu], two copies
constexpr size_t kLdsBytes = (U_FLOATS + 2 * V_FLOATS) * sizeof(float);    // 64 KB

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic of this wave complete, then the workgroup barrier.  Unlike __syncthreads() this does
// not wait for the global loads in flight for the chunk after next (vmcnt is left alone).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

template <int EPI, int TXW>
__global__ __launch_bounds__(NT, 2) void conv_wino2h_kernel(WinoArgs a) {
    constexpr int TYW = Geo<TXW>::TYW, PR = Geo<TXW>::PR, PC = Geo<TXW>::PC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    STX_T(t_start);
    // A wave that streams MFMAs leaves the other wave of its SIMD about one issue slot per MFMA
    // (tools/ubench/coissue.hip): without a priority the prologue of a new workgroup crawls
    // under its neighbour's chunk loop (measured 30-45 k cycles instead of 3.3 k).  Prologue and
    // epilogue run at high priority, the chunk loop at the default.
    __builtin_amdgcn_s_setprio(STX_W2H_PRIO);
    // The two workgroups of a CU start together and would stay in step -- prologues and epilogues
    // at the same time, nothing to overlap.  One of each first pair waits about the length of an
    // epilogue + a prologue (127 x 64 cycles), once per launch.
#if STX_W2H_STAGGER == 1
    if ((int)blockIdx.x >= 256 && (int)blockIdx.x < 512) __builtin_amdgcn_s_sleep(127);
#elif STX_W2H_STAGGER == 2
    if ((int)blockIdx.x < 512 && ((blockIdx.x >> 3) & 1)) __builtin_amdgcn_s_sleep(127);
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int xi = wave;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_mfma.hip
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int kslice = sgpr(EPI == kEpiPartial ? L % a.ksplit : 0);
    const int Lt = sgpr(EPI == kEpiPartial ? L / a.ksplit : L);
    // (channel tile slowest, so that an XCD keeps one filter slice in L2, measured no different)
    // Eight channel tiles: the 32 workgroups an XCD runs at a time take 4 channel tiles x 8
    // patches instead of 8 x 4 -- per round 8.4 MB of filters + 6.5 MB of input through the L2
    // instead of 16.8 + 3.2 (512 -> 512 channels; tools/pmc_layers.py)
    int ptile = sgpr(Lt / m_tiles);
    int mtile = Lt - ptile * m_tiles;
    if (m_tiles == 8 && ((a.tiles_x * a.tiles_y) & 7) == 0) {
        const int g = Lt >> 5, r = Lt & 31;
        mtile = (g & 1) * 4 + (r & 3);
        ptile = (g >> 1) * 8 + (r >> 2);
    }
    const int c_begin = sgpr(EPI == kEpiPartial ? kslice * a.n_chunks / a.ksplit : 0);
    const int c_end = sgpr(EPI == kEpiPartial ? (kslice + 1) * a.n_chunks / a.ksplit : a.n_chunks);
    const int y0 = sgpr((ptile / a.tiles_x) * PR);
    const int x0 = sgpr((ptile % a.tiles_x) * PC);
    const int m0 = mtile * BM;
    const int HW = a.H * a.W;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- staging roles: this thread transforms the patch of channel 2 * wave + half of the chunk
    // for tile l31.  The four columns of a patch row
    // are one 16-byte load (dword aligned; neighbouring lanes overlap by half, which the texture
    // unit coalesces).  Rows outside the plane get an offset beyond the descriptor's range and
    // read as zero, so do channels past K.  Columns outside the plane are zeroed when the values
    // are consumed, by workgroups on the left / right edge only: x = -1 is the first patch column
    // of tile column 0, x >= W can only be one of the last two patch columns of a tile that
    // still has a column inside (tiles entirely outside compute garbage nobody stores).
    // The one patch row whose x = -1 would lie before the start of the tensor (channel 0, row 0
    // of the first workgroup) is loaded from x = 0 and shifted by one instead.
    const int st_ch = 2 * wave + half;                     // its channel of the chunk
    const int st_x = x0 + 2 * (l31 % TXW) - 1;            // first patch column
    const bool left = st_x < 0;
    const bool corner = wave == 0 && y0 == 0 && x0 == 0;  // uniform: lane 0, patch row 1
    const bool corner_lane = lane == 0;
    unsigned xvoff[4];
    {
        const int ty = l31 / TXW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = y0 + 2 * ty - 1 + i;
            const int off = st_ch * HW + yy * a.W + st_x;
            xvoff[i] = (unsigned)yy < (unsigned)a.H && st_x < a.W
                           ? (unsigned)(off < 0 ? 0 : off) * 4u : kOob;
        }
    }
    const bool edge_l = x0 == 0, edge_r = x0 + PC + 2 > a.W;      // workgroup-uniform
    const bool ok2 = st_x + 2 < a.W, ok3 = st_x + 3 < a.W;
    const unsigned w_base = (unsigned)(mtile * a.w_tile_stride) * 4u;
    constexpr unsigned w_chunk = (unsigned)U_FLOATS * 4u;
    const unsigned x_chunk = (unsigned)(KC * HW) * 4u;
    // LDS byte offsets of this thread's writes within a buffer
    const unsigned u_dst = (unsigned)tid * 16u;
    const unsigned v_dst = (unsigned)((st_ch * 32 + l31) * 4) * 4u;      // within a V buffer

    u32x4 wreg[8];
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
    // (vector n of a thread is input channel wave of transform row n / 2 for even n, channel
    // wave + 4 for odd n: the even ones are the first half of the image, the odd ones the second)
    char *const u_lds = reinterpret_cast<char *>(lds);
    auto u_write = [&](int n) {
        *reinterpret_cast<u32x4 *>(u_lds + u_dst + n * (NT * 16)) = wreg[n];
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
        *reinterpret_cast<f32x4 *>(ldsb + v_dst + x * (KC * 32 * 16)) = vq[x];
    };
    auto load_stage = [&](int chunk) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * w_chunk));
        const unsigned xs = (unsigned)sgpr((int)((unsigned)chunk * x_chunk));
#pragma unroll
        for (int n = 0; n < 8; ++n) u_load(n, ws);
#pragma unroll
        for (int i = 0; i < 4; ++i) x_load(i, xs);
    };
    auto v_buf = [&](int buf) { return reinterpret_cast<char *>(lds) + (U_FLOATS + buf * V_FLOATS) * 4; };
    auto store_stage = [&](int buf) {
        char *ldsb = v_buf(buf);
#pragma unroll
        for (int n = 0; n < 8; ++n) u_write(n);
        fix_edges();
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
    const int b_off = U_FLOATS + ((xi * KC + half) * 32 + l31) * 4;          // + buffer * V_FLOATS
    constexpr int NS = KC / 2;

    // One chunk of matrix work: patches out of V buffer `cur`, with (STORE) the hand-over of the
    // next chunk -- patches into the other V buffer, the filter image in two halves into the one U
    // buffer -- and (LOAD) the loads of the chunk after that dealt out between the MFMAs (pieces
    // ride behind MFMA p = 0..31):
    //    barrier before MFMA 8   k-steps 0, 1 of this chunk have been read by every wave
    //    8-11   first half of the next filter image -> LDS      12 border fix-up, 13 Bt d B
    //    14-17  next patches -> LDS                              18-21 loads: first half, chunk + 2
    //    20-23  patch loads, chunk + 2
    //    barrier before MFMA 24  k-steps 2, 3 have been read; next patches and first half complete
    //    24-27  second half of the next filter image -> LDS     28-31 loads: second half, chunk + 2
    // and the operand reads of the next k-step behind the first three MFMAs of each k-step (after
    // the last one: the next chunk's first, from the other V buffer and the fresh first half).
    f32x4 av[2][2], bv[2];
    auto run_chunk = [&](int cur, int chunk, auto store_c, auto load_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
        const float *vcur = lds + cur * V_FLOATS, *vnext = lds + (cur ^ 1) * V_FLOATS;
        char *ldsb = v_buf(cur ^ 1);
        unsigned ws = 0, xs = 0;
        if (LOAD) {
            ws = (unsigned)sgpr((int)(w_base + (unsigned)(chunk + 2) * w_chunk));
            xs = (unsigned)sgpr((int)((unsigned)(chunk + 2) * x_chunk));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = c * 2 + i, p = s * 8 + m;
                    if (STORE && (p == 8 || p == 24)) {
                        __builtin_amdgcn_sched_barrier(0);
                        lds_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i][c], bv[s & 1][c],
                                                                      acc[i][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 3 && (s + 1 < NS || STORE)) {
                        const int oa = s + 1 < NS ? (2 * (s + 1)) * 64 * 4 : 0;
                        const int ob = s + 1 < NS ? (2 * (s + 1)) * 32 * 4 : 0;
                        const float *vsrc = s + 1 < NS ? vcur : vnext;
                        if (m == 0) av[(s + 1) & 1][0] = *reinterpret_cast<const f32x4 *>(lds + a_off + oa);
                        if (m == 1) av[(s + 1) & 1][1] = *reinterpret_cast<const f32x4 *>(lds + a_off + oa + 32 * 4);
                        if (m == 2) bv[(s + 1) & 1] = *reinterpret_cast<const f32x4 *>(vsrc + b_off + ob);
                    }
                    if (STORE) {
                        if (p >= 8 && p < 12) u_write(2 * (p - 8));
                        if (p == 12) fix_edges();
                        if (p == 13) {       // Bt d B as ONE burst of vector work (solo_issue.hip)
#pragma unroll
                            for (int q = 0; q < 8; ++q) row_op(q);
#pragma unroll
                            for (int q = 0; q < 8; ++q) col_op(q);
                        }
                        if (p >= 14 && p < 18) v_write(p - 14, ldsb);
                        if (p >= 24 && p < 28) u_write(2 * (p - 24) + 1);
                    }
                    if (LOAD) {
                        if (p >= 18 && p < 22) u_load(2 * (p - 18), ws);
                        if (p >= 20 && p < 24) x_load(p - 20, xs);
                        if (p >= 28) u_load(2 * (p - 28) + 1, ws);
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

    int cur = 0;
    int chunk = c_begin;
    [[maybe_unused]] long long t_work = 0;
    __builtin_amdgcn_s_setprio(0);
    STX_T(t_begin);
    // two chunks per trip: the LDS buffer index is a constant in each half, so every LDS address
    // of the hand-over and of the operand reads is a register plus an immediate
    // k-step 0 operands of the first chunk; from then on every chunk leaves those of its
    // successor behind (the hand-over barrier sits inside the chunk, before k-step 3)
    av[0][0] = *reinterpret_cast<const f32x4 *>(lds + a_off);
    av[0][1] = *reinterpret_cast<const f32x4 *>(lds + a_off + 32 * 4);
    bv[0] = *reinterpret_cast<const f32x4 *>(lds + b_off);
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
    const int yy = y0 + 2 * (l31 / TXW), xx0 = x0 + 2 * (l31 % TXW);
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
        a.y + (EPI == kEpiPartial ? (size_t)kslice * a.M * HW : 0), 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.mask), 0, a.mask ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.inj.sgrad), 0, a.inj.sgrad ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rft = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.inj.feat), 0, a.inj.feat ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.bias), 0, a.bias ? a.M * 4 : 0, 0x00020000);
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
    auto st2 = [&](int y, unsigned so, f32x2 v, auto even_c) __attribute__((always_inline)) {
        if (decltype(even_c)::value) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), ry, vo[y][0], so, 0);
        } else {
            const float v0 = v.x, v1 = v.y;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ry, vo[y][0], so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), ry, vo[y][1], so, 0);
        }
    };
    // The ReLU mask and the style term, for all sixteen outputs of the lane (see kEarly above).
    f32x2 mk[16], sg[16];
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
                const unsigned so = (unsigned)chan(n >> 3, (n >> 1) & 3) * HW4;
                if (a.mask) mk[n] = ld2(rmask, n & 1, so, even_c);
                if (EPI == kEpiDgradInject) {
                    if (a.inj.sgrad) sg[n] = ld2(rsg, n & 1, so, even_c);
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

    __builtin_amdgcn_s_setprio(STX_W2H_PRIO);
    // exchange: [wave][i][register pair q][lane] x (col 0 of r, col 0 of r + 1, col 1 of r, col 1
    // of r + 1), 64 KB in all
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
        const int cw_ch = a.inj.win.ch, cw_cw = a.inj.win.cw, cw_oy = a.inj.win.oy - a.inj.win.sy,
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
        auto tail = [&](auto even_c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                f32x4 p[4];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    p[x] = ex[(x * 16 + i * 8 + 2 * xi + qq) * 64 + lane];
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
                    const unsigned so = (unsigned)c * HW4;
                    f32x2 o[2];
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        const int n = (i * 4 + rr) * 2 + y;
                        f32x2 v = {h ? rows[y][0].y : rows[y][0].x, h ? rows[y][1].y : rows[y][1].x};
                        if (EPI == kEpiForward) {
                            if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                        } else if (EPI != kEpiPartial) {
                            if (a.mask) {
                                v.x = mk[n].x > 0.f ? v.x : 0.f;
                                v.y = mk[n].y > 0.f ? v.y : 0.f;
                            }
                            if (EPI == kEpiDgradInject) {
                                if (content) {
                                    // (one layer per tile evaluation takes this: the content map
                                    // is read where it is used, not ahead of time)
                                    const f32x2 ft = ld2(rft, y, so, even_c);
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
                        st2(y, so, v, even_c);
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
    }
#undef STX_PK_ADD
#undef STX_PK_SUB
#ifdef STX_WINO2_TIMING
    if (blockIdx.x == gridDim.x - 3 && lane == 0) {     // a workgroup of the last round
        g_wino2h_timing[wave][0] = t_work, g_wino2h_timing[wave][2] = 0;   // (the barrier is inside the chunk now)
        g_wino2h_timing[wave][3] = t_main_end - t_begin;
        g_wino2h_timing[wave][4] = t_begin - t_start;
        g_wino2h_timing[wave][6] = t_loads - t_start, g_wino2h_timing[wave][7] = t_stored - t_loads;
        g_wino2h_timing[wave][5] = clock64() - t_main_end;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
ConvConfig wino2h_config(int geometry) {
    ConvConfig c;
    c.id = 220 + geometry;            // ids 220.. mark the half-tile 2-D Winograd configurations
    c.bm = BM;
    c.kc = KC;
    const int txw = geometry == 0 ? 32 : geometry == 1 ? 8 : 16;
    c.pr = 2 * (32 / txw);
    c.pc = 2 * txw;
    c.threads = NT;
    c.lds_bytes = kLdsBytes;
    return c;
}

// Model cost of a launch in microseconds: rounds of 512 workgroups (two per CU), each half the
// chunk time of a whole-CU workgroup; prologue and epilogue of one run under the other's matrix
// work, so only a fraction of them counts.  Shape only.
double wino2h_geometry_cost(int geometry, int K, int M, int H, int W) {
    const ConvConfig cfg = wino2h_config(geometry);
    const int n_chunks = ceil_div(K, KC);
    const long n = (long)ceil_div(M, BM) * ceil_div(H, cfg.pr) * ceil_div(W, cfg.pc);
    return (double)ceil_div((int)std::min<long>(n, 1 << 30), 512) * (n_chunks * 2.05 + 3.0);
}

int wino2h_pick_geometry(int K, int M, int H, int W) {
    int best = 0;
    double best_cost = 0;
    for (int g = 0; g < 3; ++g) {
        const int geo = g == 0 ? 0 : g == 1 ? 2 : 1;       // long rows first
        const double c = wino2h_geometry_cost(geo, K, M, H, W);
        if (g == 0 || c < best_cost * 0.97) {
            best = geo;
            best_cost = c;
        }
    }
    return best;
}

template <int EPI, int TXW>
static int wino2h_launch_epi(hipStream_t s, const WinoArgs &args, int n_wg) {
    auto kern = conv_wino2h_kernel<EPI, TXW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(lds=%zu): %s", kLdsBytes, hipGetErrorString(e));
        return STX_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(NT), kLdsBytes, s, args);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int wino2h_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
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
    if (xb >= 2147483648.0 || wb >= 2147483648.0) {
        set_error("wino2h_launch: plane set exceeds the 2 GiB buffer-addressing limit");
        return STX_ERR_UNSUPPORTED;
    }
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
    const bool inject = p.epilogue == kEpiDgrad && (p.inject.sgrad || p.inject.content);
    int n_wg = a.m_tiles * a.tiles_x * a.tiles_y;
    const bool split = ksplit > 1 && p.splitk_ws &&
                       p.splitk_ws_floats >= (size_t)ksplit * p.M * p.H * p.W;
    if (split) {
        a.ksplit = ksplit;
        a.y = p.splitk_ws;
        n_wg *= ksplit;
    } else if (p.epilogue == kEpiForward && wino2_fuses_pool(p)) {
        a.pool_out = p.pool_out;
        a.pool_codes = p.pool_codes;
    }
    const int epi = split ? kEpiPartial : inject ? kEpiDgradInject : p.epilogue;
#define STX_W2H_CASE(E)                                                                           \
    case E:                                                                                       \
        STX_TRY(cfg.id == 221   ? (wino2h_launch_epi<E, 8>(s, a, n_wg))                           \
                : cfg.id == 222 ? (wino2h_launch_epi<E, 16>(s, a, n_wg))                          \
                                : (wino2h_launch_epi<E, 32>(s, a, n_wg)));                        \
        break;
    switch (epi) {
        STX_W2H_CASE(kEpiForward)
        STX_W2H_CASE(kEpiDgrad)
        STX_W2H_CASE(kEpiDgradInject)
        STX_W2H_CASE(kEpiPartial)
        default:
            set_error("wino2h_launch: no kernel for epilogue %d", p.epilogue);
            return STX_ERR_UNSUPPORTED;
    }
#undef STX_W2H_CASE
    return split ? splitk_reduce_launch(s, p, ksplit) : STX_OK;
}

#ifdef STX_WINO2_TIMING
void wino2h_read_timing(long long *out) {      // tools/ubench/wino2_bench.hip
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino2h_timing), sizeof(g_wino2h_timing));
}
#endif

}  // namespace stx
