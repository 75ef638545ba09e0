// 3x3 convolution through 2-D Winograd F(2x2, 3x3) on the fp32 matrix cores -- four-wave form.
//
// Same arithmetic, packed filter bank, LDS layout and epilogues as conv_wino2.hip (see the
// algebra there); what changes is who does what.  conv_wino2 runs eight waves, two per SIMD,
// which share the SIMD's matrix pipe and hand the staging work back and forth.  Measured on
// MI355X (tools/ubench/solo_issue.hip, tools/ubench/wino2_bench.hip):
//   * v_mfma_f32_32x32x2_f32 occupies the SIMD for 64 cycles; memory instructions of the same
//     wave (buffer_load, ds_read_b128, ds_write_b128) issue in its shadow for free: 64 MFMAs +
//     32 LDS reads + 16 loads + 16 LDS writes = 4168 cycles, 98 % of the pipe;
//   * vector ALU work is NOT hidden: one v_add / v_pk_add between two MFMAs costs +13.5 cycles,
//     every further one of the same burst +4 (the fp32 MFMA runs at the vector rate -- it
//     evidently shares the vector lanes).  Bt d B is therefore issued as one burst per patch
//     instead of one packed add per MFMA;
//   * the eight-wave kernel spends 4680 cycles per 64 MFMAs of a SIMD (the younger wave of a
//     pair finishes its tail alone) and 15 400 cycles in an epilogue that exchanges the xi rows
//     of the output transform through LDS.
// Here a workgroup is FOUR waves, one per SIMD, 64 output channels x 64 tiles as before.  Wave
// (cb, tb) owns channel block cb (32 channels) and tile block tb (32 tiles) for ALL sixteen
// (xi, nu) components: 16 accumulators of one 32x32 block each = 256 registers (the unified
// 512-register file of gfx950 makes that possible).  Consequences:
//   * nothing to arbitrate on a SIMD; a wave's own staging pieces ride in its MFMA shadows;
//   * the whole output transform At M A happens in registers: no LDS exchange, no barrier
//     between the main loop and the stores;
//   * one barrier per chunk, placed after the third k-step: by then every wave has written its
//     share of the next chunk and issued its last reads of the current one, so the first
//     operands of the next chunk are fetched during the last k-step of this one and the matrix
//     pipe never waits at a chunk boundary.  The barrier waits for LDS traffic only
//     (lgkmcnt), never for the global loads in flight for the chunk after next.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace stx {

#ifndef STX_W4_SKIP
#define STX_W4_SKIP 0   // timing experiments only (wrong results when non-zero): 1 no transform, 2 no
#endif                  // barrier, 4 no patch loads, 8 no filter loads, 16 no LDS writes in the main loop

#ifdef STX_WINO4_TIMING   // cycle counters for tools/ubench/wino2_bench.hip
__device__ long long g_wino4_timing[4][4];
#define STX_T4(var) const long long var = clock64()
#define STX_T4R(var) const long long var = wall_clock64()
#else
#define STX_T4R(var) const long long var = 0
#define STX_T4(var) const long long var = 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));

namespace {

constexpr int KC = 8, BM = 64, NT = 256;
constexpr int U_FLOATS = 4 * KC * BM * 4;     // [xi][ci][m][nu]
constexpr int V_FLOATS = 4 * KC * 64 * 4;     // [xi][ci][tile][nu]
constexpr int STAGE = U_FLOATS + V_FLOATS;    // 64 KB
constexpr size_t kLdsBytes = 2 * STAGE * sizeof(float);

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic of this wave complete (reads returned, writes performed), then the workgroup
// barrier.  vmcnt is left alone: global loads for the chunk after next stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

// TXW = tiles per tile row within a 32-tile block: 32 -> 4 x 64 pixel patches, 16 -> 8 x 32,
// 8 -> 16 x 16.  Every geometry computes every output with the same arithmetic in the same
// order (and the same as conv_wino2): the choice never changes a result.
template <int EPI, int TXW>
__global__ __launch_bounds__(NT) void conv_wino4_kernel(WinoArgs a) {
    constexpr int TYW = 32 / TXW, PR = 4 * TYW, PC = 2 * TXW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    STX_T4(t_start);
    STX_T4R(r_start);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int cb = wave & 1, tb = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_mfma.hip
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int kslice = sgpr(EPI == kEpiPartial ? L % a.ksplit : 0);
    const int Lt = sgpr(EPI == kEpiPartial ? L / a.ksplit : L);
    const int ptile = sgpr(Lt / m_tiles);
    const int mtile = Lt - ptile * m_tiles;
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

    // ---- staging roles.  Thread (wave, lane) transforms, for tile `lane` of the patch, input
    // channels `wave` and `wave + 4` of the chunk.  Addressing, zero padding through the buffer
    // descriptor's range check and the edge / corner fix-ups are those of conv_wino2.hip.
    const int st_x = x0 + 2 * (lane % TXW) - 1;           // first patch column
    const bool left = st_x < 0;
    const bool corner = wave == 0 && y0 == 0 && x0 == 0;  // uniform: lane 0, patch row 1
    const bool corner_lane = lane == 0;
    // (the second channel goes into the vector offset, not the scalar one: the range check
    // compares against num_records - soffset, which must not go negative)
    unsigned xvoff[2][4];
    {
        const int ty = lane / TXW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = y0 + 2 * ty - 1 + i;
            const int off = wave * HW + yy * a.W + st_x;
            const bool in = (unsigned)yy < (unsigned)a.H && st_x < a.W;
            xvoff[0][i] = in ? (unsigned)(off < 0 ? 0 : off) * 4u : kOob;
            xvoff[1][i] = in ? (unsigned)(off + 4 * HW) * 4u : kOob;
        }
    }
    const bool edge_l = x0 == 0, edge_r = x0 + PC + 2 > a.W;      // workgroup-uniform
    const bool ok2 = st_x + 2 < a.W, ok3 = st_x + 3 < a.W;
    const unsigned w_base = (unsigned)(mtile * a.w_tile_stride) * 4u;
    constexpr unsigned w_chunk = (unsigned)U_FLOATS * 4u;
    const unsigned x_chunk = (unsigned)(KC * HW) * 4u;
    // LDS byte addresses of this thread's writes, one register per buffer (kept opaque: the
    // second buffer lies beyond the 16-bit immediate of ds_write, and an address re-derived
    // between two MFMAs is a vector instruction the matrix pipe waits for)
    unsigned u_dst[2], v_dst[2];
    u_dst[0] = (unsigned)tid * 16u;
    v_dst[0] = (unsigned)(U_FLOATS + (wave * 64 + lane) * 4) * 4u;
    u_dst[1] = u_dst[0] + (unsigned)(STAGE * 4);
    v_dst[1] = v_dst[0] + (unsigned)(STAGE * 4);
    asm volatile("" : "+v"(u_dst[0]), "+v"(u_dst[1]), "+v"(v_dst[0]), "+v"(v_dst[1]));
    constexpr unsigned v_half = (unsigned)(4 * 64 * 4) * 4u;      // channel wave + 4 in LDS
    char *const lds_bytes = reinterpret_cast<char *>(lds);

    u32x4 wreg[8];
    f32x4 xreg[2][4];
    f32x2 tq[4][2];    // Bt d, two columns at a time
    f32x4 vq[4];       // Bt d B, one transform row each

#define STX_PK(dst, a_, b_, mods) asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(dst) : "v"(a_), "v"(b_))
    auto u_load = [&](int n, unsigned ws) {
        wreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(tid + n * NT) * 16u, ws, 0);
    };
    auto x_load = [&](int pz, int i, unsigned xs) {
        xreg[pz][i] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[pz][i], xs, 0));
    };
    auto u_write = [&](int n, int buf) {
        *reinterpret_cast<u32x4 *>(lds_bytes + u_dst[buf] + n * (NT * 16)) = wreg[n];
    };
    // Border fix-ups of patch pz (workgroups on the left / right edge of the plane only).  It is
    // ONE uniform branch, issued right after an MFMA so that the instruction-fetch bubble of the
    // jump falls into that MFMA's shadow; inside, the selects are unconditional.  (Measured:
    // three separate conditionals inside the transform burst cost ~230 cycles per chunk.)
    auto fix_edges = [&](int pz) {
        f32x4 *xr = xreg[pz];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xr[i]));
        if (edge_l || edge_r) {
            asm volatile("");      // keeps this a (scalar) branch
            if (corner && pz == 0) {
                asm volatile("");  // one wave in the whole launch takes it
                const float r0 = xr[1].x, r1 = xr[1].y, r2 = xr[1].z, r3 = xr[1].w;
                xr[1].y = corner_lane ? r0 : r1;
                xr[1].z = corner_lane ? r1 : r2;
                xr[1].w = corner_lane ? r2 : r3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[i].x = left ? 0.f : xr[i].x;
                xr[i].z = ok2 ? xr[i].z : 0.f;
                xr[i].w = ok3 ? xr[i].w : 0.f;
            }
        }
    };
    // Bt d B of patch pz as one burst of vector work: sixteen packed adds, nothing else.
    auto transform = [&](int pz) {
        f32x4 *xr = xreg[pz];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xr[i]));
        // rows: t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3 on column pair h
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 d0 = h ? xr[0].zw : xr[0].xy, d1 = h ? xr[1].zw : xr[1].xy;
            const f32x2 d2 = h ? xr[2].zw : xr[2].xy, d3 = h ? xr[3].zw : xr[3].xy;
            STX_PK(tq[0][h], d0, d2, "neg_lo:[0,1] neg_hi:[0,1]");
            STX_PK(tq[1][h], d1, d2, "");
            STX_PK(tq[2][h], d2, d1, "neg_lo:[0,1] neg_hi:[0,1]");
            STX_PK(tq[3][h], d1, d3, "neg_lo:[0,1] neg_hi:[0,1]");
        }
        // columns, with P = (t[.][0], t[.][1]) and Q = (t[.][2], t[.][3]) of transform row x:
        //   (v0, v1) = (P.x - Q.x, P.y + Q.x)      (v2, v3) = (Q.x - P.y, P.y - Q.y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            f32x2 r;
            STX_PK(r, tq[x][0], tq[x][1], "op_sel_hi:[1,0] neg_lo:[0,1]");
            vq[x].xy = r;
            STX_PK(r, tq[x][0], tq[x][1], "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]");
            vq[x].zw = r;
        }
    };
    auto v_write = [&](int pz, int x, int buf) {
        *reinterpret_cast<f32x4 *>(lds_bytes + v_dst[buf] + (pz ? v_half : 0u) + x * (KC * 64 * 16)) = vq[x];
    };
    auto load_stage = [&](int chunk) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * w_chunk));
        const unsigned xs = (unsigned)sgpr((int)((unsigned)chunk * x_chunk));
#pragma unroll
        for (int n = 0; n < 8; ++n) u_load(n, ws);
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int i = 0; i < 4; ++i) x_load(pz, i, xs);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int n = 0; n < 8; ++n) u_write(n, buf);
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            fix_edges(pz);
            transform(pz);
#pragma unroll
            for (int x = 0; x < 4; ++x) v_write(pz, x, buf);
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // operand addresses (floats): transform row xi, k-step q -> channels 2q + half of the chunk
    const int a_off = (half * BM + cb * 32 + l31) * 4;
    const int b_off = U_FLOATS + (half * 64 + tb * 32 + l31) * 4;
    constexpr int XI_STRIDE = KC * 64 * 4, Q_STRIDE = 2 * 64 * 4;
    constexpr int NS = KC / 2;
    f32x4 av[2][4], bv[2][4];
    auto read_operand = [&](const float *base, int q, int which) {
        // which 0..3: A of transform rows 0..3; 4..7: B
        const int xi = which & 3, o = xi * XI_STRIDE + q * Q_STRIDE;
        if (which < 4) av[q & 1][xi] = *reinterpret_cast<const f32x4 *>(base + a_off + o);
        else bv[q & 1][xi] = *reinterpret_cast<const f32x4 *>(base + b_off + o);
    };

    // One chunk of matrix work out of LDS buffer `cur`: 4 k-steps x 16 MFMAs.  Slot p = 16 s + m
    // carries, after its MFMA: the operand reads of the next k-step (m < 8); with STORE the
    // hand-over of the next chunk into the other buffer (filter image p 8..15, patch 0 fix-up,
    // transform and writes p 24..29, patch 1 p 30..35) and the barrier before slot 48, after
    // which the next chunk's first operands are read from the other buffer; with LOAD the loads
    // of the chunk after next (filter image p 16..23, patches p 36..43).
    auto run_chunk = [&](int cur, int chunk, auto store_c, auto load_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
        const float *base = lds + cur * STAGE;
        const float *next = lds + (cur ^ 1) * STAGE;
        const int nb_ = cur ^ 1;
        unsigned ws = 0, xs = 0;
        if (LOAD) {
            ws = (unsigned)sgpr((int)(w_base + (unsigned)(chunk + 2) * w_chunk));
            xs = (unsigned)sgpr((int)((unsigned)(chunk + 2) * x_chunk));
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int p = s * 16 + m, xi = m >> 2, nu = m & 3;
                if (STORE && p == 48 && !(STX_W4_SKIP & 2)) {
                    __builtin_amdgcn_sched_barrier(0);
                    lds_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][xi][nu], bv[s & 1][xi][nu],
                                                              acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m < 8) {
                    if (s + 1 < NS) read_operand(base, s + 1, m);
                    else if (STORE) read_operand(next, 0, m);
                }
                if (STORE) {
                    if (!(STX_W4_SKIP & 16) && p >= 8 && p < 16) u_write(p - 8, nb_);
                    if (!(STX_W4_SKIP & 1) && p == 24) fix_edges(0);
                    if (!(STX_W4_SKIP & 1) && p == 25) transform(0);
                    if (!(STX_W4_SKIP & 16) && p >= 26 && p < 30) v_write(0, p - 26, nb_);
                    if (!(STX_W4_SKIP & 1) && p == 30) fix_edges(1);
                    if (!(STX_W4_SKIP & 1) && p == 31) transform(1);
                    if (!(STX_W4_SKIP & 16) && p >= 32 && p < 36) v_write(1, p - 32, nb_);
                }
                if (LOAD) {
                    if (!(STX_W4_SKIP & 8) && p >= 16 && p < 24) u_load(p - 16, ws);
                    if (!(STX_W4_SKIP & 4) && p >= 36 && p < 44) x_load((p - 36) >> 2, (p - 36) & 3, xs);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    load_stage(c_begin);
    // clear the accumulators while the first loads are in flight (the compiler would sink the 256
    // moves to just before the first MFMA, behind the barrier)
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+a"(acc[c]));
    store_stage(0);
    if (c_begin + 1 < c_end) load_stage(c_begin + 1);
    lds_barrier();
#pragma unroll
    for (int which = 0; which < 8; ++which) read_operand(lds, 0, which);
    STX_T4(t_mid);

    int cur = 0;
    int chunk = c_begin;
    // two chunks per trip: the LDS buffer index is a constant in each half, so every LDS address
    // is a register plus an immediate
    for (; chunk + 3 < c_end; chunk += 2) {
        run_chunk(0, chunk, yes{}, yes{});
        run_chunk(1, chunk + 1, yes{}, yes{});
    }
    for (; chunk + 2 < c_end; ++chunk) {
        run_chunk(cur, chunk, yes{}, yes{});
        cur ^= 1;
    }
    if (chunk + 1 < c_end) {
        run_chunk(cur, chunk, yes{}, no{});
        cur ^= 1;
        ++chunk;
    }
    // ---- epilogue, all in registers.  D register r of an accumulator is channel
    // (r & 3) + 8 * (r >> 2) + 4 * half of the block, column l31 is the tile.  At M A:
    //   nu -> two output columns   (c0 + c1 + c2,  c1 - c2 - c3)
    //   xi -> two output rows      (p0 + p1 + p2,  p1 - p2 - p3)
    // Everything the epilogue touches in memory goes through buffer descriptors whose range
    // check does the predication: a lane whose output lies outside the plane (or whose channel
    // is past M) carries an out-of-range offset, its loads return 0 and its stores are dropped.
    // No divergent branch, no 64-bit address arithmetic: per access one vector offset that
    // depends on the lane only (row / column / upper-half channels) plus one scalar offset for
    // the channel.  (The branchy form of this epilogue took 12 000 cycles, of which the stores
    // themselves were nothing: measured with the stores removed.)
    float s_scale = 0.f, c_scale = 0.f;
    if (EPI == kEpiDgradInject) {
        const float n = (float)((size_t)a.M * HW);
        if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / n + kEps));
        if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / n + kEps));
    }
    // (scalars, not a reference to a.inj.win: taking the address of a member of the argument
    // block sends the whole block to scratch)
    const int cw_ch = a.inj.win.ch, cw_cw = a.inj.win.cw,
              cw_oy = a.inj.win.oy - a.inj.win.sy,
              cw_ox = a.inj.win.ox - a.inj.win.sx;
    const float *const content = a.inj.content;
    auto content_at = [&](int c, int y, int x) __attribute__((always_inline)) {      // common.h: content_index
        int ry_ = (cw_oy + y) % cw_ch, rx_ = (cw_ox + x) % cw_cw;
        if (ry_ < 0) ry_ += cw_ch;
        if (rx_ < 0) rx_ += cw_cw;
        return content[((size_t)c * cw_ch + ry_) * cw_cw + rx_];
    };
    const bool weven = (a.W & 1) == 0;       // pairs never straddle the end of a row
    const int tix = tb * 32 + l31;
    const int yy = y0 + 2 * (tix / TXW), xx0 = x0 + 2 * (tix % TXW);
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
    const unsigned vpool = (yy < a.H && xx0 < a.W)
                               ? (unsigned)((4 * half) * ph * pw + (yy >> 1) * pw + (xx0 >> 1)) * 4u
                               : kOob;
    // channel of output register r on this wave's lower lane half, clamped to M: a scalar offset
    // must not exceed the descriptor's range (the check is offset >= num_records - soffset)
    const int M_ = a.M;
    auto chan = [&](int r) __attribute__((always_inline)) {
        const int c = m0 + cb * 32 + (r & 3) + 8 * (r >> 2);
        return c < M_ ? c : M_;
    };
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t &rs, int y, unsigned so, auto even_c) __attribute__((always_inline)) {
        float2 v;
        if (decltype(even_c)::value) {
            const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo[y][0], so, 0));
            v = make_float2(t.x, t.y);
        } else {
            v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[y][0], so, 0));
            v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[y][1], so, 0));
        }
        return v;
    };
    auto st2 = [&](int y, unsigned so, float2 v, auto even_c) __attribute__((always_inline)) {
        if (decltype(even_c)::value) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v.x, v.y}), ry, vo[y][0], so, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), ry, vo[y][0], so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.y), ry, vo[y][1], so, 0);
        }
    };
    // what one group of four output registers (four channels x 2 rows) reads
    struct Group {
        float2 mk[8], sg[8];
        float bs[4];
    };
    auto fetch = [&](int g, Group &G, auto even_c) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * g + rr, c = chan(r);
            const unsigned so = (unsigned)sgpr(c) * HW4;
            if (EPI == kEpiForward) {
                if (a.bias)
                    G.bs[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rbias, (unsigned)half * 16u, (unsigned)sgpr(c) * 4u, 0));
            } else if (EPI != kEpiPartial) {
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int n = 2 * rr + y;
                    if (a.mask) G.mk[n] = ld2(rmask, y, so, even_c);
                    if (EPI == kEpiDgradInject) {
                        if (a.inj.sgrad) G.sg[n] = ld2(rsg, y, so, even_c);
                    }
                }
            }
        }
    };
    auto finish = [&](int g, const Group &G, auto even_c) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * g + rr, c = chan(r);
            const unsigned so = (unsigned)sgpr(c) * HW4;
            float2 p[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float c0 = acc[4 * x + 0][r], c1 = acc[4 * x + 1][r];
                const float c2 = acc[4 * x + 2][r], c3 = acc[4 * x + 3][r];
                p[x] = make_float2(c0 + c1 + c2, c1 - c2 - c3);
            }
            float2 o[2] = {make_float2(p[0].x + p[1].x + p[2].x, p[0].y + p[1].y + p[2].y),
                           make_float2(p[1].x - p[2].x - p[3].x, p[1].y - p[2].y - p[3].y)};
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int n = 2 * rr + y;
                float2 v = o[y];
                if (EPI == kEpiForward) {
                    if (a.bias) v.x += G.bs[rr], v.y += G.bs[rr];
                    if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                    o[y] = v;
                } else if (EPI != kEpiPartial) {
                    if (a.mask) {
                        v.x = G.mk[n].x > 0.f ? v.x : 0.f;
                        v.y = G.mk[n].y > 0.f ? v.y : 0.f;
                    }
                    if (EPI == kEpiDgradInject) {
                        if (a.inj.content) {
                            // (one layer per tile evaluation takes this: the content map is read
                            // where it is used, not ahead of time like the mask and style terms)
                            const float2 ft = ld2(rft, y, so, even_c);
                            const int mm = c + 4 * half;
                            const bool ok = yy + y < a.H && xx0 < a.W && mm < a.M;
                            const int cy = ok ? yy + y : 0, cx = ok ? xx0 : 0, cm = ok ? mm : 0;
                            v.x += c_scale * (ft.x - content_at(cm, cy, cx));
                            v.y += c_scale * (ft.y - content_at(cm, cy, cx + 1 < a.W ? cx + 1 : cx));
                        }
                        if (a.inj.sgrad) {
                            v.x += s_scale * G.sg[n].x;
                            v.y += s_scale * G.sg[n].y;
                        }
                    }
                }
                st2(y, so, v, even_c);
            }
            // the lane's 2x2 outputs are exactly one window of the 2x2/2 pooling layer that follows
            // (ceil mode: the second row may be missing): pool.hip's arithmetic (the divisor is 4
            // or 2, so the product with its reciprocal is the same float)
            if (EPI == kEpiForward && a.pool_out) {
                const bool hy = yy + 1 < a.H;
                float pr;
                if (a.pool_mode == STX_POOL_MAX) {
                    pr = fmaxf(o[0].x, o[0].y);
                    pr = hy ? fmaxf(fmaxf(pr, o[1].x), o[1].y) : pr;
                } else {
                    pr = (o[0].x + o[0].y + (hy ? o[1].x : 0.f) + (hy ? o[1].y : 0.f)) * (hy ? 0.25f : 0.5f);
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pr), rpool, vpool,
                                                      (unsigned)sgpr(c) * (unsigned)(ph * pw * 4), 0);
            }
        }
    };

    // The first group's reads go out before the last chunk of matrix work and land during it; each
    // further group is requested before the previous one is finished.  (Requesting more ahead
    // -- two groups, or all four -- makes the compiler spill around the last chunk: measured
    // 1.1 - 1.6 x slower.)  Two instantiations, for even and odd plane widths; the accumulators
    // are only read here.
    Group ga, gb;
    if (weven) fetch(0, ga, yes{});
    else fetch(0, ga, no{});
    run_chunk(cur, chunk, no{}, no{});
    STX_T4(t_epi);
    auto tail = [&](auto even_c) __attribute__((always_inline)) {
        fetch(1, gb, even_c);
        finish(0, ga, even_c);
        fetch(2, ga, even_c);
        finish(1, gb, even_c);
        fetch(3, gb, even_c);
        finish(2, ga, even_c);
        finish(3, gb, even_c);
    };
    if (weven) tail(yes{});
    else tail(no{});
#ifdef STX_WINO4_TIMING
    if (blockIdx.x == gridDim.x / 2 + 8 && lane == 0) {
        g_wino4_timing[wave][0] = t_mid - t_start;
        g_wino4_timing[wave][1] = t_epi - t_mid;
        g_wino4_timing[wave][2] = clock64() - t_epi;
        g_wino4_timing[wave][3] = wall_clock64() - r_start;      // 100 MHz ticks
    }
#endif
}

// ------------------------------------------------------------------------------------------------
ConvConfig wino4_config(int geometry) {
    ConvConfig c;
    c.id = 210 + geometry;            // ids 210.. mark the four-wave 2-D Winograd configurations
    c.bm = BM;
    c.kc = KC;
    const int txw = geometry == 0 ? 32 : geometry == 1 ? 8 : 16;
    c.pr = 4 * (32 / txw);
    c.pc = 2 * txw;
    c.threads = NT;
    c.lds_bytes = kLdsBytes;
    return c;
}

// Patch geometry of a launch.  All three compute identical results, so the choice is free; what
// differs is how much of the last patch row / column is padding and how evenly whole rounds of
// 256 workgroups (one per CU) come out, which matters for the odd planes of a pyramid (a
// 724-pixel tile has 91 x 91 and 46 x 46 planes: 4 x 64 patches waste 41 % of a 91-pixel row).
// Cost of a candidate = the K-split model's estimate (wino2_splitk_factor: rounds x (chunks x
// 2.05 us + 6 us) + the reduce pass), minimised over the split; depends on the shape only.
static double wino4_cost(const ConvConfig &cfg, int K, int M, int H, int W) {
    const int n_chunks = ceil_div(K, KC);
    const long n = (long)ceil_div(M, BM) * ceil_div(H, cfg.pr) * ceil_div(W, cfg.pc);
    const double out_mb = 4e-6 * M * (double)H * W;
    double best_cost = 0;
    for (int f = 1; f <= 8 && (f == 1 || n_chunks / f >= 4); ++f) {
        const double rounds = (double)ceil_div((int)std::min<long>(n * f, 1 << 30), 256);
        double cost = rounds * ((double)n_chunks / f * 2.05 + 6.0);
        if (f > 1) cost += (f + 1) * out_mb / 3.0 + 5.0;
        if (f == 1 || cost < best_cost) best_cost = cost;
    }
    return best_cost;
}

double wino4_geometry_cost(int geometry, int K, int M, int H, int W) {
    return wino4_cost(wino4_config(geometry), K, M, H, W);
}

int wino4_pick_geometry(int K, int M, int H, int W) {
    int best = 0;
    double best_cost = 0;
    for (int g = 0; g < 3; ++g) {
        // long rows first: on a near tie the 4 x 64 patch wins (its loads and stores are the
        // longest row segments), then 8 x 32
        const int geo = g == 0 ? 0 : g == 1 ? 2 : 1;
        const double c = wino4_cost(wino4_config(geo), K, M, H, W);
        if (g == 0 || c < best_cost * 0.97) {
            best = geo;
            best_cost = c;
        }
    }
    return best;
}

template <int EPI, int TXW>
static int wino4_launch_epi(hipStream_t s, const WinoArgs &args, int n_wg) {
    auto kern = conv_wino4_kernel<EPI, TXW>;
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

// Same contract as wino2_launch (conv_wino2.hip), which forwards here for cfg.id >= 210.
int wino4_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
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
    a.pool_codes = nullptr;       // (this kernel never writes window codes: engine.cpp conv_writes_pool_codes)
    a.pool_mode = p.pool_mode;
    const double xb = 4.0 * p.K * (double)p.H * p.W;
    const double wb = 4.0 * (double)wino2_packed_floats(p.K, p.M);
    const double yb = 4.0 * p.M * (double)p.H * p.W;       // the epilogue addresses the output planes
    if (xb >= 2147483648.0 || wb >= 2147483648.0 || yb >= 2147483648.0) {   // through descriptors too
        set_error("wino4_launch: plane set exceeds the 2 GiB buffer-addressing limit");
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
    }
    const int epi = split ? kEpiPartial : inject ? kEpiDgradInject : p.epilogue;
    const int geo = cfg.id - 210;
#define STX_W4_CASE(E)                                                                            \
    case E:                                                                                       \
        STX_TRY(geo == 0   ? (wino4_launch_epi<E, 32>(s, a, n_wg))                                \
                : geo == 1 ? (wino4_launch_epi<E, 8>(s, a, n_wg))                                 \
                           : (wino4_launch_epi<E, 16>(s, a, n_wg)));                              \
        break;
    switch (epi) {
        STX_W4_CASE(kEpiForward)
        STX_W4_CASE(kEpiDgrad)
        STX_W4_CASE(kEpiDgradInject)
        STX_W4_CASE(kEpiPartial)
        default:
            set_error("wino4_launch: no kernel for epilogue %d", p.epilogue);
            return STX_ERR_UNSUPPORTED;
    }
#undef STX_W4_CASE
    return split ? splitk_reduce_launch(s, p, ksplit) : STX_OK;
}

}  // namespace stx
