// 3x3 convolution on the fp16 matrix cores at fp32 accuracy: 1-D Winograd F(2,3) along x with both
// operands split into two fp16 pieces (three products per operand pair).
//
// Same job and interface as conv_wino2_kernel (forward + bias + ReLU + fused 2x2 pooling with window
// codes, backward-to-data + ReLU mask + loss-gradient terms, split-K partials; Caffe's Convolution
// layer as the reference drives it at style_transfer.py:566,606-610), for layers with a multiple of
// 32 input channels.  For a pair of neighbouring outputs (x = 2t, 2t+1) of one row and the three taps
// g0, g1, g2 of kernel row ky:
//     d0..d3 = in[2t-1 .. 2t+2]          (row y + ky - 1)
//     V = [d0-d2, d1+d2, d2-d1, d1-d3]   U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
//     M[xi] = sum over input channels and ky of U[ky][xi] * V[ky][xi]
//     out[2t] = M0 + M1 + M2             out[2t+1] = M1 - M2 - M3
// V and U are formed in fp32 exactly as an fp32 Winograd kernel forms them.  Each is then scaled by a
// power of two and written as hi + lo, hi = fp16(s a), lo = fp16(s a - hi) (round to nearest even, the
// residual is exact): 2 x 11 significand bits.  A product is hi hi + hi lo + lo hi -- three
// v_mfma_f32_32x32x16_f16, every fp16 x fp16 product exact in the fp32 accumulator; lo lo, 2^-22 of the
// product, is dropped.  tools/f16x2_numerics.py: 2e-7 .. 6e-7 of max against float64, the same as the
// fp32 2-D Winograd kernel on the same data.  Matrix time per output and channel pair: 3 (ky) x 4/2
// (xi per pixel) x 3 products of 1/512 cycle = 18/512 against 4 x 1/32 for the fp32 2-D Winograd form:
// 0.28 of its matrix time, and -- unlike fp32 MFMAs, which run at and on the vector rate -- fp16 MFMAs
// leave the vector pipe to the transform and the split.
//
// Scales.  No fp16 value can overflow, by construction: the kernel that wrote the input blob left
// max |x| behind (kAmaxSlots words of float bits, combined with atomicMax: `x_amax`), and this kernel
// scales its input by the power of two that puts that maximum into [2^13, 2^14) -- |V| < 2^15 < 65504.
// The filter bank is scaled the same way when it is packed (its exponent sits behind the bank).  What
// is small against the maximum loses relative precision only below 2^-17 of it.  The epilogue undoes
// both scales (an exact multiplication) and leaves max |y| of its own output for the next layer.
//
// Work split.  A workgroup of eight waves computes 64 MB channels x (8 PB rows x 32 columns) = 128 PB
// x-tiles; three tilings (MB, PB) = (1, 1), (2, 1), (1, 2).  Wave (xi, h) owns transform component xi of
// MB 32-channel blocks for all 4 PB pixel blocks: 4 MB PB accumulators of one 32 x 32 MFMA block.  Its A
// operand (the U pieces of its component and channel blocks) is nobody else's, so it comes straight
// from global memory into registers, as ready fragments (1 KB per piece and step) -- one chunk ahead
// at (1, 1), kernel row by kernel row through two slots at the two wide tilings (128 accumulator
// registers).  The B operand (V pieces, shared by all channel blocks) goes through LDS:
// [xi][piece][row][tile][16 channels], 40 KB (PB = 2: 72), double buffered, one barrier per chunk of 16
// channels; a staging thread loads the four inputs of one tile for 4 channels (four 16-byte loads),
// transforms, splits and writes eight 8-byte pieces.  Pixel block (m, s) holds rows 4m + s and
// 4m + 2 + s of the patch, so that a lane's outputs in blocks (m, 0) and (m, 1) are the two rows of one
// 2x2 pooling window.
//
// Epilogue.  The four components of an output pair live in four waves: the accumulators of one
// 32-channel block and one 8-row patch per wave go through LDS (128 KB per pass, MB PB passes); wave
// (hh, m, rq pair) then finishes eight channels of four rows: a 2 x 2 window per lane and channel,
// exactly the shape conv_wino2's epilogue works on -- bias / ReLU / pooling window and its code /
// the window's ReLU sign nibble for the consumer's backward pass (forward); mask from the fp32 blob
// or from nibbles, loss terms (backward).  What a pass reads from memory is requested before the
// exchange: the epilogues of a round of workgroups run at the same moment, a bandwidth-bound burst.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "f16x2.h"

#ifndef STX_H2_SKIP
#define STX_H2_SKIP 0   // timing experiments (tools/ubench/h2conv_bench.hip): 1 no staging, 2 no filter
#endif                 // loads, 4 no patch loads in the main loop, 8 no stores, 16 no mask loads in the
                       // epilogue, 32 / 64 the staging's vector work / LDS writes alone.  Wrong results when non-zero.

namespace stx {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));

constexpr int KC = 16;                      // input channels per chunk = one MFMA k-step
constexpr int NT = 512;
constexpr int PC = 32, TX = PC / 2;         // pixel patch columns; x-tiles per patch row
constexpr int EX_BYTES = 4 * 2 * 4 * 4 * 64 * 16;   // epilogue exchange: 128 KB
// PB: pixel patches of 8 rows stacked in one workgroup (1 or 2): the staged patch, the accumulators
// and the matrix work per filter fragment double, the prologue and the filter traffic do not
template <int PB>
struct Geo {
    static constexpr int PR = 8 * PB;               // patch rows
    static constexpr int XR = PR + 2;               // input rows of the patch
    static constexpr int RT = XR * TX;              // (row, tile) positions of V: 160 / 288
    static constexpr int V_PIECE = RT * 32;         // bytes of one [rt][16 ch] fp16 array
    static constexpr int V_BYTES = 4 * 2 * V_PIECE; // [xi][piece]: 40 / 72 KB
    static constexpr int FULL = RT * 4 / 512;       // staging units every thread has: 1 / 2
    static constexpr int NU = FULL + 1;             // ... and one more for the threads of waves 0 and 1
    static constexpr size_t kLds = (EX_BYTES > 2 * V_BYTES ? EX_BYTES : 2 * V_BYTES) + 64;   // + the waves' maxima
};
constexpr int FRAG = 1024;                  // one operand fragment: 64 lanes x 8 fp16
constexpr int U_KY = 4 * 2 * FRAG;          // [xi][piece] of one (channel block, chunk, ky): 8 KB
constexpr int U_BLK = 3 * U_KY;             // one (channel block, chunk): 24 KB
constexpr int kHeaderFloats = 64;           // behind the bank: [0] max |U| (float bits), [1] the scale's exponent

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a loop the compiler cannot decline to unroll (the bodies index register arrays)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (h2_scale_exp, pow2f: f16x2.h)

}  // namespace

#ifdef STX_H2_TIMING   // cycle counters for tools/ubench/h2conv_bench.hip
__device__ long long g_h2_timing[8][8];
__device__ long long g_h2_epi[8][8];     // epilogue, pass 0: loads issued, first barrier, exchange written + barrier, done
#define STX_H2_STAMP(i) if (pass == 0) t_epi[i] = clock64()
#else
#define STX_H2_STAMP(i)
#endif

// PIN: the input planes arrive pooled -- the gradient of a 2x2/2 pooling layer's output plus that layer's
// window codes (ConvProblem::pin_codes); the staging rebuilds the four inputs of an x-tile from three
// pooled gradients and three code bytes, the arithmetic of pool_bwd_codes_kernel (pool.hip) to the bit.
// (PIN = 1 + the pooling mode: STX_POOL_MAX / STX_POOL_AVE)
template <int EPI, int MB, int PB, int PIN>
__global__ __launch_bounds__(NT) void conv_h2_kernel(WinoArgs a) {
    using G = Geo<PB>;
    [[maybe_unused]] constexpr int PR = G::PR, RT = G::RT, V_PIECE = G::V_PIECE, V_BYTES = G::V_BYTES, NU = G::NU, FULL = G::FULL;
    constexpr size_t kLdsBytes = G::kLds;
    constexpr int NJ = 4 * PB;                  // pixel blocks of 32 positions
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
#ifdef STX_H2_TIMING
    const long long t_start = clock64(), w_start = wall_clock64();
#endif
    constexpr int BM = 64 * MB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int xi = wave & 3, wh = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_wino2.hip.  (Round 6: every kernel argument the index arithmetic in front
    // of the first loads needs is fetched in ONE batch -- the empty asm pins them here.  Read where they are used
    // they came in six dependent batches of scalar loads, each waited for, and the input's maximum, which no
    // load needs, in four more in front of the first patch load: a workgroup's prologue 9.4 k -> 5.9 k cycles,
    // the kernels 2 .. 5 % -- profiles/r06_h2_prologue_stamps.txt, r06_h2_early_args_ab.txt.)
    const int m_tiles = a.m_tiles, a_tiles_x = a.tiles_x, a_tiles_y = a.tiles_y, a_H = a.H, a_W = a.W;
    const int a_n_chunks = a.n_chunks, a_ksplit = a.ksplit, a_item_base = a.item_base, a_x_bytes = a.x_bytes,
              a_w_bytes = a.w_bytes;
    const float *const a_x = a.x, *const a_w = a.w;
    asm volatile("" ::"s"(m_tiles), "s"(a_tiles_x), "s"(a_tiles_y), "s"(a_H), "s"(a_W), "s"(a_n_chunks), "s"(a_ksplit),
                 "s"(a_item_base), "s"(a_x_bytes), "s"(a_w_bytes), "s"(a_x), "s"(a_w));
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot_x = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot_x;
    const int kslice = sgpr(EPI == kEpiPartial ? L % a_ksplit : 0);
    const int Lt = sgpr(EPI == kEpiPartial ? a_item_base + L / a_ksplit : L);
    int ptile, mtile;
    wino2_item_tiles(Lt, m_tiles, a_tiles_x * a_tiles_y, ptile, mtile);
    ptile = sgpr(ptile);
    mtile = sgpr(mtile);
    // (K slices are whole pairs of chunks: the chunk loop below runs two per trip)
    const int c_begin = sgpr(EPI == kEpiPartial ? 2 * (kslice * (a_n_chunks >> 1) / a_ksplit) : 0);
    const int c_end = sgpr(EPI == kEpiPartial ? 2 * ((kslice + 1) * (a_n_chunks >> 1) / a_ksplit) : a_n_chunks);
    const int y0 = sgpr((ptile / a_tiles_x) * PR);
    const int x0 = sgpr((ptile % a_tiles_x) * PC);
    const int m0 = mtile * BM;
    const int HW = a_H * a_W;
    const unsigned HW4 = (unsigned)HW * 4u;
    // PIN: the pooled plane the input is given on (ceil mode)
    const int pih = (a_H + 1) >> 1, piw = (a_W + 1) >> 1;
    const unsigned XP4 = PIN ? (unsigned)(pih * piw) * 4u : HW4;      // bytes of one input plane

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a_x), 0, a_x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a_w), 0, a_w_bytes, 0x00020000);
    // (three code bytes are fetched as one dword at any byte offset: the range ends 3 bytes behind the array)
    const __amdgpu_buffer_rsrc_t rcx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.pin_codes), 0, PIN ? (a_x_bytes >> 2) + 3 : 0, 0x00020000);

    // ---- scales: the input's from the maximum its producer left, the bank's from its header -- read behind the
    // first loads' requests (read_scales, in the prologue): only the staging needs them
    float sv = 0.f, out_scale = 0.f, out_scale2 = 0.f;
    auto read_scales = [&]() __attribute__((always_inline)) {
        // (read-only in this launch and wave-uniform: the constant address space makes them scalar loads wherever
        // the block stands -- behind the first vector loads the compiler turned them into sixteen 16-byte VMEM loads)
        typedef const __attribute__((address_space(4))) unsigned *kconst_u32;
        const kconst_u32 slots = (kconst_u32)(uintptr_t)a.x_amax, header = (kconst_u32)(uintptr_t)a_w;
        unsigned amax_bits = 0;
#pragma unroll
        for (int i = 0; i < kAmaxSlots; ++i) amax_bits = max(amax_bits, slots[i]);
        const int es = sgpr(h2_scale_exp(amax_bits));
        const int ew = sgpr((int)header[(a_w_bytes >> 2) + 1]);
        sv = pow2f(es);
        // (both scales undone exactly: 2^-(es + ew) in two factors when it leaves the normal range -- a blob
        // whose maximum is below 2^-99 with a bank exponent near 14; the second factor is 1 otherwise)
        const int eo = -es - ew, eo1 = eo < -126 ? -126 : eo > 127 ? 127 : eo;
        out_scale = pow2f(eo1), out_scale2 = pow2f(eo - eo1);
    };

    // ---- staging role.  The V array of a chunk has RT positions (a row of the patch, an x-tile) x 16
    // channels.  Positions 0 .. 128 FULL - 1: a unit = one position x four channels (quad), every thread
    // has unit tid (and 512 + tid).  The last 32 positions: one position x ONE channel per thread (32 x 16 =
    // all 512 threads; index FULL below) -- as four-channel units they were 128, a second / third unit for
    // the threads of waves 0 and 1 alone: sixteen registers in every wave, and the vector work of two waves
    // half as much again as the others'.
    const bool edge = x0 == 0 || x0 + PC + 2 > a.W;       // workgroup-uniform
    unsigned xvoff[NU], v_dst[NU];
    bool left[NU], ok2[NU], ok3[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const int u = tid + n * NT;
        const int rt = n < FULL ? u >> 2 : 128 * FULL + (tid >> 4);
        const int ch0 = n < FULL ? (u & 3) * 4 : tid & 15;       // first (only) channel of the chunk
        const int st_r = rt / TX, st_t = rt % TX;
        const int st_y = y0 - 1 + st_r, st_x = x0 + 2 * st_t - 1;
        left[n] = st_x < 0;                                // x = -1: loaded from x = 0 and shifted
        ok2[n] = st_x + 2 < a.W, ok3[n] = st_x + 3 < a.W;
        xvoff[n] = kOob;
        if (PIN) {
            // columns st_x .. st_x + 3 lie in windows w0 (its right column), w0 + 1 (both), w0 + 2 (its left one)
            const int w0 = (x0 >> 1) + st_t - 1;
            // (the two free bits of the offset: the row's place in its window, and whether the window has two rows)
            if ((unsigned)st_y < (unsigned)a.H && st_x + 1 < a.W)
                xvoff[n] = (unsigned)(ch0 * pih * piw + (st_y >> 1) * piw + (left[n] ? 0 : w0)) * 4u +
                           (unsigned)(st_y & 1) + ((st_y | 1) < a.H ? 2u : 0u);
        } else if ((unsigned)st_y < (unsigned)a.H && st_x + 1 < a.W)
            xvoff[n] = (unsigned)(ch0 * HW + st_y * a.W + (left[n] ? 0 : st_x)) * 4u;
        // the two 16-byte halves of a position swap places on tiles 8..15: conflict-free ds_read_b128
        v_dst[n] = (unsigned)(rt * 32 + ((ch0 * 2) ^ ((st_t & 8) << 1)));
    }

    f32x4 xr[FULL][4], xe;
    // what a thread holds of one channel: the four inputs d0 .. d3 of its x-tile; PIN: three pooled
    // gradients and (in .w: one dword, fetched at any byte offset) their three code bytes
    auto x_fetch = [&](int n, unsigned soff) __attribute__((always_inline)) {
        if constexpr (PIN) {
            // (what derives from the offset is derived again at every use: hoisted out of the chunk loop it
            // would hold a dozen registers the 128-accumulator tilings do not have)
            unsigned xo = xvoff[n];
            asm volatile("" : "+v"(xo));
            const u32x3 g = __builtin_amdgcn_raw_buffer_load_b96(rx, xo & ~3u, soff, 0);
            // (out of range stays out of range: < 2^29 bytes of codes)
            const unsigned cw = __builtin_amdgcn_raw_buffer_load_b32(rcx, xo >> 2, soff >> 2, 0);
            return __builtin_bit_cast(f32x4, u32x4{g.x, g.y, g.z, cw});
        } else {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[n], soff, 0));
        }
    };
    auto x_load = [&](int n, int chunk) __attribute__((always_inline)) {
        const unsigned xs = (unsigned)sgpr(chunk * KC) * XP4;
        if (n == FULL) {
            xe = x_fetch(FULL, xs);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[n][i] = x_fetch(n, xs + (unsigned)i * XP4);
        }
    };
    // the border fix-up of one channel (workgroups on the left / right border only)
    auto border = [&](f32x4 d, int n) __attribute__((always_inline)) {
        f32x4 e;
        e.x = left[n] ? 0.f : d.x;
        e.y = left[n] ? d.x : d.y;
        e.z = left[n] ? d.y : d.z;
        e.w = left[n] ? d.z : d.w;
        e.z = ok2[n] ? e.z : 0.f;
        e.w = ok3[n] ? e.w : 0.f;
        return e;
    };
    // PIN: the four inputs of a channel from what x_fetch brought -- pool_bwd_codes_kernel's routing (MAX:
    // the window's gradient goes to its first maximum, if that was positive where the blob is rectified;
    // AVE: a quarter / half / all of it to every element that was).
    struct Unpool {
        unsigned cmask, t_l, t_r, sh;
        float q0;
        bool ok4;
    };
    auto unpool_setup = [&](int n) __attribute__((always_inline)) {
        Unpool r;
        const unsigned keep = a.pin_mask ? 4u : 0u;
        r.cmask = keep | 3u;
        unsigned xo = xvoff[n], vd = v_dst[n];
        asm volatile("" : "+v"(xo), "+v"(vd));             // (see x_fetch)
        // MAX codes that select this row's left / right element; AVE: the row's first bit
        r.sh = (xo & 1u) << 1;
        r.t_l = keep | r.sh, r.t_r = r.t_l | 1u;
        r.q0 = (xo & 2u) ? 0.25f : 0.5f;                               // full windows: g / 4 (exact either way)
        r.ok4 = x0 + 2 * (int)((vd >> 5) & 15u) + 3 < a.W;             // (st_x + 4 < W; used on the border only)
        return r;
    };
    auto unpool_channel = [&](f32x4 d, int n, const Unpool &r, bool fix) __attribute__((always_inline)) {
        constexpr bool is_max = PIN == 1 + STX_POOL_MAX;
        float g0 = d.x, g1 = d.y, g2 = d.z;
        // (through a scalar: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
        const float dw = d.w;
        unsigned cw = __builtin_bit_cast(unsigned, dw);
        if (fix) {                                         // x = -1: windows 0, 1, 2 were fetched
            g2 = left[n] ? g1 : g2, g1 = left[n] ? g0 : g1;
            cw = left[n] ? cw << 8 : cw;
        }
        float p0, p1, p2, p3;
        if (is_max) {
            const unsigned b0 = cw & r.cmask, b1 = (cw >> 8) & r.cmask, b2 = (cw >> 16) & r.cmask;
            p0 = b0 == r.t_r ? g0 : 0.f;
            p1 = b1 == r.t_l ? g1 : 0.f;
            p2 = b1 == r.t_r ? g1 : 0.f;
            p3 = b2 == r.t_l ? g2 : 0.f;
        } else {
            // (a window cut by the right border has one column: twice the share)
            const float q1 = fix && !ok2[n] ? r.q0 + r.q0 : r.q0, q2 = fix && !r.ok4 ? r.q0 + r.q0 : r.q0;
            const unsigned bits = a.pin_mask ? cw >> r.sh : 0xffffffffu;
            p0 = (bits & 2u) ? g0 * r.q0 : 0.f;
            p1 = (bits & 0x100u) ? g1 * q1 : 0.f;
            p2 = (bits & 0x200u) ? g1 * q1 : 0.f;
            p3 = (bits & 0x10000u) ? g2 * q2 : 0.f;
        }
        if (fix) {
            p0 = left[n] ? 0.f : p0;
            p2 = ok2[n] ? p2 : 0.f;
            p3 = ok3[n] ? p3 : 0.f;
        }
        return f32x4{p0, p1, p2, p3};
    };
    auto unpool = [&](int n, bool fix) __attribute__((always_inline)) {
        const Unpool r = unpool_setup(n);
        if (n == FULL) {
            xe = unpool_channel(xe, FULL, r, fix);
            return;
        }
        // (channel by channel: let the scheduler interleave the four and their temporaries all live at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xr[n][i] = unpool_channel(xr[n][i], n, r, fix);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto component = [](f32x4 d, int c) __attribute__((always_inline)) {
        return c == 0 ? d.x - d.z : c == 1 ? d.y + d.z : c == 2 ? d.z - d.y : d.y - d.w;
    };
    // One piece of the staging work of unit n: component c of its four channels -- four additions, the
    // split (eight v_fma_mix: hi = fp16(s v), lo = fp16(s v - hi), each ONE instruction with mixed
    // fp32 / fp16 operands; the residual is exact) and the two 8-byte writes.  One asm block: around
    // single-instruction asm statements the compiler pads with s_nop, every one an issue slot beside
    // the MFMAs; inside, a partially written register (op_sel destination) is read two instructions
    // after its last write at the earliest.
    auto piece = [&](int n, int c, char *vbuf, bool fix) __attribute__((always_inline)) {
        if (!PIN && c == 0 && fix) {
            asm volatile("");      // (a scalar branch)
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[n][i] = border(xr[n][i], n);
        }
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = component(xr[n][i], c);
        unsigned h0, h1, l0, l1;
        if (STX_H2_SKIP & 64) {       // (timing experiment: the LDS writes without the vector work)
            h0 = __builtin_bit_cast(unsigned, v[0]), h1 = __builtin_bit_cast(unsigned, v[1]);
            l0 = __builtin_bit_cast(unsigned, v[2]), l1 = __builtin_bit_cast(unsigned, v[3]);
        } else {
            asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
                "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
                "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
                "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
                "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
                : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(sv));
        }
        if (STX_H2_SKIP & 32) {       // (timing experiment: the vector work without the LDS writes)
            asm volatile("" :: "v"(h0), "v"(h1), "v"(l0), "v"(l1));
            return;
        }
        *reinterpret_cast<u32x2v *>(vbuf + v_dst[n] + (c * 2 + 0) * V_PIECE) = u32x2v{h0, h1};
        *reinterpret_cast<u32x2v *>(vbuf + v_dst[n] + (c * 2 + 1) * V_PIECE) = u32x2v{l0, l1};
    };
    // ... and of the single-channel item: components c and c + 1 (c = 0, 2) -- two additions, four
    // v_fma_mixlo and four 2-byte writes.
    auto piece_one = [&](int c, char *vbuf, bool fix) __attribute__((always_inline)) {
        if (!PIN && c == 0 && fix) {
            asm volatile("");
            xe = border(xe, FULL);
        }
        const float va = component(xe, c), vb = component(xe, c + 1);
        unsigned ha, hb, la, lb;
        asm("v_fma_mixlo_f16 %0, %4, %6, 0\n\t"
            "v_fma_mixlo_f16 %1, %5, %6, 0\n\t"
            "v_fma_mixlo_f16 %2, %4, %6, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %5, %6, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
            : "=&v"(ha), "=&v"(hb), "=&v"(la), "=&v"(lb)
            : "v"(va), "v"(vb), "v"(sv));
        if (STX_H2_SKIP & 32) {
            asm volatile("" :: "v"(ha), "v"(hb), "v"(la), "v"(lb));
            return;
        }
        char *dst = vbuf + v_dst[FULL];
        *reinterpret_cast<unsigned short *>(dst + (c * 2 + 0) * V_PIECE) = (unsigned short)ha;
        *reinterpret_cast<unsigned short *>(dst + (c * 2 + 1) * V_PIECE) = (unsigned short)la;
        *reinterpret_cast<unsigned short *>(dst + (c * 2 + 2) * V_PIECE) = (unsigned short)hb;
        *reinterpret_cast<unsigned short *>(dst + (c * 2 + 3) * V_PIECE) = (unsigned short)lb;
    };
    auto stage_all = [&](int n, int buf) __attribute__((always_inline)) {
        char *vbuf = ldsb + buf * V_BYTES;
        if (PIN) unpool(n, edge);
        if (n == FULL) {
            piece_one(0, vbuf, edge);
            piece_one(2, vbuf, edge);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) piece(n, c, vbuf, edge);
        }
    };

    // ---- A operand: fragments of (xi, channel block), [ky][block][piece], straight from the packed bank
    const unsigned a_voff = (unsigned)(xi * 2 * FRAG + lane * 16);
    const unsigned w_blk0 = (unsigned)(mtile * 2 * MB + wh * MB);
    // MB = 1: the three kernel rows of a chunk, requested one chunk ahead; MB = 2 (half the registers
    // left): two slots, kernel row by kernel row, each requested while the one before it runs
    constexpr bool LEAN = MB * PB == 2;
    f16x8 af[LEAN ? 2 : 3][MB][2];
    auto a_load = [&](int aslot, int ky, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const unsigned ws = (unsigned)sgpr(
                (int)(((w_blk0 + b) * (unsigned)a.n_chunks + (unsigned)chunk) * U_BLK + (unsigned)ky * U_KY));
#pragma unroll
            for (int q = 0; q < 2; ++q)
                af[aslot][b][q] = __builtin_bit_cast(
                    f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, a_voff + q * FRAG, ws, 0));
        }
    };

    f32x16 acc[MB][NJ];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;

    // B fragments of block blk = NJ ky + j of a chunk, j = (m, s): rows 4m + s + ky and two below
    const unsigned b_base = (unsigned)(xi * 2 * V_PIECE + ((l31 >> 4) * 2 * TX + (l31 & 15)) * 32 +
                                       ((half * 16) ^ ((l31 & 8) << 1)));
    f16x8 bq[2][2];
    auto b_read = [&](int slot, int buf, int blk) __attribute__((always_inline)) {
        const int j = blk % NJ, ky = blk / NJ;
        const char *vb = ldsb + buf * V_BYTES + b_base + (4 * (j >> 1) + (j & 1) + ky) * TX * 32;
#pragma unroll
        for (int q = 0; q < 2; ++q) bq[slot][q] = *reinterpret_cast<const f16x8 *>(vb + q * V_PIECE);
    };

    // ---- prologue
#ifdef STX_H2_TIMING
    const long long t_p0 = clock64();          // index setup, descriptors, scales done
#endif
    static_for<0, NU>([&](auto n_c) __attribute__((always_inline)) { x_load(decltype(n_c)::value, c_begin); });
    if (LEAN) {
        a_load(0, 0, c_begin);
    } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a_load(ky, ky, c_begin);
    }
    __builtin_amdgcn_sched_barrier(0);
    read_scales();
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[b][j]));
    // ---- the epilogue's addresses and descriptors, set up here, under the first loads' latency (a thousand
    // cycles of scalar and vector work where they stood, between the chunk loop and the first store).
    // Epilogue: components through LDS, [xi][wh][j][rq][lane] x (registers 4 rq .. 4 rq + 3), one
    // 32-channel block per wave and pass.  Then wave (hh, m, rqp) finishes pixel blocks (m, 0) and (m, 1)
    // of channel block hh for register quads 2 rqp, 2 rqp + 1: per lane and channel (D register r of a
    // block is channel (r & 3) + 8 (r >> 2) + 4 half) the 2 x 2 window at rows y, y + 1, columns x, x + 1.
    float s_scale = 0.f, c_scale = 0.f;
    if (EPI == kEpiDgradInject) {
        const float n = (float)((size_t)a.M * HW);
        if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / n + kEps));
        if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / n + kEps));
    }
    const int hh = wave & 1, mrow = (wave >> 1) & 1, rqp = wave >> 2;
    const bool weven = (a.W & 1) == 0;
    const int xx0 = x0 + 2 * (l31 & 15);
    const unsigned plane_bytes = (unsigned)a.M * HW4;
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
    const int ph = (a.H + 1) >> 1, pw = (a.W + 1) >> 1;      // (ceil mode: an odd plane's last window has one column)
    const __amdgpu_buffer_rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc(
        a.pool_out, 0, a.pool_out ? a.M * ph * pw * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcodes = __builtin_amdgcn_make_buffer_rsrc(
        a.pool_codes, 0, a.pool_codes ? a.M * ph * pw : 0, 0x00020000);
    // ReLU sign nibbles of the output blob (ConvProblem::mask_codes): the lane's 2 x 2 outputs are one window
    const int cph = (a.H + 1) >> 1, cpw = (a.W + 1) >> 1;
    const __amdgpu_buffer_rsrc_t rmc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.mask_codes), 0, a.mask_codes ? a.M * cph * cpw : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t roc = __builtin_amdgcn_make_buffer_rsrc(
        a.out_codes, 0, a.out_codes ? a.M * cph * cpw : 0, 0x00020000);
    const int M_ = a.M;
    // The lane's 2 x 2 outputs of a pass: rows yy, yy + 1 (yy = y0 + 8 pbi + 4 m + 2 (l31 >> 4)), columns
    // xx0, xx0 + 1.  Byte offsets of the four outputs in a channel plane (out of range where the
    // plane ends), of the window in the pooled plane / the nibble plane, the content map's rows.
    struct LaneRows {
        int yy;
        unsigned vo[2][2], vpool, vmc;
    };
    const float *const content = a.inj.content;
    const int cw_ch = a.inj.win.ch, cw_cw = a.inj.win.cw, cw_oy = a.inj.win.oy - a.inj.win.sy,
              cw_ox = a.inj.win.ox - a.inj.win.sx;
    auto lane_rows = [&](int pbi) __attribute__((always_inline)) {
        LaneRows r;
        r.yy = y0 + 8 * pbi + 4 * mrow + 2 * (l31 >> 4);
        const unsigned lane_base = (unsigned)((4 * half) * HW + r.yy * a.W + xx0) * 4u;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                r.vo[y][e] = (r.yy + y < a.H && xx0 + e < a.W) ? lane_base + (unsigned)(y * a.W + e) * 4u : kOob;
        const bool in = r.yy < a.H && xx0 < a.W;
        r.vpool = in ? (unsigned)((4 * half) * ph * pw + (r.yy >> 1) * pw + (xx0 >> 1)) * 4u : kOob;
        r.vmc = in ? (unsigned)((4 * half) * cph * cpw + (r.yy >> 1) * cpw + (xx0 >> 1)) : kOob;
        return r;
    };
    // (hoisted: a thousand cycles under the first loads' latency -- except where the registers are not there)
    constexpr bool HOIST = !(PIN && PB == 2);
    LaneRows rows0{};
    if (HOIST) rows0 = lane_rows(0);
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t &rs, const LaneRows &lr, int y, unsigned so, auto even_c)
                   __attribute__((always_inline)) {
        if (decltype(even_c)::value)
            return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, lr.vo[y][0], so, 0));
        f32x2 v;
        v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lr.vo[y][0], so, 0));
        v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lr.vo[y][1], so, 0));
        return v;
    };
    auto st2 = [&](const __amdgpu_buffer_rsrc_t &rs, const LaneRows &lr, int y, unsigned so, f32x2 v, auto even_c)
                   __attribute__((always_inline)) {
        if (STX_H2_SKIP & 8) {        // (timing experiment: the epilogue without its stores)
            if (v.x == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(0u, rs, lr.vo[y][0], so, 0);
            return;
        }
        if (decltype(even_c)::value) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), rs, lr.vo[y][0], so, 0);
        } else {
            const float v0 = v.x, v1 = v.y;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs, lr.vo[y][0], so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs, lr.vo[y][1], so, 0);
        }
    };
    int ccol[2] = {0, 0};
    auto content_columns = [&](const LaneRows &lr) __attribute__((always_inline)) {
        if (EPI == kEpiDgradInject && content) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool ok = lr.yy < a.H && xx0 < a.W;
                const int x = ok ? (xx0 + e < a.W ? xx0 + e : xx0) : (e && 1 < a.W ? 1 : 0);
                int r = (cw_ox + x) % cw_cw;
                ccol[e] = r < 0 ? r + cw_cw : r;
            }
        }
    };
    if (HOIST) content_columns(rows0);
#ifdef STX_H2_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const long long t_p1 = clock64();          // first loads requested, epilogue addresses set up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_p2 = clock64();          // ... and landed
#endif
    static_for<0, NU>([&](auto n_c) __attribute__((always_inline)) {
        constexpr int n = decltype(n_c)::value;
        stage_all(n, 0);
        if (c_begin + 1 < c_end) x_load(n, c_begin + 1);
    });
#ifdef STX_H2_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const long long t_p3 = clock64();          // first chunk staged
#endif
    lds_barrier();
    b_read(0, 0, 0);

    // ---- main loop.  Twelve blocks of 3 MB MFMAs per chunk; the B fragments of a block are read
    // while the block before it runs; the hand-over barrier sits before the last block (every
    // wave has issued its last reads of this chunk and written its share of the next by then).
    // The staging pieces are dealt out behind the MFMAs; sched_barrier pins the order.
    constexpr int NBLK = 3 * NJ;                                     // blocks of a chunk: 12 / 24
    constexpr int PER_BLK = 3 * MB, SLOTS = (NBLK - 1) * PER_BLK;    // slots in front of the barrier
    // Staging pieces k = 0 .. NP - 1 sit behind slots 1, 1 + STEP, ...: piece k / NU of unit k % NU, the
    // units' pieces interleaved.  (All of unit 0 first, unit 1 behind it, so that each unit's registers can
    // be asked for again three quarters of a chunk ahead: 5350 -> 5720 cycles per chunk -- the vector work
    // wants to be spread evenly.)  A four-channel unit has four pieces, one component each (PIN: five, the
    // un-pooling first); the single-channel item two, two components each (PIN: three).
    constexpr int NQ = PIN ? 5 : 4, NQ1 = PIN ? 3 : 2;
    constexpr int NP = NQ * NU;
    constexpr int STEP = (SLOTS - 2) / NP;
#ifdef STX_H2_NO_BRANCH       // (timing experiment: no border fix-up -- wrong results)
    const bool edge_fix = false;
#else
    const bool edge_fix = edge;
#endif
    auto run_chunk = [&](auto buf_c, int chunk, auto more_c) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int BUF = decltype(buf_c)::value, buf = BUF;
        char *vnext = ldsb + (buf ^ 1) * V_BYTES;
        static_for<0, NBLK>([&](auto blk_c) __attribute__((always_inline)) {
            constexpr int blk = decltype(blk_c)::value;
            constexpr int ky = blk / NJ, j = blk % NJ, slot = blk & 1;
            if (blk == NBLK - 1) {
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (blk < NBLK - 1) b_read(slot ^ 1, buf, blk + 1);
            else if (MORE) b_read(0, buf ^ 1, 0);
            static_for<0, PER_BLK>([&](auto m_c) __attribute__((always_inline)) {
                constexpr int m = decltype(m_c)::value;
                constexpr int b = m % MB, t = m / MB;      // t: 0 lo hi, 1 hi lo, 2 hi hi (smallest first)
                __builtin_amdgcn_sched_barrier(0);
                constexpr int as = LEAN ? (BUF + ky) & 1 : ky;
                acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[as][b][t == 0 ? 1 : 0],
                                                                   bq[slot][t == 1 ? 1 : 0], acc[b][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int s = blk * PER_BLK + m;
                if (LEAN && j == 0 && m == 0 && !(STX_H2_SKIP & 2)) {      // the next kernel row's fragments
                    if (ky < 2) a_load(as ^ 1, ky + 1, chunk);
                    else if (MORE) a_load(as ^ 1, 0, chunk + 1);
                }
                if (MORE && !(STX_H2_SKIP & 1) && s >= 1 && (s - 1) % STEP == 0) {
                    constexpr int k = (s - 1) / STEP;             // 0 .. NP - 1
                    constexpr int n = k % NU, q = k / NU;
                    if (k < NP && PIN && q == 0) unpool(n, edge_fix);
                    else if (k < NP && n < FULL) piece(n, q - (PIN ? 1 : 0), vnext, edge_fix);
                    else if (k < NP && q < NQ1) piece_one(2 * (q - (PIN ? 1 : 0)), vnext, edge_fix);
                }
                if (MORE && !(STX_H2_SKIP & 4)) {
                    // (a unit's registers are asked for again right behind its last piece)
                    static_for<0, NU>([&](auto u_c) __attribute__((always_inline)) {
                        constexpr int u = decltype(u_c)::value;
                        if (s == 1 + (((u < FULL ? NQ : NQ1) - 1) * NU + u) * STEP + 1 && chunk + 2 < c_end)
                            x_load(u, chunk + 2);
                    });
                }
            });
            if (!LEAN && j == NJ - 1 && MORE && !(STX_H2_SKIP & 2)) a_load(ky, ky, chunk + 1);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    using b0 = std::integral_constant<int, 0>;
    using b1 = std::integral_constant<int, 1>;
#ifdef STX_H2_TIMING
    const long long t_loop = clock64(), w_loop = wall_clock64();
#endif
    // stx_clock_marks: the workgroup in the middle of the launch times its chunk loop with both counters
    const bool mark = a.clock_out != nullptr && (int)blockIdx.x == (int)(gridDim.x >> 1);
    long long mark_c = 0, mark_w = 0;
    if (mark) {
        mark_c = clock64();
        mark_w = wall_clock64();
    }
    {
        int chunk = c_begin;
        for (; chunk + 2 < c_end; chunk += 2) {
            run_chunk(b0{}, chunk, yes{});
            run_chunk(b1{}, chunk + 1, yes{});
        }
        // (an even number of chunks per launch or K slice: h2_usable, the slice bounds above)
        run_chunk(b0{}, chunk, yes{});
        run_chunk(b1{}, chunk + 1, no{});
    }
    if (mark) {
        const long long dc = clock64() - mark_c, dw = wall_clock64() - mark_w;
        if (tid == 0) {
            a.clock_out[0] = dw >= 100 ? dc : 0;
            a.clock_out[1] = dw >= 100 ? dw : 0;
        }
    }
#ifdef STX_H2_TIMING
    const long long t_end = clock64(), w_end = wall_clock64();
#endif

    float amax = 0.f;                          // max |y| over what this lane stores
    f32x4 *ex = reinterpret_cast<f32x4 *>(ldsb);

    // One pass = the accumulators of one 32-channel block per wave.  Everything the pass reads from
    // memory (bias; ReLU mask and style term: 128 KB each per workgroup, which all workgroups of a round
    // want at the same moment) is requested BEFORE the exchange and lands during it -- read where it
    // is used, every one of the sixteen loads of a lane exposed its full latency: 28 k cycles per
    // epilogue of two passes instead of 12 k.
#ifdef STX_H2_TIMING
    long long t_epi[4] = {0, 0, 0, 0};
#endif
    auto epilogue_pass = [&](auto pass_c, auto even_c) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_c)::value;
        constexpr int mbi = pass % MB, pbi = pass / MB;        // channel block of the wave, patch of the stack
        const LaneRows lr = pbi == 0 && HOIST ? rows0 : lane_rows(pbi);
        if (!HOIST && pass == 0) content_columns(lr);
        int crow[2] = {0, 0};                                  // common.h: content_index, once per lane and pass
        if (EPI == kEpiDgradInject && content) {
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const bool ok = lr.yy + y < a.H && xx0 < a.W;
                int q = (cw_oy + (ok ? lr.yy + y : 0)) % cw_ch;
                crow[y] = (q < 0 ? q + cw_ch : q) * cw_cw;
            }
        }
        const int cblk = m0 + (hh * MB + mbi) * 32;
        auto chan = [&](int n) __attribute__((always_inline)) {     // n = 4 rqi + e
            const int c0 = cblk + (n & 3) + 8 * (2 * rqp + (n >> 2));
            return sgpr(c0 < M_ ? c0 : M_);
        };
        float bs[8];
        f32x2 mk[16], sg[16];
        unsigned mkb[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int c = chan(n);
            const unsigned so = (unsigned)c * HW4;
            if (EPI == kEpiForward) {
                bs[n] = a.bias ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                               rbias, (unsigned)half * 16u, (unsigned)c * 4u, 0))
                               : 0.f;
            } else if (EPI != kEpiPartial) {
                if (a.mask_codes) mkb[n] = __builtin_amdgcn_raw_buffer_load_b8(rmc, lr.vmc, (unsigned)(c * cph * cpw), 0);
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    if (!a.mask_codes && a.mask && !(STX_H2_SKIP & 16)) mk[2 * n + y] = ld2(rmask, lr, y, so, even_c);
                    if (EPI == kEpiDgradInject && a.inj.sgrad) sg[2 * n + y] = ld2(rsg, lr, y, so, even_c);
                }
            }
        }
        STX_H2_STAMP(0);
        __syncthreads();          // the last B reads (pass 0) / the previous pass's reads are done
        STX_H2_STAMP(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                ex[(((xi * 2 + wh) * 4 + j) * 4 + rq) * 64 + lane] =
                    f32x4{acc[mbi][4 * pbi + j][4 * rq], acc[mbi][4 * pbi + j][4 * rq + 1],
                          acc[mbi][4 * pbi + j][4 * rq + 2], acc[mbi][4 * pbi + j][4 * rq + 3]};
        __syncthreads();
        STX_H2_STAMP(2);
#pragma unroll
        for (int rqi = 0; rqi < 2; ++rqi) {
            const int rq = 2 * rqp + rqi;
            f32x4 o[2][2];                     // [row s][column]: four channels each
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 p[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) p[c] = ex[(((c * 2 + hh) * 4 + 2 * mrow + s) * 4 + rq) * 64 + lane];
                o[s][0] = (p[0] + p[1] + p[2]) * out_scale * out_scale2;
                o[s][1] = (p[1] - p[2] - p[3]) * out_scale * out_scale2;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = 4 * rqi + e;
                const int c = chan(n);
                const unsigned so = (unsigned)c * HW4;
                f32x2 v[2];
#pragma unroll
                for (int y = 0; y < 2; ++y) v[y] = f32x2{o[y][0][e], o[y][1][e]};
                if (EPI == kEpiForward) {
                    if (a.bias) {
#pragma unroll
                        for (int y = 0; y < 2; ++y) v[y].x += bs[n], v[y].y += bs[n];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int y = 0; y < 2; ++y) v[y].x = fmaxf(v[y].x, 0.f), v[y].y = fmaxf(v[y].y, 0.f);
                    }
                } else if (EPI != kEpiPartial) {
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        if (a.mask_codes) {
                            const unsigned nib = mkb[n] >> (2 * y);
                            v[y].x = (nib & 1u) ? v[y].x : 0.f;
                            v[y].y = (nib & 2u) ? v[y].y : 0.f;
                        } else if (a.mask && !(STX_H2_SKIP & 16)) {
                            v[y].x = mk[2 * n + y].x > 0.f ? v[y].x : 0.f;
                            v[y].y = mk[2 * n + y].y > 0.f ? v[y].y : 0.f;
                        }
                        if (EPI == kEpiDgradInject) {
                            if (content) {
                                // (one layer per tile evaluation takes this: read where it is used)
                                const f32x2 ft = ld2(rft, lr, y, so, even_c);
                                const int mm = c + 4 * half;
                                const int cm = mm < a.M ? mm : 0;     // (lanes past M store nothing)
                                const float *cp = content + (size_t)cm * cw_ch * cw_cw + crow[y];
                                v[y].x += c_scale * (ft.x - cp[ccol[0]]);
                                v[y].y += c_scale * (ft.y - cp[ccol[1]]);
                            }
                            if (a.inj.sgrad) {
                                v[y].x += s_scale * sg[2 * n + y].x;
                                v[y].y += s_scale * sg[2 * n + y].y;
                            }
                        }
                    }
                }
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    if (EPI != kEpiForward || !a.skip_y) st2(ry, lr, y, so, v[y], even_c);
                    if (EPI != kEpiPartial)
                        amax = fmaxf(amax, fmaxf(lr.vo[y][0] != kOob ? fabsf(v[y].x) : 0.f,
                                                 lr.vo[y][1] != kOob ? fabsf(v[y].y) : 0.f));
                }
                // the ReLU sign nibble of the lane's window, for the backward pass of the layer that
                // reads this blob: one byte instead of 16 bytes of fp32 mask per lane and channel
                if (EPI == kEpiForward && a.out_codes) {
                    const unsigned nib = (lr.vo[0][0] != kOob && v[0].x > 0.f ? 1u : 0u) |
                                         (lr.vo[0][1] != kOob && v[0].y > 0.f ? 2u : 0u) |
                                         (lr.vo[1][0] != kOob && v[1].x > 0.f ? 4u : 0u) |
                                         (lr.vo[1][1] != kOob && v[1].y > 0.f ? 8u : 0u);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, roc, lr.vmc, (unsigned)(c * cph * cpw), 0);
                }
                // the lane's 2x2 outputs are one window of the 2x2/2 pooling layer that follows
                // (pool.hip's arithmetic; ceil mode: the second row may be missing)
                if (EPI == kEpiForward && a.pool_out) {
                    // (hx: the window's second column exists -- always on even planes)
                    const bool hy = lr.yy + 1 < a.H, hx = decltype(even_c)::value || xx0 + 1 < a.W;
                    float pr;
                    if (a.pool_mode == STX_POOL_MAX) {
                        pr = hx ? fmaxf(v[0].x, v[0].y) : v[0].x;
                        if (hy) pr = hx ? fmaxf(fmaxf(pr, v[1].x), v[1].y) : fmaxf(pr, v[1].x);
                    } else {
                        // (the sum in pool.hip's order, the divisor the number of elements the clipped window has)
                        pr = (v[0].x + (hx ? v[0].y : 0.f) + (hy ? v[1].x : 0.f) + (hx && hy ? v[1].y : 0.f)) *
                             (hy ? (hx ? 0.25f : 0.5f) : (hx ? 0.5f : 1.0f));
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pr), rpool, lr.vpool,
                                                          (unsigned)c * (unsigned)(ph * pw * 4), 0);
                    if (a.pool_codes) {
                        const unsigned code = a.pool_mode == STX_POOL_MAX
                                                  ? pool_max_code(v[0].x, v[0].y, v[1].x, v[1].y, hx, hy)
                                                  : pool_ave_code(v[0].x, v[0].y, v[1].x, v[1].y, hx, hy);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)code, rcodes, lr.vpool >> 2,
                                                             (unsigned)c * (unsigned)(ph * pw), 0);
                    }
                }
            }
        }
    };
    static_for<0, MB * PB>([&](auto pass_c) __attribute__((always_inline)) {
        [[maybe_unused]] constexpr int pass = decltype(pass_c)::value;
        if (weven) epilogue_pass(pass_c, yes{});
        else epilogue_pass(pass_c, no{});
        STX_H2_STAMP(3);
    });
    // max |y| of this launch's output, for the kernel that reads it next: one atomic per workgroup
    // (thousands of atomics on a few words take longer than the kernel's last round)
    if (EPI != kEpiPartial && a.y_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
        float *wmax = reinterpret_cast<float *>(ldsb + kLdsBytes - 64);
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < 8; ++w) amax = fmaxf(amax, wmax[w]);
            atomicMax(a.y_amax + (blockIdx.x & (kAmaxSlots - 1)), __builtin_bit_cast(unsigned, amax));
        }
    }
#ifdef STX_H2_TIMING
    if (blockIdx.x == gridDim.x - 3 && lane == 0) {     // a workgroup of the last round
        g_h2_timing[wave][0] = t_loop - t_start, g_h2_timing[wave][1] = t_end - t_loop;
        g_h2_timing[wave][2] = clock64() - t_end;
        g_h2_timing[wave][3] = w_loop - w_start, g_h2_timing[wave][4] = w_end - w_loop;
        g_h2_timing[wave][5] = wall_clock64() - w_end;
        for (int i = 0; i < 4; ++i) g_h2_epi[wave][i] = t_epi[i] - t_end;
        g_h2_epi[wave][4] = t_p0 - t_start, g_h2_epi[wave][5] = t_p1 - t_p0, g_h2_epi[wave][6] = t_p2 - t_p1, g_h2_epi[wave][7] = t_p3 - t_p2;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// max |x| of an array into kAmaxSlots words of float bits (zeroed here first): what conv_h2_kernel needs
// of its input when the kernel that wrote it left nothing (the first layer's output, a pooled blob
// whose producer did not fuse, the gradient the loss terms start from).
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, unsigned *slots) {
    float m = 0.f;
    // 16-byte loads over the aligned middle; the (at most three) floats in front of it and behind it one by one
    const size_t head = ((16 - (reinterpret_cast<size_t>(x) & 15)) & 15) >> 2;
    const size_t h = head < n ? head : n;
    const size_t n4 = (n - h) >> 2;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + h);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) m = fmaxf(m, fabsf(x[threadIdx.x]));
        const size_t tail = h + (n4 << 2);
        if (tail + threadIdx.x < n && threadIdx.x < 4) m = fmaxf(m, fabsf(x[tail + threadIdx.x]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)          // one atomic per block: they serialise on the few words
        atomicMax(slots + (blockIdx.x & (kAmaxSlots - 1)),
                  __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

int absmax_launch(hipStream_t s, const float *x, size_t n, unsigned *slots) {
    STX_HIP(hipMemsetAsync(slots, 0, kAmaxSlots * sizeof(unsigned), s));
    if ((reinterpret_cast<size_t>(x) & 3) != 0) {
        set_error("absmax_launch: the array is not float-aligned");
        return STX_ERR_ARG;
    }
    const int blocks = (int)std::min<size_t>((n / 4 + 2047) / 2048 + 1, 1024);
    absmax_kernel<<<blocks, 256, 0, s>>>(x, n, slots);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ------------------------------------------------------------------------------------------------
size_t h2_packed_floats(int K, int M) {
    return (size_t)ceil_div(M, 32) * ceil_div(K, KC) * (U_BLK / 4) + kHeaderFloats;
}

// (G g)[xi] of kernel row ky of the filter W(m, k), fp32 -- the same expression in both passes
__device__ __forceinline__ float h2_u(const float *__restrict__ w, int Ko, int transpose_flip, int m, int k,
                                      int ky, int x) {
    float g[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int t = ky * 3 + b;
        g[b] = transpose_flip ? w[((size_t)k * Ko + m) * 9 + (8 - t)] : w[((size_t)m * Ko + k) * 9 + t];
    }
    return x == 0 ? g[0] : x == 1 ? (g[0] + g[1] + g[2]) * 0.5f : x == 2 ? (g[0] - g[1] + g[2]) * 0.5f : g[2];
}

__global__ void h2_bank_max_kernel(const float *__restrict__ w, int Ko, int transpose_flip, int M, int K,
                                   unsigned *header) {
    const size_t total = (size_t)M * K * 12;
    float mx = 0.f;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int x = idx % 4, ky = (idx / 4) % 3;
        const size_t mk = idx / 12;
        mx = fmaxf(mx, fabsf(h2_u(w, Ko, transpose_flip, (int)(mk / K), (int)(mk % K), ky, x)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(header, __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// packed[mblk][chunk][ky][xi][piece][half][l31][e] (fp16) = piece of s (G g)[xi] of kernel row ky of the
// filter W(m = mblk*32 + l31, k = chunk*16 + half*8 + e); header[1] = the exponent of s
__global__ void h2_pack_kernel(const float *__restrict__ w, int Ko, int transpose_flip, int M, int K,
                               int n_chunks, unsigned short *__restrict__ packed, size_t total,
                               unsigned *header) {
    const int ew = h2_scale_exp(header[0]);
    const float sw = pow2f(ew);
    if (blockIdx.x == 0 && threadIdx.x == 0) header[1] = (unsigned)ew;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int e = r % 8; r /= 8;
        const int l31 = r % 32; r /= 32;
        const int half = r % 2; r /= 2;
        const int piece = r % 2; r /= 2;
        const int x = r % 4; r /= 4;
        const int ky = r % 3; r /= 3;
        const int chunk = r % n_chunks;
        const int mblk = r / n_chunks;
        const int m = mblk * 32 + l31, k = chunk * KC + half * 8 + e;
        _Float16 out = (_Float16)0.f;
        if (m < M && k < K) {
            const float u = h2_u(w, Ko, transpose_flip, m, k, ky, x) * sw;
            const _Float16 hi = (_Float16)u;
            out = piece == 0 ? hi : (_Float16)(u - (float)hi);
        }
        packed[idx] = __builtin_bit_cast(unsigned short, out);
    }
}

int h2_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip, float *packed) {
    const int M = transpose_flip ? Ko : Mo;
    const int K = transpose_flip ? Mo : Ko;
    const size_t bank_floats = h2_packed_floats(K, M) - kHeaderFloats;
    unsigned *header = reinterpret_cast<unsigned *>(packed + bank_floats);
    STX_HIP(hipMemsetAsync(header, 0, kHeaderFloats * sizeof(float), s));
    h2_bank_max_kernel<<<(int)std::min<size_t>(((size_t)M * K * 12 + 255) / 256, 256), 256, 0, s>>>(
        w_caffe, Ko, transpose_flip, M, K, header);
    STX_CHECK_LAUNCH();
    const size_t total = bank_floats * 2;
    h2_pack_kernel<<<(int)std::min<size_t>((total + 255) / 256, 8192), 256, 0, s>>>(
        w_caffe, Ko, transpose_flip, M, K, ceil_div(K, KC), reinterpret_cast<unsigned short *>(packed), total,
        header);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ids 300: 64 channels x 8 x 32 pixels per workgroup, 301: 128 channels x 8 x 32, 302: 64 channels x 16 x 32
ConvConfig h2_config(int mb, int pb) {
    ConvConfig c;
    c.id = mb == 2 ? 301 : pb == 2 ? 302 : 300;
    c.bm = mb == 2 ? 128 : 64;
    c.kc = KC;
    c.pr = c.id == 302 ? 16 : 8;
    c.pc = PC;
    c.threads = NT;
    c.lds_bytes = c.id == 302 ? Geo<2>::kLds : Geo<1>::kLds;
    return c;
}

// What the kernel takes: a multiple of 32 input channels (the chunk loop runs two chunks per trip), plane sets under 2 GiB, no ReLU nibbles.
bool h2_usable(const ConvProblem &p) {
    if (p.ksize != 3 || p.K % (2 * KC) != 0 || p.K < 2 * KC) return false;
    if (p.epilogue != kEpiForward && p.epilogue != kEpiDgrad) return false;
    // (it reads ReLU sign nibbles in place of the fp32 mask; it does not write them)
    if (p.epilogue == kEpiForward && (p.wants_codes || p.in_codes)) return false;
    const double xb = 4.0 * p.K * (double)p.H * p.W, yb = 4.0 * p.M * (double)p.H * p.W;
    const double wb = 4.0 * (double)h2_packed_floats(p.K, p.M);
    return xb < 2147483648.0 && yb < 2147483648.0 && wb < 2147483648.0;
}

// K split of a launch and the channel tiling: the round model of conv_wino2.hip with this kernel's
// measured chunk (16 channels: 2.0 us for 64 channels x 256 pixels, 2.75 us for 128 channels, 3.5 us for
// two stacked patches) and its
// prologue + epilogue (8.5 / 17 us) -- whole rounds of 256 workgroups plus the reduce pass over the
// slices.  Shape and epilogue only, never timing.
static double h2_plan(const ConvConfig &cfg, const ConvProblem &p, int *factor) {
    const int n_pairs = p.K / (2 * KC);
    const long n = (long)ceil_div(p.M, cfg.bm) * ceil_div(p.H, cfg.pr) * ceil_div(p.W, PC);
    const double out_mb = 4e-6 * p.M * (double)p.H * p.W;
    // (128 channels or two stacked patches: twice the matrix work per chunk; the stacked patches also
    // stage twice as much, 7 000 cycles per chunk against 5 400)
    const double t_chunk = cfg.id == 301 ? 2.75 : cfg.id == 302 ? 3.5 : 2.0, t_fixed = cfg.id != 300 ? 17.0 : 8.5;
    const bool may_split = p.epilogue == kEpiForward || p.epilogue == kEpiDgrad;
    double best_cost = 0;
    int best = 1;
    for (int f = 1; f <= 8 && (f == 1 || (may_split && n_pairs / f >= 2)); ++f) {
        const double rounds = (double)ceil_div((int)std::min<long>(n * f, 1 << 30), 256);
        double c = rounds * (2.0 * ceil_div(n_pairs, f) * t_chunk + t_fixed);
        if (f > 1) c += (f + 1) * out_mb / 3.0 + 5.0;      // reduce pass at ~3 TB/s + its launch
        if (f == 1 || c < best_cost) {
            best_cost = c;
            best = f;
        }
    }
    *factor = best;
    return best_cost;
}

int h2_splitk_factor(const ConvConfig &cfg, const ConvProblem &p) {
    int f;
    h2_plan(cfg, p, &f);
    return f;
}

// The cheapest of the three tilings by the round model: 128 channels where the channel count allows
// (the staged patch serves twice the matrix work), else two stacked patches (the filter fragments
// and the prologue do), else the plain one (small planes: more, shorter workgroups).
ConvConfig h2_pick_config(const ConvProblem &p) {
    int f;
    ConvConfig best = h2_config(1, 1);
    double best_cost = h2_plan(best, p, &f);
    const ConvConfig cand[2] = {h2_config(2, 1), h2_config(1, 2)};
    for (int i = 0; i < 2; ++i) {
        if (i == 0 && p.M % 128 != 0) continue;
        const double c = h2_plan(cand[i], p, &f);
        if (c < best_cost) best_cost = c, best = cand[i];
    }
    return best;
}

// A backward launch that can take its input pooled (ConvProblem::pin_codes): unsplit ones -- the K slices
// of a split launch would each un-pool the same patch again, and the planes that split are small.
bool h2_takes_pooled_input(const ConvConfig &cfg, const ConvProblem &p) {
    return cfg.id >= 300 && p.epilogue == kEpiDgrad && h2_usable(p) && h2_splitk_factor(cfg, p) == 1;
}

bool h2_fuses_pool(const ConvProblem &p) {
    // (odd planes too: the last window of a row has one column, as the last of a column may have one row)
    return p.pool_out && p.epilogue == kEpiForward && (((size_t)p.y | (size_t)p.pool_out) & 7) == 0;
}

template <int EPI, int MB, int PB, int PIN = 0>
static int h2_launch_epi(hipStream_t s, const WinoArgs &args, int n_wg) {
    auto kern = conv_h2_kernel<EPI, MB, PB, PIN>;
    constexpr size_t kLdsBytes = Geo<PB>::kLds;
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

#ifdef STX_H2_DEV_SUBSET   // (register / ISA studies: tools/h2_regs.sh, tools/spill_map.py -- a few variants, in seconds)
template __global__ void conv_h2_kernel<0, 2, 1, 0>(WinoArgs);
template __global__ void conv_h2_kernel<0, 1, 2, 0>(WinoArgs);
template __global__ void conv_h2_kernel<1, 2, 1, 0>(WinoArgs);
template __global__ void conv_h2_kernel<3, 2, 1, 0>(WinoArgs);
template __global__ void conv_h2_kernel<3, 1, 2, 1>(WinoArgs);
#else
int h2_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    if (!h2_usable(p) || !p.x_amax) {
        set_error("h2_launch: unsupported problem (K %d, epilogue %d, input maximum %s)", p.K, p.epilogue,
                  p.x_amax ? "given" : "missing");
        return STX_ERR_UNSUPPORTED;
    }
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
    a.n_chunks = p.K / KC;
    a.tiles_x = ceil_div(p.W, PC);
    a.tiles_y = ceil_div(p.H, cfg.pr);
    a.m_tiles = ceil_div(p.M, cfg.bm);
    a.ksplit = 1;
    a.w_tile_stride = 0;
    a.relu = p.relu;
    a.inj = p.inject;
    a.pool_out = nullptr;
    a.pool_codes = nullptr;
    a.pool_mode = p.pool_mode;
    a.x_bytes = (int)(4.0 * p.K * (double)p.H * p.W);
    const bool pin = p.pin_codes != nullptr;
    if (pin) {
        if (!h2_takes_pooled_input(cfg, p) || ksplit > 1) {
            set_error("h2_launch: a pooled input goes with unsplit backward launches only");
            return STX_ERR_UNSUPPORTED;
        }
        a.x_bytes = (int)(4.0 * p.K * (double)((p.H + 1) / 2) * ((p.W + 1) / 2));
        a.pin_codes = p.pin_codes;
        a.pin_mode = p.pin_mode;
        a.pin_mask = p.pin_mask ? 1 : 0;
    }
    a.w_bytes = (int)(4 * (h2_packed_floats(p.K, p.M) - kHeaderFloats));
    a.clock_out = p.clock_out;
    a.x_amax = p.x_amax;
    a.y_amax = p.y_amax;
    a.mask_codes = p.epilogue == kEpiDgrad ? p.mask_codes : nullptr;
    a.out_codes = p.epilogue == kEpiForward && p.relu ? p.out_codes : nullptr;
    const bool inject = p.epilogue == kEpiDgrad && (p.inject.sgrad || p.inject.content);
    int n_wg = a.m_tiles * a.tiles_x * a.tiles_y;
    const bool split = ksplit > 1 && p.splitk_ws && p.splitk_ws_floats >= (size_t)ksplit * p.M * p.H * p.W;
    if (split) {
        a.ksplit = ksplit;
        a.y = p.splitk_ws;
        a.y_amax = nullptr;
        a.mask_codes = nullptr;        // (the reduce pass masks, from the fp32 blob)
        a.out_codes = nullptr;         // (... and writes no nibbles: conv_writes_out_codes)
        n_wg *= ksplit;
    } else if (h2_fuses_pool(p)) {
        a.pool_out = p.pool_out;
        a.pool_codes = p.pool_codes;
        a.skip_y = p.skip_y && p.pool_codes != nullptr;
    }
    const int epi = split ? kEpiPartial : inject ? kEpiDgradInject : p.epilogue;
#define STX_H2_CASE(E)                                                                   \
    case E:                                                                              \
        STX_TRY(cfg.id == 301   ? (h2_launch_epi<E, 2, 1>(s, a, n_wg))                   \
                : cfg.id == 302 ? (h2_launch_epi<E, 1, 2>(s, a, n_wg))                   \
                                : (h2_launch_epi<E, 1, 1>(s, a, n_wg)));                 \
        break;
#define STX_H2_PIN_CASE(E, P)                                                            \
    case E:                                                                              \
        STX_TRY(cfg.id == 301   ? (h2_launch_epi<E, 2, 1, P>(s, a, n_wg))                \
                : cfg.id == 302 ? (h2_launch_epi<E, 1, 2, P>(s, a, n_wg))                \
                                : (h2_launch_epi<E, 1, 1, P>(s, a, n_wg)));              \
        break;
    if (pin && p.pin_mode == STX_POOL_MAX) {
        switch (epi) {
            STX_H2_PIN_CASE(kEpiDgrad, 1 + STX_POOL_MAX)
            STX_H2_PIN_CASE(kEpiDgradInject, 1 + STX_POOL_MAX)
        }
        return STX_OK;
    }
    if (pin) {
        switch (epi) {
            STX_H2_PIN_CASE(kEpiDgrad, 1 + STX_POOL_AVE)
            STX_H2_PIN_CASE(kEpiDgradInject, 1 + STX_POOL_AVE)
        }
        return STX_OK;
    }
#undef STX_H2_PIN_CASE
    switch (epi) {
        STX_H2_CASE(kEpiForward)
        STX_H2_CASE(kEpiDgrad)
        STX_H2_CASE(kEpiDgradInject)
        STX_H2_CASE(kEpiPartial)
        default:
            set_error("h2_launch: no kernel for epilogue %d", p.epilogue);
            return STX_ERR_UNSUPPORTED;
    }
#undef STX_H2_CASE
    return split ? splitk_reduce_launch(s, p, ksplit) : STX_OK;
}
#endif

}  // namespace stx
