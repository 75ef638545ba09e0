// 3x3 convolution on the bf16 matrix cores at fp32 accuracy: 1-D Winograd F(2,3) along x with
// both operands split into three bf16 pieces (bf16x3.h).
//
// Same job and interface as conv_wino2_kernel (forward + bias + ReLU, backward-to-data + ReLU
// mask), for layers with a multiple of 16 input channels.  For a pair of neighbouring outputs
// (x = 2t, 2t+1) of one row and the three taps g0, g1, g2 of kernel row ky:
//     d0..d3 = in[2t-1 .. 2t+2]          (row y + ky - 1)
//     V = [d0-d2, d1+d2, d2-d1, d1-d3]   U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
//     M[xi] = sum over input channels and ky of U[ky][xi] * V[ky][xi]
//     out[2t] = M0 + M1 + M2             out[2t+1] = M1 - M2 - M3
// V and U are formed in fp32 exactly as an fp32 Winograd kernel forms them; each is then written
// as x1 + x2 + x3 (bf16 pieces, residuals exact) and the product of a U and a V value is the sum
// of six exact piece products accumulated in fp32 -- what is dropped lies below 2^-24 of the
// product.  Matrix time per output and channel pair: 3 (ky) x 4/2 (xi per pixel) x 6 products on
// v_mfma_f32_32x32x16_bf16 = 36 units of 1/512 cycle against 4 units of 1/32 cycle for the fp32
// 2-D Winograd kernel: 0.56 of its matrix time, with the vector work of the split hidden behind
// the bf16 MFMAs (it is not hidden behind fp32 MFMAs).  The 2-D form of this (24 units) does not
// fit a CU: its transformed operands need 192 KB of LDS per 16 channels and more than the
// ~32 B/clk a CU can load.
//
// Work split.  A workgroup of eight waves computes 64 channels x (8 rows x 32 columns) = 128
// x-tiles.  Wave (xi, mb) owns transform component xi of the 32 channels of block mb for all
// four pixel blocks (2 rows x 16 tiles each): 4 accumulators of one 32 x 32 MFMA block.  Its A
// operand (the U pieces of its component and channel block) is nobody else's, so it comes straight
// from global memory into registers, as ready fragments (1 KB per piece and step), one chunk
// ahead.  The B operand (V pieces, shared by the two channel blocks) goes through LDS:
// [xi][piece][row][tile][16 channels], double buffered, one barrier per chunk of 16 channels; a
// staging thread loads the four inputs of one tile for 8 channels (eight 16-byte loads),
// transforms, splits and writes twelve 16-byte pieces.
//
// Epilogue.  The four components of an output pair live in four waves: all accumulators go
// through LDS once (128 KB), wave (xi', mb) then finishes pixel block xi' of channel block mb.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

// (round 4: moved out of libstx.so; compile with -DSTX_EXPERIMENT_BF3, which restores the two
// fields of ConvProblem / WinoArgs this kernel used; tools/ubench/bf3conv_bench.hip does)
#ifdef STX_EXPERIMENT_BF3      // (without it this file compiles to nothing)
#include "bf16x3.h"
#include "common.h"

namespace stx {
// 1-D Winograd on the bf16 matrix cores with three-piece operands (conv_bf3.hip); config id 300.
ConvConfig bf3_config();
size_t bf3_packed_floats(int K, int M);
int bf3_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip, float *packed);
int bf3_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit);
bool bf3_usable(const ConvProblem &p);      // what the kernel takes (shape, epilogue, addressing)
// The pre-split form: bf3_split_launch writes the transformed, three-piece operand of a whole layer
// once (bf3_split_bytes of scratch), bf3_launch with p.x_split set reads it instead of p.x.
size_t bf3_split_bytes(int K, int H, int W);
int bf3_split_launch(hipStream_t s, const float *x, int K, int H, int W, void *split);
}  // namespace stx

#ifndef STX_BF3_HILO
#define STX_BF3_HILO 0   // 1: the five small products of a step go into accumulators of their own
#endif
#ifndef STX_BF3_SKIP
#define STX_BF3_SKIP 0   // timing experiments (tools/ubench/bf3conv_bench.hip): 1 no staging, 2 no filter
#endif                  // loads, 4 no patch loads in the main loop.  Wrong results when non-zero.

namespace stx {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));

constexpr int KC = 16;                      // input channels per chunk = one MFMA k-step
constexpr int BM = 64, NT = 512;
constexpr int PR = 8, PC = 32, TX = PC / 2; // pixel patch; x-tiles per patch row
constexpr int XR = PR + 2;                  // input rows of the patch
constexpr int RT = XR * TX;                 // (row, tile) positions of V: 160
constexpr int V_PIECE = RT * 32;            // bytes of one [rt][16 ch] bf16 array
constexpr int V_BYTES = 4 * 3 * V_PIECE;    // [xi][piece]: 60 KB
constexpr int FRAG = 1024;                  // one operand fragment: 64 lanes x 8 bf16
constexpr int U_STEP = 4 * 2 * 3 * FRAG;    // [xi][mb][piece] of one (chunk, ky): 24 KB
constexpr int U_CHUNK = 3 * U_STEP;         // 72 KB
constexpr int EX_BYTES = 4 * 2 * 4 * 4 * 64 * 16;   // epilogue exchange: 128 KB
constexpr size_t kLdsBytes = EX_BYTES > 2 * V_BYTES ? EX_BYTES : 2 * V_BYTES;

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

#ifdef STX_BF3_TIMING   // cycle counters for tools/ubench/bf3conv_bench.hip
__device__ long long g_bf3_timing[8][8];
#endif

// PRE: the B operand comes pre-split (bf3_split_kernel below wrote the V pieces of the whole layer
// once, instead of every channel tile's workgroup transforming and splitting its patch again -- an
// eighth of that vector work at 512 output channels): the staging of a chunk is then 60 16-byte-per-
// lane loads straight into LDS (`buffer_load ... lds`, 1 KB each), dealt out over the eight waves.
template <int EPI, bool PRE = false>
__global__ __launch_bounds__(NT) void conv_bf3_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
#ifdef STX_BF3_TIMING
    const long long t_start = clock64(), w_start = wall_clock64();
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int xi = wave & 3, mb = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_mfma.hip: channel tile fastest, so that the workgroups an
    // XCD runs at a time share input patches AND filter slices through its L2
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int ptile = sgpr(L / m_tiles);
    int mtile = L - ptile * m_tiles;
    if (m_tiles == 8 && ((a.tiles_x * a.tiles_y) & 7) == 0) {    // 4 channel tiles x 8 patches per round
        const int g = L >> 5, r = L & 31;
        mtile = (g & 1) * 4 + (r & 3);
        ptile = (g >> 1) * 8 + (r >> 2);
    }
    const int y0 = sgpr((ptile / a.tiles_x) * PR);
    const int x0 = sgpr((ptile % a.tiles_x) * PC);
    const int m0 = mtile * BM;
    const int HW = a.H * a.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int n_chunks = a.n_chunks;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- staging role.  A unit = position rt of the V array x four channels (quad): 640 units per
    // chunk; every thread has unit tid, the threads of waves 0 and 1 also unit 512 + tid.
    const int n_units = wave < 2 ? 2 : 1;                 // wave-uniform
    const bool edge = x0 == 0 || x0 + PC + 2 > a.W;       // workgroup-uniform
    unsigned xvoff[2], v_dst[2];
    bool left[2], ok2[2], ok3[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int u = tid + n * NT;
        const int quad = u & 3, rt = u >> 2;
        const int st_r = rt / TX, st_t = rt % TX;
        const int st_y = y0 - 1 + st_r, st_x = x0 + 2 * st_t - 1;
        left[n] = st_x < 0;                                // x = -1: loaded from x = 0 and shifted
        ok2[n] = st_x + 2 < a.W, ok3[n] = st_x + 3 < a.W;
        xvoff[n] = kOob;
        if (rt < RT && (unsigned)st_y < (unsigned)a.H && st_x + 1 < a.W)
            xvoff[n] = (unsigned)(quad * 4 * HW + st_y * a.W + (left[n] ? 0 : st_x)) * 4u;
        v_dst[n] = (unsigned)(rt * 32 + quad * 8);
    }

    f32x4 xr[2][4];
    auto x_load = [&](int n, int chunk) __attribute__((always_inline)) {
        const unsigned xs = (unsigned)sgpr(chunk * KC) * HW4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xr[n][i] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[n], xs + (unsigned)i * HW4, 0));
    };
    // One piece of the staging work of unit n: component c of channel pair pr (13 vector
    // instructions), plus -- behind the second pair -- the three 8-byte writes of the component.
    unsigned pk[3][2];
    auto piece = [&](int n, int c, int pr, char *vbuf, bool fix) __attribute__((always_inline)) {
        if (fix && c == 0 && pr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 d = xr[n][i];
                f32x4 e;
                e.x = left[n] ? 0.f : d.x;
                e.y = left[n] ? d.x : d.y;
                e.z = left[n] ? d.y : d.z;
                e.w = left[n] ? d.z : d.w;
                e.z = ok2[n] ? e.z : 0.f;
                e.w = ok3[n] ? e.w : 0.f;
                xr[n][i] = e;
            }
        }
        const f32x4 da = xr[n][2 * pr], db = xr[n][2 * pr + 1];
        const float va = c == 0 ? da.x - da.z : c == 1 ? da.y + da.z : c == 2 ? da.z - da.y : da.y - da.w;
        const float vb = c == 0 ? db.x - db.z : c == 1 ? db.y + db.z : c == 2 ? db.z - db.y : db.y - db.w;
        unsigned u1, u2, u3;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(va), "v"(vb));
        const float ra = va - __builtin_bit_cast(float, u1 << 16);
        const float rb = vb - __builtin_bit_cast(float, u1 & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u2) : "v"(ra), "v"(rb));
        const float sa = ra - __builtin_bit_cast(float, u2 << 16);
        const float sb = rb - __builtin_bit_cast(float, u2 & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u3) : "v"(sa), "v"(sb));
        pk[0][pr] = u1, pk[1][pr] = u2, pk[2][pr] = u3;
        if (pr == 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<u32x2v *>(vbuf + v_dst[n] + (c * 3 + q) * V_PIECE) = u32x2v{pk[q][0], pk[q][1]};
        }
    };
    auto stage_all = [&](int n, int buf) __attribute__((always_inline)) {
        char *vbuf = ldsb + buf * V_BYTES;
#pragma unroll
        for (int k = 0; k < 8; ++k) piece(n, k >> 1, k & 1, vbuf, edge);
    };

    // ---- A operand: fragments of (xi, mb), [ky][piece], straight from the packed bank
    const unsigned a_voff = (unsigned)((xi * 2 + mb) * 3 * FRAG + lane * 16);
    const unsigned w_base = (unsigned)mtile * (unsigned)a.w_tile_stride * 4u;
    bf16x8 af[3][3];
    auto a_load = [&](int ky, int chunk) __attribute__((always_inline)) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * U_CHUNK + (unsigned)ky * U_STEP));
#pragma unroll
        for (int q = 0; q < 3; ++q)
            af[ky][q] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, a_voff + q * FRAG, ws, 0));
    };

    // hi: the x1 y1 products; lo: the five small ones (their roundings happen 2^-8 further down)
    f32x16b hi[4], lo[STX_BF3_HILO ? 4 : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) hi[j][r] = 0.f, lo[STX_BF3_HILO ? j : 0][r] = 0.f;
#if STX_BF3_HILO
#define STX_LO(j) lo[j]
#else
#define STX_LO(j) hi[j]
#endif

    // B fragments of block blk = 4 ky + j of a chunk: rows 2 j + ky, 2 j + ky + 1 of the V array
    const unsigned b_base = (unsigned)(xi * 3 * V_PIECE + l31 * 32 + half * 16);
    bf16x8 bq[2][3];
    auto b_read = [&](int slot, int buf, int blk) __attribute__((always_inline)) {
        const char *vb = ldsb + buf * V_BYTES + b_base + (2 * (blk & 3) + (blk >> 2)) * TX * 32;
#pragma unroll
        for (int q = 0; q < 3; ++q) bq[slot][q] = *reinterpret_cast<const bf16x8 *>(vb + q * V_PIECE);
    };

    // ---- PRE: piece q = (xi', piece, row pair) of a chunk's V array, 1 KB: rows y0 - 1 + 2 rp and the
    // next one, sixteen x-tiles from x0 / 2, sixteen channels -- 512 contiguous bytes per row in the
    // split buffer [chunk][xi][piece][row][tile][16 ch]; wave w moves pieces w, w + 8, ...
    const unsigned plane_bytes_v = (unsigned)a.vp_rows * (unsigned)a.vp_tp * 32u;
    const unsigned dma_voff = (unsigned)((y0 + (lane >> 5)) * a.vp_tp + (x0 >> 1)) * 32u + (unsigned)(lane & 31) * 16u;
    auto dma = [&](int n, int buf, int chunk) __attribute__((always_inline)) {
        const int q = wave + 8 * n;                       // < 60
        const int comp = q / 5, rp = q - comp * 5;        // comp = xi' * 3 + piece
        const unsigned so = (unsigned)sgpr((chunk * 12 + comp)) * plane_bytes_v +
                            (unsigned)sgpr(rp * 2 * a.vp_tp * 32);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rx, (__attribute__((address_space(3))) void *)(ldsb + buf * V_BYTES + comp * V_PIECE + rp * 1024),
            16, dma_voff, so, 0, 0);
    };
    const int n_dma = wave < 4 ? 8 : 7;                   // 60 pieces over eight waves

    // ---- prologue
    if (PRE) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < n_dma) dma(n, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a_load(ky, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        x_load(0, 0);
        if (n_units == 2) x_load(1, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a_load(ky, 0);
        stage_all(0, 0);
        if (1 < n_chunks) x_load(0, 1);
        if (n_units == 2) {
            stage_all(1, 0);
            if (1 < n_chunks) x_load(1, 1);
        }
    }
    lds_barrier();
    b_read(0, 0, 0);

    // ---- main loop.  Twelve blocks of six MFMAs per chunk; the B fragments of a block are read
    // while the block before it runs; the hand-over barrier sits before the last block (every
    // wave has issued its last reads of this chunk and written its share of the next by then).
    // The staging pieces are dealt out behind the MFMAs; sched_barrier pins the order.
    auto run_chunk = [&](int buf, int chunk, auto nu_c, auto more_c, auto edge_c) __attribute__((always_inline)) {
        constexpr int NU = decltype(nu_c)::value;
        constexpr bool MORE = decltype(more_c)::value, EDGE = decltype(edge_c)::value;
        char *vnext = ldsb + (buf ^ 1) * V_BYTES;
#pragma unroll
        for (int blk = 0; blk < 12; ++blk) {
            const int ky = blk >> 2, j = blk & 3, slot = blk & 1;
            if (blk == 11) {
                __builtin_amdgcn_sched_barrier(0);
                // (PRE: this wave's pieces of the next chunk have landed -- they were requested
                // behind block 8, after the last first use of a filter fragment in this chunk, so
                // that the compiler's own waits for those never have to drain them)
                if (PRE && MORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (blk < 11) b_read(slot ^ 1, buf, blk + 1);
            else if (MORE) b_read(0, buf ^ 1, 0);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_barrier(0);
                if (m == 0) STX_LO(j) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][1], bq[slot][1], STX_LO(j), 0, 0, 0);
                if (m == 1) STX_LO(j) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][0], bq[slot][2], STX_LO(j), 0, 0, 0);
                if (m == 2) STX_LO(j) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][2], bq[slot][0], STX_LO(j), 0, 0, 0);
                if (m == 3) STX_LO(j) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][0], bq[slot][1], STX_LO(j), 0, 0, 0);
                if (m == 4) STX_LO(j) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][1], bq[slot][0], STX_LO(j), 0, 0, 0);
                if (m == 5) hi[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ky][0], bq[slot][0], hi[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int s = blk * 6 + m;
                if (PRE && MORE && s >= 49 && s < 57) {
                    if (s - 49 < 7 || wave < 4) dma(s - 49, buf ^ 1, chunk + 1);
                }
                if (!PRE && MORE && !(STX_BF3_SKIP & 1)) {
                    // unit 0: pieces behind slots 4, 8 (12) ...; unit 1 (waves 0, 1) in between
                    constexpr int STEP = NU == 2 ? 4 : 8;
                    if (s >= 4 && (s - 4) % STEP == 0) {
                        const int k = (s - 4) / STEP;             // 0 .. 8 NU - 1
                        if (k < 8 * NU) piece(k >> 3, (k >> 1) & 3, k & 1, vnext, EDGE);
                    }
                }
                if (!PRE && MORE && !(STX_BF3_SKIP & 4)) {
                    constexpr int STEP = NU == 2 ? 4 : 8;
                    if (s == 4 + 7 * STEP + 1 && chunk + 2 < n_chunks) x_load(0, chunk + 2);
                    if (NU == 2 && s == 4 + 15 * STEP + 1 && chunk + 2 < n_chunks) x_load(1, chunk + 2);
                }
            }
            if (j == 3 && MORE && !(STX_BF3_SKIP & 2)) a_load(ky, chunk + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    auto main_loop = [&](auto nu_c, auto edge_c) __attribute__((always_inline)) {
        int chunk = 0;
        for (; chunk + 2 < n_chunks; chunk += 2) {
            run_chunk(0, chunk, nu_c, yes{}, edge_c);
            run_chunk(1, chunk + 1, nu_c, yes{}, edge_c);
        }
        if (chunk + 1 < n_chunks) {
            run_chunk(0, chunk, nu_c, yes{}, edge_c);
            run_chunk(1, chunk + 1, nu_c, no{}, edge_c);
        } else {
            run_chunk(0, chunk, nu_c, no{}, edge_c);
        }
    };
    using one = std::integral_constant<int, 1>;
    using two = std::integral_constant<int, 2>;
#ifdef STX_BF3_TIMING
    const long long t_loop = clock64(), w_loop = wall_clock64();
#endif
    if (PRE) {
        main_loop(one{}, no{});
    } else if (n_units == 2) {
        if (edge) main_loop(two{}, yes{});
        else main_loop(two{}, no{});
    } else {
        if (edge) main_loop(one{}, yes{});
        else main_loop(one{}, no{});
    }
#ifdef STX_BF3_TIMING
    const long long t_end = clock64(), w_end = wall_clock64();
#endif
    f32x16b acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = STX_BF3_HILO ? hi[j] + lo[STX_BF3_HILO ? j : 0] : hi[j];
#undef STX_LO

    // ---- epilogue: components through LDS, [xi][mb][j][rq][lane] x (registers 4 rq .. 4 rq + 3)
    f32x4 *ex = reinterpret_cast<f32x4 *>(ldsb);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            ex[(((xi * 2 + mb) * 4 + j) * 4 + rq) * 64 + lane] =
                f32x4{acc[j][4 * rq], acc[j][4 * rq + 1], acc[j][4 * rq + 2], acc[j][4 * rq + 3]};
    __syncthreads();

    // this wave: pixel block xi (rows 2 xi, 2 xi + 1 of the patch) of channel block mb
    const int yy = y0 + 2 * xi + (l31 >> 4), xx = x0 + 2 * (l31 & 15);
    const bool weven = (a.W & 1) == 0;
    const int M_ = a.M;
    const unsigned plane_bytes = (unsigned)a.M * HW4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.mask), 0, a.mask ? (int)plane_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.bias), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    unsigned vo[2];
    {
        const unsigned lane_base = (unsigned)((4 * half) * HW + yy * a.W + xx) * 4u;
        vo[0] = (yy < a.H && xx < a.W) ? lane_base : kOob;
        vo[1] = (yy < a.H && xx + 1 < a.W) ? lane_base + 4u : kOob;
    }
    auto tail = [&](auto even_c) __attribute__((always_inline)) {
        constexpr bool EVEN = decltype(even_c)::value;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 p[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = ex[(((c * 2 + mb) * 4 + xi) * 4 + rq) * 64 + lane];
            const f32x4 o0 = p[0] + p[1] + p[2];
            const f32x4 o1 = p[1] - p[2] - p[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c0 = m0 + mb * 32 + e + 8 * rq;
                const int c = sgpr(c0 < M_ ? c0 : M_);
                const unsigned so = (unsigned)c * HW4;
                f32x2 v = {o0[e], o1[e]};
                if (EPI == kEpiForward) {
                    if (a.bias) {
                        const float bs = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(rbias, (unsigned)half * 16u,
                                                                        (unsigned)c * 4u, 0));
                        v.x += bs, v.y += bs;
                    }
                    if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                } else if (a.mask) {
                    f32x2 mk;
                    if (EVEN) {
                        mk = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rmask, vo[0], so, 0));
                    } else {
                        mk.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, vo[0], so, 0));
                        mk.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, vo[1], so, 0));
                    }
                    v.x = mk.x > 0.f ? v.x : 0.f;
                    v.y = mk.y > 0.f ? v.y : 0.f;
                }
                if (EVEN) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), ry, vo[0], so, 0);
                } else {
                    const float v0 = v.x, v1 = v.y;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ry, vo[0], so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), ry, vo[1], so, 0);
                }
            }
        }
    };
    if (weven) tail(std::integral_constant<bool, true>{});
    else tail(std::integral_constant<bool, false>{});
#ifdef STX_BF3_TIMING
    if (blockIdx.x == gridDim.x - 3 && lane == 0) {     // a workgroup of the last round
        g_bf3_timing[wave][0] = t_loop - t_start, g_bf3_timing[wave][1] = t_end - t_loop;
        g_bf3_timing[wave][2] = clock64() - t_end;
        g_bf3_timing[wave][3] = w_loop - w_start, g_bf3_timing[wave][4] = w_end - w_loop;
        g_bf3_timing[wave][5] = wall_clock64() - w_end;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// The B operand of a whole layer, transformed and split once: split[chunk][xi][piece][row][tile][16]
// (bf16), row 0 = image row -1, rows and tiles padded with zeros to whole patches, so that the
// convolution's workgroups fetch their V arrays without a predicate.  A thread = one (row, tile)
// position x 8 channels: eight 16-byte loads, twelve 16-byte stores.
static int bf3_split_rows(int H) { return ceil_div(H, PR) * PR + 2; }
static int bf3_split_tp(int W) { return ceil_div(W, PC) * TX; }

size_t bf3_split_bytes(int K, int H, int W) {
    return (size_t)(K / KC) * 12 * bf3_split_rows(H) * bf3_split_tp(W) * 32;
}

__global__ __launch_bounds__(256) void bf3_split_kernel(const float *__restrict__ x, int K, int H, int W,
                                                        int rows, int tp, char *__restrict__ split) {
    const int u = blockIdx.x * 256 + threadIdx.x;         // (chunk, row, tile, channel group)
    const int cg = u & 1, t = (u >> 1) % tp, r = ((u >> 1) / tp) % rows, chunk = (u >> 1) / (tp * rows);
    if (chunk >= K / KC) return;
    const int yy = r - 1, xx = 2 * t - 1;
    const size_t HW = (size_t)H * W;
    f32x4 d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float *px = x + (size_t)(chunk * KC + cg * 8 + i) * HW + (size_t)yy * W;
        const bool row = (unsigned)yy < (unsigned)H;
        d[i].x = row && xx >= 0 && xx < W ? px[xx] : 0.f;
        d[i].y = row && xx + 1 < W ? px[xx + 1] : 0.f;
        d[i].z = row && xx + 2 < W ? px[xx + 2] : 0.f;
        d[i].w = row && xx + 3 < W ? px[xx + 3] : 0.f;
    }
    const size_t plane = (size_t)rows * tp * 32;
    char *dst = split + (size_t)chunk * 12 * plane + ((size_t)r * tp + t) * 32 + cg * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = c == 0 ? d[i].x - d[i].z : c == 1 ? d[i].y + d[i].z : c == 2 ? d[i].z - d[i].y : d[i].y - d[i].w;
        bf16x8 q[3];
        split3_bf16(v, q[0], q[1], q[2]);
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<bf16x8 *>(dst + (c * 3 + n) * plane) = q[n];
    }
}

int bf3_split_launch(hipStream_t s, const float *x, int K, int H, int W, void *split) {
    const int rows = bf3_split_rows(H), tp = bf3_split_tp(W);
    const size_t units = (size_t)(K / KC) * rows * tp * 2;
    bf3_split_kernel<<<(unsigned)((units + 255) / 256), 256, 0, s>>>(x, K, H, W, rows, tp,
                                                                   static_cast<char *>(split));
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ------------------------------------------------------------------------------------------------
size_t bf3_packed_floats(int K, int M) {
    return (size_t)ceil_div(M, BM) * ceil_div(K, KC) * (U_CHUNK / 4);
}

// packed[mt][chunk][ky][xi][mb][piece][half][l31][e] (bf16) = piece of (G g)[xi] of kernel row ky of
// the filter W(m = mt*64 + mb*32 + l31, k = chunk*16 + half*8 + e)
__global__ void bf3_pack_kernel(const float *__restrict__ w, int Mo, int Ko, int transpose_flip,
                                int M, int K, int n_chunks, unsigned short *__restrict__ packed,
                                size_t total) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int e = r % 8; r /= 8;
        const int l31 = r % 32; r /= 32;
        const int half = r % 2; r /= 2;
        const int piece = r % 3; r /= 3;
        const int mb = r % 2; r /= 2;
        const int x = r % 4; r /= 4;
        const int ky = r % 3; r /= 3;
        const int chunk = r % n_chunks;
        const int mt = r / n_chunks;
        const int m = mt * BM + mb * 32 + l31, k = chunk * KC + half * 8 + e;
        unsigned short s[3] = {0, 0, 0};
        if (m < M && k < K) {
            float g[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int t = ky * 3 + b;
                g[b] = transpose_flip ? w[((size_t)k * Ko + m) * 9 + (8 - t)]
                                      : w[((size_t)m * Ko + k) * 9 + t];
            }
            const float u = x == 0 ? g[0]
                          : x == 1 ? (g[0] + g[1] + g[2]) * 0.5f
                          : x == 2 ? (g[0] - g[1] + g[2]) * 0.5f
                                   : g[2];
            split3_bf16_scalar(u, s[0], s[1], s[2]);
        }
        packed[idx] = s[piece];
    }
}

int bf3_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                     float *packed) {
    const int M = transpose_flip ? Ko : Mo;
    const int K = transpose_flip ? Mo : Ko;
    const size_t total = bf3_packed_floats(K, M) * 2;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    bf3_pack_kernel<<<blocks, 256, 0, s>>>(w_caffe, Mo, Ko, transpose_flip, M, K, ceil_div(K, KC),
                                           reinterpret_cast<unsigned short *>(packed), total);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

ConvConfig bf3_config() {
    ConvConfig c;
    c.id = 300;
    c.bm = BM;
    c.kc = KC;
    c.pr = PR;
    c.pc = PC;
    c.threads = NT;
    c.lds_bytes = kLdsBytes;
    return c;
}

// What the kernel takes: a multiple of 16 input channels, plain forward / backward epilogues,
// plane sets under 2 GiB.
bool bf3_usable(const ConvProblem &p) {
    if (p.ksize != 3 || p.K % KC != 0) return false;
    if (p.epilogue != kEpiForward && p.epilogue != kEpiDgrad) return false;
    if (p.inject.sgrad || p.inject.content) return false;
    const double xb = 4.0 * p.K * (double)p.H * p.W, yb = 4.0 * p.M * (double)p.H * p.W;
    const double wb = 4.0 * (double)bf3_packed_floats(p.K, p.M);
    return xb < 2147483648.0 && yb < 2147483648.0 && wb < 2147483648.0;
}

template <int EPI, bool PRE>
static int bf3_launch_epi(hipStream_t s, const WinoArgs &args, int n_wg) {
    auto kern = conv_bf3_kernel<EPI, PRE>;
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

int bf3_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    (void)cfg, (void)ksplit;
    if (!bf3_usable(p)) {
        set_error("bf3_launch: unsupported problem (K %d, epilogue %d)", p.K, p.epilogue);
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
    a.tiles_y = ceil_div(p.H, PR);
    a.m_tiles = ceil_div(p.M, BM);
    a.ksplit = 1;
    a.w_tile_stride = a.n_chunks * (U_CHUNK / 4);
    a.relu = p.relu;
    a.inj = p.inject;
    a.pool_out = nullptr;
    a.pool_codes = nullptr;
    a.pool_mode = p.pool_mode;
    a.x_bytes = (int)(4.0 * p.K * (double)p.H * p.W);
    a.w_bytes = (int)(4 * bf3_packed_floats(p.K, p.M));
    const int n_wg = a.m_tiles * a.tiles_x * a.tiles_y;
    if (p.x_split) {
        const size_t bytes = bf3_split_bytes(p.K, p.H, p.W);
        if (bytes >= 2147483648ull) {
            set_error("bf3_launch: the split operand of a %d x %d x %d blob exceeds 2 GiB", p.K, p.H, p.W);
            return STX_ERR_UNSUPPORTED;
        }
        a.x = static_cast<const float *>(p.x_split);
        a.x_bytes = (int)bytes;
        a.vp_rows = bf3_split_rows(p.H);
        a.vp_tp = bf3_split_tp(p.W);
        switch (p.epilogue) {
            case kEpiForward: return bf3_launch_epi<kEpiForward, true>(s, a, n_wg);
            case kEpiDgrad: return bf3_launch_epi<kEpiDgrad, true>(s, a, n_wg);
        }
        return STX_ERR_UNSUPPORTED;
    }
    switch (p.epilogue) {
        case kEpiForward: return bf3_launch_epi<kEpiForward, false>(s, a, n_wg);
        case kEpiDgrad: return bf3_launch_epi<kEpiDgrad, false>(s, a, n_wg);
    }
    return STX_ERR_UNSUPPORTED;
}

}  // namespace stx
#endif  // STX_EXPERIMENT_BF3
