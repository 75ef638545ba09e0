// First layer of the network (3 -> 64 channels, 3 x 3, pad 1; conv1_1 of VGG-16 / VGG-19) with the
// Gram partials of its own output.
//
// The layer is all output: 268 MB written for 12.6 MB read on a 1024 x 1024 tile, 3.8 GFLOP -- and
// the first thing anybody does with the blob is read all of it again for its Gram matrix (the
// shallowest style tap, style_transfer.py:584-590).  This kernel keeps a tile of the blob on chip
// for that: a workgroup walks 128-pixel row segments; per segment
//   * the 3 x 3 x 130 input patch sits in LDS (next segment's patch is in flight in registers);
//   * wave w computes pixels 32 w .. 32 w + 31 for all 64 channels: D[32][64] = X^T[32][36] W^T[36][64]
//     (pixels as MFMA rows: a lane then holds four CONSECUTIVE pixels of one channel per register
//     quad, 16-byte LDS writes and a per-lane bias) as 2 x 18 v_mfma_f32_32x32x2_f32 with the filter
//     bank resident in registers, + bias, ReLU --
//     k-step (q, t) pairs input planes 2q and 2q + 1 at tap t (plane 3 is zero), the order
//     conv_mfma_kernel's first-layer configuration adds them in: the blob is BIT-IDENTICAL to that
//     kernel's, so nothing downstream (ReLU / pooling decisions, L-BFGS trajectories) moves;
//   * the 64 x 128 result goes to LDS as [channel][pixel]; from there it is stored with 16-byte
//     row segments, and -- GRAM -- read back as the bf16 MFMA's fragments (8 consecutive pixels of a
//     channel per lane), split into three bf16 pieces and multiplied into the workgroup's running
//     64 x 64 Gram tile exactly as gram_partial_bf3_kernel does (bf16x3.h: six exact products per
//     pair, fp32-class accuracy), wave w taking pixels 32 w .. 32 w + 31 as two 16-pixel steps.
// At the end the four waves' tiles are added in wave order and written as ONE partial tile per
// workgroup, in the layout gram_finish_kernel reads (splits = workgroups, one tile).  Segments are
// dealt out statically (workgroup b takes a contiguous run of them): every sum has a fixed order.
//
// Replaces, for this shape, conv_mfma_kernel's first-layer configuration (84 us at 3.2 TB/s of
// output on a 1024^2 tile) and gram_partial_bf3_kernel's pass over the blob (56 us, 268 MB read).
// Reference: Caffe Convolution + in-place ReLU of conv1_1 (vgg19.prototxt), num_utils.py:143-147.

#include <algorithm>
#include <cstdlib>

#include "bf16x3.h"
#include "common.h"

namespace stx {

namespace {

constexpr int kFP = 128;             // pixels per segment
constexpr int kFM = 64;              // output channels
constexpr int kPW = kFP + 4;         // patch row in LDS: [x0 - 1 .. x0 + 128] + pad
constexpr int kPatch = 12 * kPW;     // 4 planes (the fourth stays zero) x 3 rows
constexpr int kFLd = kFP + 4;        // row of the output tile in LDS (floats)
constexpr int kTile = kFM * kFLd;
constexpr int kNT = 256;
constexpr int kPL = (9 * (kFP + 2) + kNT - 1) / kNT;     // patch loads per thread (5)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));

// LDS traffic of this wave complete, then the workgroup barrier.  NOT __syncthreads(): that also
// waits for the wave's global STORES to be acknowledged -- here a whole segment of the blob, twice
// per segment: 5.6 us per segment instead of ~2 (the first version: 93 us per 1024^2 plane).
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FirstArgs {
    const float *x;          // [K <= 3][H][W]
    const float *w;          // Caffe bank [64][K][3][3]
    const float *bias;       // [64] or null
    float *y;                // [64][H][W]
    float *gram;             // GRAM: one 64 x 64 partial tile per workgroup
    unsigned *y_amax;        // (or null) max |y| as float bits into kAmaxSlots words: ConvProblem::y_amax
    int K, H, W, tiles_x, n_tiles, relu, vec_store, strided;
};

}  // namespace

// STORE: how the blob leaves -- 0: 16-byte buffer stores (row length a multiple of 4, the common
// case), 1: dword buffer stores (odd row lengths), 2: plain pointers behind bounds tests (blobs of
// 4 GiB and more, beyond a buffer descriptor's reach).  The buffer forms issue the SAME number of
// stores on every path (a lane outside the plane carries an out-of-range offset and is dropped by
// the hardware): the compiler then knows how many younger memory operations stand between a patch
// load and its use, and waits for the load, not for the stores.
template <bool GRAM, int STORE>
__global__ __launch_bounds__(kNT, GRAM ? 2 : 3) void conv_first_kernel(FirstArgs a) {
    // [patch 0 | patch 1 | output tile]; after the last segment the workgroup's Gram tile
    // (64 x 64 floats) is put together at its start
    constexpr int kWork = 2 * kPatch + kTile;
    static_assert(kFM * kFM <= kWork, "the Gram tile must fit the working set");
    __shared__ __attribute__((aligned(16))) float lds[kWork];
    float *const patch = lds;
    float *const tile = lds + 2 * kPatch;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)a.H * a.W;

    // ---- the filter bank as MFMA B operands (columns = channels).  k-step s = 9 q + t: lane half h
    // supplies input plane c = 2 q + h at tap t = 3 ky + kx (conv_mfma_kernel's order: planes in
    // pairs, tap by tap -- the same products summed in the same order, whichever operand is
    // which); planes past K are zero.  aw[mb][s] = W[mb * 32 + l31][c][t].
    float aw[2][18];
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int c = 2 * (s / 9) + half, tap = s % 9;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            aw[mb][s] = c < a.K ? a.w[((size_t)(mb * 32 + l31) * a.K + c) * 9 + tap] : 0.f;
    }
    // the bank has landed before the segment loop starts: otherwise the compiler, which must assume
    // these loads may still be in flight on the loop's first trip, waits for ALL memory traffic
    // (vmcnt(0)) in front of every segment's first MFMA -- including the patch loads just issued
#pragma unroll
    for (int s = 0; s < 18; ++s) asm volatile("" ::"v"(aw[0][s]), "v"(aw[1][s]));
    // (the lane's accumulator COLUMN is channel mb * 32 + l31: one bias value per block)
    float bias_r[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) bias_r[mb] = a.bias ? a.bias[mb * 32 + l31] : 0.f;
    asm volatile("" ::"v"(bias_r[0]), "v"(bias_r[1]));

    // ---- patch staging: element e of the 9 x 130 patch = (plane c * 3 + row r, column col).
    // Buffer loads: what lies outside the picture carries an out-of-range offset and reads as zero
    // -- no select behind the load, so nothing waits for it before the segment's matrix work (with a
    // conditional load the compiler put `s_waitcnt vmcnt(0)` in front of the first MFMA: a full
    // memory round trip per segment, 93 us per 1024^2 plane).
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((size_t)a.K * HW * 4), 0x00020000);
    constexpr unsigned kOob = 0x80000000u;
    // per-thread constants of its (up to) five patch elements: row - 1, column - 1 and the offset
    // of (plane, row - 1, column - 1) relative to the segment's first pixel; an element past the
    // patch (or of a plane past K) gets a row that is never inside the picture
    int p_row[kPL], p_col[kPL], p_off[kPL];
#pragma unroll
    for (int n = 0; n < kPL; ++n) {
        const int e = tid + n * kNT;
        const int pr = e / (kFP + 2), col = e - pr * (kFP + 2);
        const int c = pr / 3, r = pr - 3 * c;
        const bool live = e < 9 * (kFP + 2) && c < a.K;
        p_row[n] = live ? r - 1 : -(1 << 28);
        p_col[n] = col - 1;
        p_off[n] = (c * a.H + r - 1) * a.W + col - 1;
    }
    auto patch_load = [&](int t, float (&preg)[kPL]) {
        // (t < 0: past the last segment -- every row outside the picture)
        const int ty = sgpr(t < 0 ? -4 : t / a.tiles_x);
        const int x0 = sgpr(t < 0 ? 0 : (t - ty * a.tiles_x) * kFP);
        const int origin = sgpr(ty * a.W + x0);
#pragma unroll
        for (int n = 0; n < kPL; ++n) {
            const bool ok = (unsigned)(ty + p_row[n]) < (unsigned)a.H && (unsigned)(x0 + p_col[n]) < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(origin + p_off[n]) * 4u : kOob;
            preg[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
        }
    };
    int p_dst[kPL];      // where the element goes in a patch buffer (-1: not part of the patch)
#pragma unroll
    for (int n = 0; n < kPL; ++n) {
        const int e = tid + n * kNT;
        const int pr = e / (kFP + 2), col = e - pr * (kFP + 2);
        p_dst[n] = e < 9 * (kFP + 2) ? pr * kPW + col : -1;
    }
    auto patch_store = [&](int buf, const float (&preg)[kPL]) {
#pragma unroll
        for (int n = 0; n < kPL; ++n)
            if (p_dst[n] >= 0) patch[buf * kPatch + p_dst[n]] = preg[n];
    };

    f32x16 g[3];       // Gram blocks (0,0), (1,0), (1,1) of this wave's pixels
    if (GRAM) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[b][r] = 0.f;
    }

    // One segment: the patch of segment t sits in LDS buffer `buf`; `pin` holds the patch of the
    // workgroup's next segment (loaded one segment ago), `pout` receives the one after that.  The
    // loads are TWO segments ahead of their use so that waiting for them (the memory counter is in
    // issue order) never means waiting for stores younger than one whole segment -- with the loads
    // one segment ahead every segment drained its own stores before it ended (93 us per plane).
    // Segments are dealt out in CONTIGUOUS runs (STX_FIRST_STRIDED=1, read at launch: one by one
    // across the workgroups instead): a workgroup then writes every channel row of the blob as a run
    // of consecutive 512-byte pieces instead of one piece here and one 256 workgroups further on.
    const int per_wg = a.strided ? 1 : (a.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int grid = a.strided ? (int)gridDim.x : 1;                        // distance to the next segment
    const int t_begin = a.strided ? (int)blockIdx.x : (int)blockIdx.x * per_wg;
    const int t_end = a.strided ? a.n_tiles : (t_begin + per_wg < a.n_tiles ? t_begin + per_wg : a.n_tiles);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.y, 0, STORE == 2 ? 0 : (int)(unsigned)((size_t)kFM * HW * 4), 0x00020000);
    constexpr unsigned kOobY = 0xfffffff0u;
    const int y_ch0 = tid >> 5, y_c4 = (tid & 31) * 4;
    const unsigned y_voff = (unsigned)(((size_t)y_ch0 * HW + y_c4) * 4);
    const bool track = a.y_amax != nullptr;      // uniform
    float amax = 0.f;
    auto segment = [&](int t, int buf, const float (&pin)[kPL], float (&pout)[kPL]) {
        // (past the last segment: every offset out of range, the loads return zeros nobody uses --
        // unconditional, like the stores, so that the memory counter's arithmetic is exact)
        patch_load(t + 2 * grid < t_end ? t + 2 * grid : -1, pout);
        const int ty = sgpr(t / a.tiles_x), x0 = sgpr((t - ty * a.tiles_x) * kFP);

        // ---- convolution: 2 channel blocks x 18 k-steps on this wave's 32 pixels
        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
        // (plane 2 q + half; the fourth plane of the patch array is zero and so are its filters)
        const float *pb = patch + buf * kPatch + half * (3 * kPW) + wave * 32 + l31;
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int q = s / 9, ky = (s % 9) / 3, kx = s % 3;
            const float xv = pb[(2 * q * 3 + ky) * kPW + kx];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, aw[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, aw[1][s], acc[1], 0, 0, 0);
        }
        // ---- bias, ReLU: the lane's column is channel mb * 32 + l31, register quad j its pixels
        // 8 j + 4 half .. + 3 of the wave's 32.  Pixels past the end of the row are ZERO in the tile
        // (they take part in the Gram sums and must not: relu(bias) is not zero) -- only the last
        // segment of a row can have any (a uniform branch).
        const bool ragged = x0 + kFP > a.W;
        const int room = a.W - x0 - wave * 32 - 4 * half;      // pixels of this lane's quads inside
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {acc[mb][4 * j], acc[mb][4 * j + 1], acc[mb][4 * j + 2], acc[mb][4 * j + 3]};
                v += bias_r[mb];
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
                if (ragged) {
                    v.x = 8 * j + 0 < room ? v.x : 0.f, v.y = 8 * j + 1 < room ? v.y : 0.f;
                    v.z = 8 * j + 2 < room ? v.z : 0.f, v.w = 8 * j + 3 < room ? v.w : 0.f;
                }
                if (track) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                *reinterpret_cast<f32x4 *>(tile + (mb * 32 + l31) * kFLd + wave * 32 + 8 * j + 4 * half) = v;
            }
        lds_barrier();      // the tile is complete (and everybody has left the previous segment's reads)

        // ---- the blob: 16-byte row segments out of LDS
        // thread (row y_ch0 + 8 n of the tile, columns y_c4 .. + 3), n = 0 .. 7: one vector offset per
        // thread, the segment's origin and the row as a scalar offset
        {
            float *const yrow = a.y + (size_t)ty * a.W + x0;
            const unsigned origin = (unsigned)sgpr((ty * a.W + x0) * 4);
            const unsigned vo4 = x0 + y_c4 < a.W ? y_voff : kOobY;
#pragma unroll
            for (int n = 0; n < kFM * (kFP / 4) / kNT; ++n) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + (y_ch0 + 8 * n) * kFLd + y_c4);
                if (STORE == 2) {
                    float *const dst = yrow + (size_t)(y_ch0 + 8 * n) * HW + y_c4;
                    if (x0 + y_c4 + 0 < a.W) dst[0] = v.x;
                    if (x0 + y_c4 + 1 < a.W) dst[1] = v.y;
                    if (x0 + y_c4 + 2 < a.W) dst[2] = v.z;
                    if (x0 + y_c4 + 3 < a.W) dst[3] = v.w;
                    continue;
                }
                const unsigned so = origin + (unsigned)sgpr((int)((unsigned)(8 * n) * (unsigned)HW * 4u));
                if (STORE == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4f, v), ry, vo4, so, 0);
                } else {
                    const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vs[i]), ry,
                                                              x0 + y_c4 + i < a.W ? y_voff + 4u * i : kOobY, so, 0);
                }
            }
        }
        // ---- Gram: this wave's 32 pixels as two 16-pixel steps
        if (GRAM) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 p[2][3];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const float *q = tile + (mb * 32 + l31) * kFLd + wave * 32 + st * 16 + half * 8;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(q);
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(q + 4);
                    const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    split3_bf16(xv, p[mb][0], p[mb][1], p[mb][2]);
                }
                g[0] = mfma_split6(p[0], p[0], g[0]);
                g[1] = mfma_split6(p[1], p[0], g[1]);
                g[2] = mfma_split6(p[1], p[1], g[2]);
            }
        }
        if (t + grid < t_end) patch_store(buf ^ 1, pin);
        lds_barrier();      // tile reads done; the next patch is in place
    };

    // (the maximum of the blob, for an fp16-split convolution that reads it next: conv_h2.hip)
    // the planes past K of both patch buffers: zero (0 x garbage could be NaN)
    for (int i = tid; i < 2 * kPatch; i += kNT) patch[i] = 0.f;
    lds_barrier();
    float pa[kPL], pb2[kPL];
    const int t0 = t_begin;
    if (t0 < t_end) {
        patch_load(t0, pa);
        patch_store(0, pa);
    }
    if (t0 + grid < t_end) patch_load(t0 + grid, pa);
    lds_barrier();
    for (int t = t0; t < t_end; t += 2 * grid) {
        segment(t, 0, pa, pb2);
        if (t + grid < t_end) segment(t + grid, 1, pb2, pa);
    }

    if (track) {          // one atomic per workgroup
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
        lds_barrier();
        if (lane == 0) lds[wave] = amax;
        lds_barrier();
        if (tid == 0)
            atomicMax(a.y_amax + (blockIdx.x & (kAmaxSlots - 1)),
                      __builtin_bit_cast(unsigned, fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]))));
    }
    if (GRAM) {
        // the four waves' tiles added in wave order -- ((w0 + w1) + w2) + w3 per element, as
        // gram_partial_bf3_kernel adds them -- in four rounds through ONE 16 KB tile (all four side by
        // side would make the kernel's LDS footprint 64 KB and cost it a workgroup per CU); block
        // (0, 1) of the diagonal tile stays zero
        lds_barrier();
        float *const red = lds;
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    float *q0 = red + row * kFM + l31, *q1 = red + (32 + row) * kFM + l31;
                    q0[0] = w == 0 ? g[0][r] : q0[0] + g[0][r];
                    if (w == 0) q0[32] = 0.f;
                    q1[0] = w == 0 ? g[1][r] : q1[0] + g[1][r];
                    q1[32] = w == 0 ? g[2][r] : q1[32] + g[2][r];
                }
            }
            lds_barrier();
        }
        float *out = a.gram + (size_t)blockIdx.x * (kFM * kFM);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int e = tid + kNT * n;
            reinterpret_cast<float4 *>(out)[e] = reinterpret_cast<const float4 *>(red)[e];
        }
    }
}

// The shapes this kernel takes: a 3 x 3 layer from at most three planes into exactly 64 channels.
bool conv_first_usable(int K, int M, int ksize) {
    const char *env = sw_env("STX_CONV_FIRST_FUSED");
    if (env && atoi(env) == 0) return false;
    return ksize == 3 && K >= 1 && K <= 3 && M == kFM;
}

// Workgroups of a launch on an H x W plane = Gram partial tiles it leaves (two per CU, fewer on
// small planes).
int conv_first_workgroups(int H, int W) {
    const long tiles = (long)H * ceil_div(W, kFP);
    return (int)std::min<long>(tiles, 512);
}
// (without the Gram tile's 48 accumulator registers three workgroups fit a CU)
static int conv_first_plain_workgroups(int H, int W) {
    const long tiles = (long)H * ceil_div(W, kFP);
    const char *env = sw_env("STX_FIRST_WGS");
    return (int)std::min<long>(tiles, env ? atoi(env) : 768);
}

// y = [relu](conv(x, w) + bias); gram_partials (or null): conv_first_workgroups(H, W) partial tiles
// of 64 x 64 floats, to be finished with a GramPlan {C 64, HW, splits = that count, tiles 1}.
int conv_first_launch(hipStream_t s, const float *x, const float *w_caffe, const float *bias, float *y,
                      int K, int H, int W, int relu, float *gram_partials, unsigned *y_amax) {
    if (4.0 * K * (double)H * W >= 2147483648.0) {       // (tiles beyond 13 000 x 13 000: the 8192^2 limit
        set_error("conv_first_launch: a %d x %d plane is beyond the buffer-addressing limit", H, W);   // of the
        return STX_ERR_UNSUPPORTED;                       //  other kernels comes first)
    }
    FirstArgs a;
    a.x = x;
    a.w = w_caffe;
    a.bias = bias;
    a.y = y;
    a.gram = gram_partials;
    a.y_amax = y_amax;
    a.K = K;
    a.H = H;
    a.W = W;
    a.tiles_x = ceil_div(W, kFP);
    a.n_tiles = H * a.tiles_x;
    a.relu = relu;
    a.vec_store = W % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    const int grid = gram_partials ? conv_first_workgroups(H, W) : conv_first_plain_workgroups(H, W);
    {
        const char *env = sw_env("STX_FIRST_STRIDED");
        a.strided = env && atoi(env) != 0;
    }
    const int store = 4.0 * kFM * (double)H * W >= 4294967280.0 ? 2 : a.vec_store ? 0 : 1;
#define STX_FIRST(G, S) conv_first_kernel<G, S><<<grid, kNT, 0, s>>>(a)
    if (gram_partials) {
        if (store == 0) STX_FIRST(true, 0);
        else if (store == 1) STX_FIRST(true, 1);
        else STX_FIRST(true, 2);
    } else {
        if (store == 0) STX_FIRST(false, 0);
        else if (store == 1) STX_FIRST(false, 1);
        else STX_FIRST(false, 2);
    }
#undef STX_FIRST
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
