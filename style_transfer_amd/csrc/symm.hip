// S = sym(tril(G - Gs)) F with the partial sums of |S|: the SSYMM of the style gradient
// (num_utils.py:59-66, style_transfer.py:587-593) on the bf16 matrix cores with fp32-class
// accuracy -- three bf16 pieces per operand, six products per step (bf16x3.h).
//
// The product is a tall-skinny GEMM: M = C output channels, N = h*w pixels (up to 2^20),
// K = C.  The fp32-MFMA form of it (the 1x1 path of conv_mfma_kernel) needed 393 us per
// 1024 x 1024 tile for the five style layers: 0.42-0.62 of the fp32 matrix pipe on the deep
// layers, 4.3 TB/s on conv1_1.  Here:
//   * D arrives already split (gram_finish_kernel writes the three bf16 piece matrices next to
//     the fp32 one; symm_split_kernel does it for callers that only have the fp32 matrix, or a
//     channel count that is not a multiple of 64), a 64-row x 32-k chunk of it is staged through
//     LDS for the four waves of a workgroup, double buffered, one barrier per chunk;
//   * F never touches LDS: the B fragment of v_mfma_f32_32x32x16_bf16 is 8 consecutive k
//     (channels) of one pixel per lane (lane l: pixel l & 31, channels 8 (l >> 5) .. + 7 of the
//     16-channel step) -- eight dword loads whose lanes walk the pixel axis, i.e. 128-byte row
//     segments of F = [C][h*w], straight into registers, split there, loads two steps ahead;
//   * a wave owns 64 channels x 64 pixels of S (four 32x32 blocks), a workgroup 64 x 256; the
//     channel tiles of one pixel tile run on one XCD (they read the same columns of F).
// Sums of |S| are added per lane, per wave, per workgroup in a fixed order (deterministic).
//
// Round 5: symm_h2_kernel, the same work split on the fp16 matrix cores with TWO fp16 pieces per
// operand and three products per step (f16x2.h): half the matrix time of the six-product form, two
// vector instructions per element for the split instead of 5.5, and D staged from the fp32 matrix
// itself (8 KB per chunk instead of 12 KB of ready pieces; the split rides in the MFMAs' shadow), so
// gram_finish_kernel writes no piece matrices.  The scales are powers of two taken from the
// operands' own maxima: F's from the slots its producer left (the engine's table), D's from the
// per-block maxima gram_finish_kernel leaves beside its sums of squares.

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bf16x3.h"
#include "common.h"
#include "f16x2.h"

#ifndef STX_SYMM_SKIP
#define STX_SYMM_SKIP 0   // timing experiments (tools/ubench/symm_bench.hip): 1 no MFMAs, 2 no split, 4 no F
#endif                   // loads after the first steps, 8 no S stores.  Wrong results when non-zero.

namespace stx {

namespace {

constexpr int kSM = 64;            // output channels per workgroup
constexpr int kSN = 256;           // pixels per workgroup (64 per wave)
#ifndef STX_SYMM_KSK
#define STX_SYMM_KSK 32
#endif
constexpr int kSK = STX_SYMM_KSK;  // k per LDS chunk (kSK / 16 MFMA steps between two barriers)
constexpr int kSPC = kSK / 16;     // steps per chunk
constexpr int kSegs = kSK / 8;     // 16-byte segments per row and piece
constexpr int kDLoads = 3 * kSM * kSegs / 256;   // 16-byte loads per thread and chunk
constexpr int kRowBytes = 2 * kSK + 16;   // + 16 bytes of padding: conflict-free ds_read_b128
#ifndef STX_SYMM_RING
#define STX_SYMM_RING 3
#endif
constexpr int kRing = STX_SYMM_RING;      // F steps in flight per wave (kRing - 1 ahead of the one multiplied)
constexpr int kPieceBytes = kSM * kRowBytes;
constexpr int kChunkBytes = 3 * kPieceBytes;

typedef unsigned u32x4y __attribute__((ext_vector_type(4)));

}  // namespace

size_t symm_pieces_elems(int C) {
    const size_t cp = (size_t)ceil_div(C, kSM) * kSM;
    return 3 * cp * cp;
}

// BIG: F / S of 2 GiB and more -- the descriptors are moved to the step's 16 channels (loads) and
// to the output row (stores) instead of reaching them through the 32-bit offset.
template <bool BIG>
__global__ __launch_bounds__(256, 2) void symm_bf3_kernel(const float *__restrict__ F,
                                                          const unsigned short *__restrict__ Dp,
                                                          float *__restrict__ S,
                                                          float *__restrict__ partials, int C, int Cp,
                                                          int HW, unsigned f_bytes, int m_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kChunkBytes];
    __shared__ float wave_sum[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    // XCD-aware order (workgroup b runs on XCD b & 7): an XCD takes a contiguous range of work
    // items, channel tile fastest
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int mt = __builtin_amdgcn_readfirstlane(L % m_tiles), pt = __builtin_amdgcn_readfirstlane(L / m_tiles);
    const int m0 = mt * kSM;
    const int px0 = pt * kSN + wave * 64;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, BIG ? 0 : f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(S, 0, BIG ? 0 : f_bytes, 0x00020000);
    // per-lane byte offsets of the two pixel blocks (k group g starts 8 rows further down)
    unsigned voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = px0 + j * 32 + l31;
        voff[j] = px < HW ? (unsigned)((g * 8) * HW + px) * 4u : kOob;
    }
    const unsigned HW4 = (unsigned)HW * 4u;
    const int n_steps = Cp / 16, n_chunks = Cp / kSK;

    // ---- D chunk staging: 3 pieces x 64 rows x 64 bytes = 768 16-byte segments, three per thread
    const unsigned short *dsrc[kDLoads];
    unsigned ddst[kDLoads];
#pragma unroll
    for (int n = 0; n < kDLoads; ++n) {
        const int e = tid + 256 * n;                 // (piece, row, segment)
        const int pc = e / (kSM * kSegs), rem = e - pc * (kSM * kSegs);
        const int row = rem / kSegs, seg = rem - row * kSegs;
        dsrc[n] = Dp + ((size_t)pc * Cp + (m0 + row)) * Cp + seg * 8;
        ddst[n] = (unsigned)(pc * kPieceBytes + row * kRowBytes + seg * 16);
    }
    u32x4y dreg[kDLoads];
    auto d_load = [&](int chunk) {
#pragma unroll
        for (int n = 0; n < kDLoads; ++n) dreg[n] = *reinterpret_cast<const u32x4y *>(dsrc[n] + chunk * kSK);
    };
    auto d_store = [&](int buf) {
#pragma unroll
        for (int n = 0; n < kDLoads; ++n) *reinterpret_cast<u32x4y *>(lds + buf * kChunkBytes + ddst[n]) = dreg[n];
    };

    // ---- F fragments: step s covers channels 16 s .. 16 s + 15
    float raw[kRing][2][8];   // ring of steps in flight (kRing - 1 ahead of the one being multiplied)
    auto f_load = [&](int step, int slot_) {
        // (a scalar offset must not exceed the descriptor's range: channels past C are clamped to
        // C, where every lane is out of range and reads zero)
        const int k0 = __builtin_amdgcn_readfirstlane(step * 16);
        const int left = C - k0 < 16 ? C - k0 : 16;
        const __amdgpu_buffer_rsrc_t rk =
            BIG ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F) + (size_t)k0 * (size_t)HW, 0,
                                                    (left > 0 ? left : 0) * HW * 4, 0x00020000)
                : rf;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[slot_][j][e] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(
                               rk, voff[j], BIG ? (unsigned)e * HW4 : (unsigned)min(k0 + e, C) * HW4, 0));
    };

    f32x16b acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    d_load(0);
#pragma unroll
    for (int r = 0; r < kRing - 1; ++r)
        if (r < n_steps) f_load(r, r);
    d_store(0);
    if (n_chunks > 1) d_load(1);
    __syncthreads();

    const unsigned a_off = (unsigned)(l31 * kRowBytes + g * 16);
    auto do_step = [&](int step, int slot_, int buf, int s_in_chunk) {
        bf16x8 pb[2][3], pa[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (STX_SYMM_SKIP & 2) {
                typedef float f32x4s __attribute__((ext_vector_type(4)));
                const float *x = raw[slot_][j];
                pb[j][0] = __builtin_bit_cast(bf16x8, (f32x4s){x[0], x[1], x[2], x[3]});
                pb[j][1] = __builtin_bit_cast(bf16x8, (f32x4s){x[4], x[5], x[6], x[7]});
                pb[j][2] = pb[j][0];
            } else {
                split3_bf16(raw[slot_][j], pb[j][0], pb[j][1], pb[j][2]);
            }
        }
        const unsigned char *base = lds + buf * kChunkBytes + a_off + s_in_chunk * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                pa[i][n] = *reinterpret_cast<const bf16x8 *>(base + n * kPieceBytes + i * 32 * kRowBytes);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (STX_SYMM_SKIP & 1) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc[i][j][q] += (float)(pa[i][q][0] + pb[j][q][1]);
                    continue;
                }
                acc[i][j] = mfma_split6(pa[i], pb[j], acc[i][j]);
            }
    };

    // three steps per trip so that the ring slot is a compile-time constant; a chunk is kSPC steps
    int step = 0;
    auto advance = [&](int slot_) {
        const int chunk = step / kSPC, s_in = step % kSPC, buf = chunk & 1;
        if (step + kRing - 1 < n_steps && !(STX_SYMM_SKIP & 4)) f_load(step + kRing - 1, (slot_ + kRing - 1) % kRing);
        do_step(step, slot_, buf, s_in);
        if (s_in == kSPC - 1) {
            // the other buffer was last read in the previous chunk, which every wave has left
            if (chunk + 1 < n_chunks) d_store(buf ^ 1);
            if (chunk + 2 < n_chunks) d_load(chunk + 2);
            __syncthreads();
        }
        ++step;
    };
    while (step < n_steps) {
#pragma unroll
        for (int r = 0; r < kRing; ++r)
            if (step < n_steps) advance(r);
    }

    // ---- S tile out, |S| summed: D register r of a block is row (r & 3) + 8 (r >> 2) + 4 g
    float asum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const int row0 = min(__builtin_amdgcn_readfirstlane(m0 + i * 32 + (r & 3) + 8 * (r >> 2)), C);
            const unsigned so = BIG ? 0u : (unsigned)row0 * HW4;
            const int rows_left = C - row0 < 5 ? C - row0 : 5;
            const __amdgpu_buffer_rsrc_t rrow =
                BIG ? __builtin_amdgcn_make_buffer_rsrc(S + (size_t)row0 * (size_t)HW, 0,
                                                        (rows_left > 0 ? rows_left : 0) * HW * 4, 0x00020000)
                    : rs;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = px0 + j * 32 + l31;
                const bool ok = row < C && px < HW;
                const float v = acc[i][j][r];
                asum += ok ? fabsf(v) : 0.f;
                const unsigned vo = ok ? (unsigned)((4 * g) * HW + px) * 4u : kOob;
                if (!(STX_SYMM_SKIP & 8))
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rrow, vo, so, 0);
            }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) asum += __shfl_down(asum, off, 64);
    if (lane == 0) wave_sum[wave] = asum;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
}

// ------------------------------------------------------------------------------------------------
// The fp16 two-piece form (f16x2.h).  D = [C][C] fp32 (symmetric, dense); d_amax = n_damax words
// of float bits whose largest is max |D| (gram_finish_kernel's block maxima); f_amax = kAmaxSlots
// words of float bits bounding |F|.
constexpr int kHChunkBytes = 2 * kPieceBytes;
constexpr int kHDLoads = kSM * kSK / 4 / 256;    // 16-byte loads of fp32 D per thread and chunk

template <bool BIG>
__global__ __launch_bounds__(256, 2) void symm_h2_kernel(const float *__restrict__ F,
                                                         const float *__restrict__ D,
                                                         const unsigned *__restrict__ d_amax, int n_damax,
                                                         const unsigned *__restrict__ f_amax,
                                                         float *__restrict__ S,
                                                         float *__restrict__ partials, int C, int Cp,
                                                         int HW, unsigned f_bytes, int m_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kHChunkBytes];
    __shared__ float wave_sum[4];
    __shared__ unsigned wave_max[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int mt = __builtin_amdgcn_readfirstlane(L % m_tiles), pt = __builtin_amdgcn_readfirstlane(L / m_tiles);
    const int m0 = mt * kSM;
    const int px0 = pt * kSN + wave * 64;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, BIG ? 0 : f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(S, 0, BIG ? 0 : f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(D), 0, (unsigned)(C * C) * 4u, 0x00020000);
    unsigned voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = px0 + j * 32 + l31;
        voff[j] = px < HW ? (unsigned)((g * 8) * HW + px) * 4u : kOob;
    }
    const unsigned HW4 = (unsigned)HW * 4u;
    const int n_steps = Cp / 16, n_chunks = Cp / kSK;

    // ---- D chunk staging: 64 rows x kSK columns of fp32 = kHDLoads 16-byte loads per thread; rows
    // and columns past C read as zero (range check / explicit column test: C is a multiple of 4)
    unsigned dvoff[kHDLoads], ddst[kHDLoads];
    int dcol[kHDLoads];
#pragma unroll
    for (int n = 0; n < kHDLoads; ++n) {
        const int e = tid + 256 * n;                 // (row, 4-column segment)
        const int row = e / (kSK / 4), seg = e - row * (kSK / 4);
        dcol[n] = seg * 4;
        dvoff[n] = m0 + row < C ? (unsigned)((m0 + row) * C + seg * 4) * 4u : kOob;
        ddst[n] = (unsigned)(row * kRowBytes + seg * 8);
    }
    u32x4y dreg[kHDLoads];
    auto d_load = [&](int chunk) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(chunk * kSK * 4);
#pragma unroll
        for (int n = 0; n < kHDLoads; ++n)
            dreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rd, dcol[n] + chunk * kSK < C ? dvoff[n] : kOob, so, 0);
    };

    float raw[kRing][2][8];
    auto f_load = [&](int step, int slot_) {
        const int k0 = __builtin_amdgcn_readfirstlane(step * 16);
        const int left = C - k0 < 16 ? C - k0 : 16;
        const __amdgpu_buffer_rsrc_t rk =
            BIG ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F) + (size_t)k0 * (size_t)HW, 0,
                                                    (left > 0 ? left : 0) * HW * 4, 0x00020000)
                : rf;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[slot_][j][e] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(
                               rk, voff[j], BIG ? (unsigned)e * HW4 : (unsigned)min(k0 + e, C) * HW4, 0));
    };

    f32x16h acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the first loads go out before the scales are known
    d_load(0);
#pragma unroll
    for (int r = 0; r < kRing - 1; ++r)
        if (r < n_steps) f_load(r, r);

    // ---- scales: F's from its producer's slots, D's from the block maxima (workgroup-wide maximum)
    const int ef = h2_scale_exp(amax_of_slots(f_amax, kAmaxSlots));
    unsigned dm = 0;
    for (int i = tid; i < n_damax; i += 256) dm = max(dm, d_amax[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dm = max(dm, (unsigned)__shfl_xor((int)dm, off, 64));
    if (lane == 0) wave_max[wave] = dm;
    __syncthreads();
    dm = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    const int ed = __builtin_amdgcn_readfirstlane(h2_scale_exp(dm));
    const float sf = pow2f(ef), sd = pow2f(ed);

    auto d_store = [&](int buf) {
#pragma unroll
        for (int n = 0; n < kHDLoads; ++n) {
            unsigned h[2], l[2];
            // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
            const unsigned d0 = dreg[n].x, d1 = dreg[n].y, d2 = dreg[n].z, d3 = dreg[n].w;
            split2_f16_quad(__builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1),
                            __builtin_bit_cast(float, d2), __builtin_bit_cast(float, d3), sd, h, l);
            typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
            unsigned char *q = lds + buf * kHChunkBytes + ddst[n];
            *reinterpret_cast<u32x2s *>(q) = (u32x2s){h[0], h[1]};
            *reinterpret_cast<u32x2s *>(q + kPieceBytes) = (u32x2s){l[0], l[1]};
        }
    };
    d_store(0);
    if (n_chunks > 1) d_load(1);
    __syncthreads();

    const unsigned a_off = (unsigned)(l31 * kRowBytes + g * 16);
    auto do_step = [&](int slot_, int buf, int s_in_chunk) {
        f16x8h bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) split2_f16(raw[slot_][j], sf, bh[j], bl[j]);
        const unsigned char *base = lds + buf * kHChunkBytes + a_off + s_in_chunk * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const f16x8h *>(base + i * 32 * kRowBytes);
            al[i] = *reinterpret_cast<const f16x8h *>(base + kPieceBytes + i * 32 * kRowBytes);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_split3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
    };

    int step = 0;
    auto advance = [&](int slot_) {
        const int chunk = step / kSPC, s_in = step % kSPC, buf = chunk & 1;
        if (step + kRing - 1 < n_steps) f_load(step + kRing - 1, (slot_ + kRing - 1) % kRing);
        do_step(slot_, buf, s_in);
        if (s_in == kSPC - 1) {
            if (chunk + 1 < n_chunks) d_store(buf ^ 1);
            if (chunk + 2 < n_chunks) d_load(chunk + 2);
            __syncthreads();
        }
        ++step;
    };
    while (step < n_steps) {
#pragma unroll
        for (int r = 0; r < kRing; ++r)
            if (step < n_steps) advance(r);
    }

    // ---- S tile out (both scales undone: exact), |S| summed as in symm_bf3_kernel
    const float ud = pow2f(-ed), uf = pow2f(-ef);
    float asum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const int row0 = min(__builtin_amdgcn_readfirstlane(m0 + i * 32 + (r & 3) + 8 * (r >> 2)), C);
            const unsigned so = BIG ? 0u : (unsigned)row0 * HW4;
            const int rows_left = C - row0 < 5 ? C - row0 : 5;
            const __amdgpu_buffer_rsrc_t rrow =
                BIG ? __builtin_amdgcn_make_buffer_rsrc(S + (size_t)row0 * (size_t)HW, 0,
                                                        (rows_left > 0 ? rows_left : 0) * HW * 4, 0x00020000)
                    : rs;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = px0 + j * 32 + l31;
                const bool ok = row < C && px < HW;
                const float v = (acc[i][j][r] * ud) * uf;
                asum += ok ? fabsf(v) : 0.f;
                const unsigned vo = ok ? (unsigned)((4 * g) * HW + px) * 4u : kOob;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rrow, vo, so, 0);
            }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) asum += __shfl_down(asum, off, 64);
    if (lane == 0) wave_sum[wave] = asum;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
}

// dsym [C][C] fp32 -> pieces [3][Cp][Cp] bf16 (zero padded to a multiple of 64)
__global__ void symm_split_kernel(const float *__restrict__ dsym, int C, int Cp,
                                  unsigned short *__restrict__ pieces) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cp * Cp) return;
    const int i = idx / Cp, j = idx - i * Cp;
    unsigned short s1 = 0, s2 = 0, s3 = 0;
    if (i < C && j < C) split3_bf16_scalar(dsym[i * C + j], s1, s2, s3);
    pieces[idx] = s1;
    pieces[(size_t)Cp * Cp + idx] = s2;
    pieces[2 * (size_t)Cp * Cp + idx] = s3;
}

int symm_num_workgroups(int C, int HW) { return ceil_div(C, kSM) * ceil_div(HW, kSN); }

bool symm_bf3_usable(const float *feat, const float *out, int C, int HW) {
    const char *env = sw_env("STX_SYMM");
    if (env && !strcmp(env, "fp32")) return false;
    // (16 planes must stay under 2 GiB: the reach of one step's loads in the BIG variant)
    return 64.0 * (double)HW < 2147483648.0 &&
           ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out)) & 3) == 0;
}

int symm_bf3_launch(hipStream_t s, const float *feat, const float *dsym, unsigned short *pieces,
                    bool pieces_ready, float *out, float *partials, int C, int HW) {
    const int Cp = ceil_div(C, kSM) * kSM;
    if (!pieces_ready || Cp != C) {
        symm_split_kernel<<<ceil_div(Cp * Cp, 256), 256, 0, s>>>(dsym, C, Cp, pieces);
        STX_CHECK_LAUNCH();
    }
    const int m_tiles = Cp / kSM;
    const double bytes = 4.0 * C * (double)HW;
    const char *force_big = sw_env("STX_WINO_BIG");
    if (bytes >= 2147483648.0 || (force_big && atoi(force_big) == 1))
        symm_bf3_kernel<true><<<symm_num_workgroups(C, HW), 256, 0, s>>>(feat, pieces, out, partials, C, Cp,
                                                                       HW, 0u, m_tiles);
    else
        symm_bf3_kernel<false><<<symm_num_workgroups(C, HW), 256, 0, s>>>(feat, pieces, out, partials, C, Cp,
                                                                        HW, (unsigned)bytes, m_tiles);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// STX_SYMM=bf3 keeps the three-piece bf16 kernel, =fp32 the fp32-MFMA path (stx_reread_env after a change)
bool symm_h2_usable(const float *feat, const float *out, int C, int HW) {
    const char *env = sw_env("STX_SYMM");
    if (env && (!strcmp(env, "fp32") || !strcmp(env, "bf3"))) return false;
    return C % 4 == 0 && symm_bf3_usable(feat, out, C, HW);
}

int symm_h2_launch(hipStream_t s, const float *feat, const float *dsym, const unsigned *d_amax, int n_damax,
                   const unsigned *f_amax, float *out, float *partials, int C, int HW) {
    const int Cp = ceil_div(C, kSM) * kSM;
    const int m_tiles = Cp / kSM;
    const double bytes = 4.0 * C * (double)HW;
    const char *force_big = sw_env("STX_WINO_BIG");
    if (bytes >= 2147483648.0 || (force_big && atoi(force_big) == 1))
        symm_h2_kernel<true><<<symm_num_workgroups(C, HW), 256, 0, s>>>(feat, dsym, d_amax, n_damax, f_amax, out,
                                                                      partials, C, Cp, HW, 0u, m_tiles);
    else
        symm_h2_kernel<false><<<symm_num_workgroups(C, HW), 256, 0, s>>>(feat, dsym, d_amax, n_damax, f_amax, out,
                                                                       partials, C, Cp, HW, (unsigned)bytes, m_tiles);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
