// Gram matrix G = F F^T / (C*h*w) of a feature map F = [C][h*w] on the fp32 matrix cores.
//
// Replaces gram_matrix / ssyrk (num_utils.py:53-56,143-147; callers style_transfer.py:534,584).
// The reference's SYRK fills only the lower triangle and leaves the upper triangle zero, and
// every consumer (norm2 of the difference, ssymm) reads just that triangle, so only the
// lower-triangular 64x64 tiles are computed.  The reduction dimension is the pixel index
// (up to 2^20 for a 1024x1024 tile at conv1_1) while the output is at most 512x512, so the work is
// split along K over many workgroups; each writes a 64x64 partial and a second kernel adds the
// partials in a fixed order (deterministic, no atomics), applies the 1/(C*h*w) scale, subtracts
// the style target and emits what the rest of the tile path needs:
//   gram  (lower triangle, upper zero)                          -- style_transfer.py:584
//   dsym  = sym(tril(gram - target)), the SSYMM left operand    -- style_transfer.py:587-589
//   sum of squares of tril(gram - target), for the style loss   -- style_transfer.py:591
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): A[i][k] = F[ci][p], B[k][j] = F[cj][p]; both operands
// are channel-major in memory, so the LDS tiles are [64 channels][KP pixels + 1 pad] and the
// 32 lanes of an operand read walk the channel axis with an odd stride (conflict-free).

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bf16x3.h"
#include "common.h"
#include "f16x2.h"

namespace stx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGT = 64;        // tile edge (channels)
constexpr int kGP = 64;        // pixels per LDS stage
constexpr int kGLd = kGP + 1;  // padded LDS row

GramPlan gram_plan(int C, int HW) {
    GramPlan p;
    p.C = C;
    p.HW = HW;
    const int T = ceil_div(C, kGT);
    p.tiles = T * (T + 1) / 2;
    int target = 512, min_stages = 8;
    if (const char *env = sw_env("STX_GRAM_WGS")) target = std::max(1, atoi(env));
    if (const char *env = sw_env("STX_GRAM_MIN_STAGES")) min_stages = std::max(1, atoi(env));
    int splits = std::max(1, target / p.tiles);
    splits = std::min(splits, ceil_div(HW, min_stages * kGP));   // at least eight stages per slice
    splits = std::max(splits, 1);
    p.splits = splits;
    p.parts = 1;
    p.partial_floats = (size_t)p.splits * p.parts * p.tiles * kGT * kGT;
    return p;
}

__device__ __forceinline__ void tile_coords(int tile, int &ti, int &tj) {
    // tile index -> (ti >= tj) in row-major lower-triangular order
    int r = (int)((sqrtf(8.f * tile + 1.f) - 1.f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= tile) ++r;
    while (r * (r + 1) / 2 > tile) --r;
    ti = r;
    tj = tile - r * (r + 1) / 2;
}

template <bool VEC4>
__global__ __launch_bounds__(256, 2) void gram_partial_kernel(const float *__restrict__ F, int C,
                                                              int HW, int tiles, int slice,
                                                              int parts,
                                                              float *__restrict__ partials) {
    __shared__ float At[kGT * kGLd];
    __shared__ float Bt[kGT * kGLd];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    int ti, tj;
    tile_coords(tile, ti, tj);
    const bool diag = ti == tj;
    const int p_begin = split * slice;
    const int p_end = min(HW, p_begin + slice);
    const int bi = wave >> 1, bj = wave & 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // each thread stages 16 floats of the A tile and 16 of the B tile per stage.  VEC4 (h*w a
    // multiple of 4, so every channel row is 16-byte aligned): four float4 per tile, a wave
    // covers 4 channel rows x 256 B; otherwise scalar loads, one channel row per wave.
    float ra[16], rb[16];
    auto load = [&](int p0) {
        if (VEC4) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int e = tid + 256 * n;                 // float4 index in the 64 x 16 tile
                const int ch = e >> 4, px = p0 + (e & 15) * 4;
                const int ca = ti * kGT + ch, cb = tj * kGT + ch;
                const bool okp = px < p_end;                 // p_end is a multiple of 4 here
                const bool oka = okp && ca < C, okb = okp && cb < C && !diag;
                const float4 va = *reinterpret_cast<const float4 *>(F + (oka ? (size_t)ca * HW + px : 0));
                const float4 vb = *reinterpret_cast<const float4 *>(F + (okb ? (size_t)cb * HW + px : 0));
                ra[4 * n + 0] = oka ? va.x : 0.f;
                ra[4 * n + 1] = oka ? va.y : 0.f;
                ra[4 * n + 2] = oka ? va.z : 0.f;
                ra[4 * n + 3] = oka ? va.w : 0.f;
                rb[4 * n + 0] = okb ? vb.x : 0.f;
                rb[4 * n + 1] = okb ? vb.y : 0.f;
                rb[4 * n + 2] = okb ? vb.z : 0.f;
                rb[4 * n + 3] = okb ? vb.w : 0.f;
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = tid + 256 * n;
                const int ch = e >> 6, px = p0 + (e & 63);
                const int ca = ti * kGT + ch, cb = tj * kGT + ch;
                const bool okp = px < p_end;
                const bool oka = okp && ca < C, okb = okp && cb < C && !diag;
                const float va = F[oka ? (size_t)ca * HW + px : 0];
                const float vb = F[okb ? (size_t)cb * HW + px : 0];
                ra[n] = oka ? va : 0.f;
                rb[n] = okb ? vb : 0.f;
            }
        }
    };
    auto store = [&]() {
        if (VEC4) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int e = tid + 256 * n;
                const int ch = e >> 4, px = (e & 15) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    At[ch * kGLd + px + k] = ra[4 * n + k];
                    if (!diag) Bt[ch * kGLd + px + k] = rb[4 * n + k];
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = tid + 256 * n;
                const int ch = e >> 6, px = e & 63;
                At[ch * kGLd + px] = ra[n];
                if (!diag) Bt[ch * kGLd + px] = rb[n];
            }
        }
    };

    const float *ap = At + (bi * 32 + l31) * kGLd + half;
    const float *bp = (diag ? At : Bt) + (bj * 32 + l31) * kGLd + half;

    if (p_begin < p_end) {
        load(p_begin);
        store();
        __syncthreads();
        for (int p0 = p_begin; p0 < p_end; p0 += kGP) {
            const bool more = p0 + kGP < p_end;
            if (more) load(p0 + kGP);
#pragma unroll
            for (int k = 0; k < kGP; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
            __syncthreads();
            if (more) {
                store();
                __syncthreads();
            }
        }
    }
    float *out = partials + ((size_t)split * parts * tiles + tile) * (kGT * kGT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[i * kGT + bj * 32 + l31] = acc[r];
    }
}

// Wide variant (h*w % 4 == 0, 16-byte aligned rows).  The 64-pixel stage is split over the four
// waves instead of the 64x64 output: a wave reads its 16 pixels as two 16-byte groups per
// channel row (lane half h takes pixels 8g + 4h .. +3; MFMA i of a group multiplies pixel 8g + i
// from lanes 0-31 with pixel 8g + 4 + i from lanes 32-63) and owns all four 32x32 blocks, so one
// ds_read_b128 per operand block feeds eight MFMAs (the narrow kernel needs two LDS reads per
// MFMA).  On a diagonal tile A and B are the same rows and the upper block is skipped.  Rows are
// padded to 68 floats: the 16 lanes of a ds_read_b128 group then hit 16 different 16-byte slots.
// The waves' partial tiles are added in wave order through LDS before the tile is written.
constexpr int kGLdW = kGP + 4;
typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void gram_partial_wide_kernel(const float *__restrict__ F, int C,
                                                                   int HW, int tiles, int slice,
                                                                   unsigned f_bytes,
                                                                   float *__restrict__ partials) {
    __shared__ __attribute__((aligned(16))) float At[kGT * kGLdW];
    __shared__ __attribute__((aligned(16))) float Bt[kGT * kGLdW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // XCD-aware order (workgroup b runs on XCD b & 7): every XCD takes a contiguous range of
    // (split, tile) pairs, so the tiles of one pixel slice -- which read the same 64-channel row
    // blocks up to C / 64 times between them -- find those in their own L2 instead of fetching
    // them once per XCD (C = 512: 240 MB fetched for a 33 MB blob before, tools/pmc_layers.py)
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tile = L % tiles, split = L / tiles;
    int ti, tj;
    tile_coords(tile, ti, tj);
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
    const bool diag = ti == tj;
    const int p_begin = split * slice;
    const int p_end = min(HW, p_begin + slice);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: four float4 of the A tile (and of the B tile) per thread and stage through buffer
    // loads; rows past C get an out-of-range offset (-> 0), a stage only changes the scalar offset
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, f_bytes, 0x00020000);
    unsigned aoff[4], boff[4];
    int ldst[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;                 // float4 index in the 64 x 16 tile
        const int ch = e >> 4, px = (e & 15) * 4;
        const int ca = ti * kGT + ch, cb = tj * kGT + ch;
        aoff[n] = ca < C ? (unsigned)(ca * HW + px) * 4u : kOob;
        boff[n] = cb < C && !diag ? (unsigned)(cb * HW + px) * 4u : kOob;
        ldst[n] = ch * kGLdW + px;
    }
    u32x4g ra[4], rb[4];
    auto load = [&](int p0) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(p0 * 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) ra[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, aoff[n], so, 0);
        if (!diag) {
#pragma unroll
            for (int n = 0; n < 4; ++n) rb[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, boff[n], so, 0);
        }
    };
    auto store = [&](int p0) {
        // the last stage of a slice may reach past p_end (a multiple of 4): those pixels belong
        // to the next slice or the next channel row and must not count
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            // pixels left in the slice from this vector's first one (a plane size that is not a
            // multiple of 4 ends inside a vector: the rest of it is the next channel's row)
            const int rem = p_end - (p0 + ((tid + 256 * n) & 15) * 4);
            u32x4g va = ra[n], vb = rb[n];
            if (rem < 4) {
                va.x = rem > 0 ? va.x : 0u, va.y = rem > 1 ? va.y : 0u, va.z = rem > 2 ? va.z : 0u, va.w = 0u;
                vb.x = rem > 0 ? vb.x : 0u, vb.y = rem > 1 ? vb.y : 0u, vb.z = rem > 2 ? vb.z : 0u, vb.w = 0u;
            }
            *reinterpret_cast<u32x4g *>(At + ldst[n]) = va;
            if (!diag) *reinterpret_cast<u32x4g *>(Bt + ldst[n]) = vb;
        }
    };

    const float *ap = At + l31 * kGLdW + wave * 16 + half * 4;
    const float *bp = (diag ? At : Bt) + l31 * kGLdW + wave * 16 + half * 4;

    if (p_begin < p_end) {
        load(p_begin);
        store(p_begin);
        __syncthreads();
        for (int p0 = p_begin; p0 < p_end; p0 += kGP) {
            const bool more = p0 + kGP < p_end;
            if (more) load(p0 + kGP);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x4g a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const f32x4g *>(ap + i * 32 * kGLdW + g * 8);
                    b[i] = *reinterpret_cast<const f32x4g *>(bp + i * 32 * kGLdW + g * 8);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (diag && j > i) continue;      // upper block of a diagonal tile
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k],
                                                                              acc[i][j], 0, 0, 0);
                        }
            }
            __syncthreads();
            if (more) {
                store(p0 + kGP);
                __syncthreads();
            }
        }
    }
    // add the four waves' partial tiles in wave order through LDS (deterministic), then one
    // coalesced write of the 64x64 tile
    __syncthreads();
    float *red = At;                                   // 64 x 64 floats fit in the A stage
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float *q = red + row * kGT + j * 32 + l31;
                        *q = w == 0 ? acc[i][j][r] : *q + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float *out = partials + ((size_t)split * tiles + tile) * (kGT * kGT);
#pragma unroll
    for (int n = 0; n < 4; ++n)
        reinterpret_cast<float4 *>(out)[tid + 256 * n] = reinterpret_cast<const float4 *>(red)[tid + 256 * n];
}


// ------------------------------------------------------------------------------------------------
// Gram on the bf16 matrix cores with fp32-class accuracy: three bf16 pieces per operand, six
// products per step (bf16x3.h has the arithmetic and its error budget).
//
// The A / B fragment of the 32x32x16 MFMA is 8 consecutive pixels of one channel row per lane
// (lane l: channel l & 31, pixels 8 (l >> 5) .. + 7 of the 16-pixel step).  Loading those 32
// bytes per lane straight from F = [C][h*w] was tried first and is SLOWER than the fp32 kernel
// (conv1_1: 115 vs 73 us): a wave load then touches 32 different cache lines.  So the tile is
// staged exactly as in gram_partial_wide_kernel -- coalesced 16-byte loads, fp32 in LDS -- and
// split into bf16 pieces in registers after the fragment read (5.5 vector instructions per
// element, issued in the shadow of the bf16 MFMAs, which -- unlike the fp32 MFMA -- leave the
// vector pipe free).
#ifndef STX_GRAM_SKIP
#define STX_GRAM_SKIP 0   // timing experiments (tools/ubench/gram_bench.hip): 1 no MFMAs, 2 no split (raw
#endif                   // bits as pieces), 4 no loads after the first stage, 8 no LDS stores.  Wrong results.
__global__ __launch_bounds__(256, 2) void gram_partial_bf3_kernel(const float *__restrict__ F, int C,
                                                                  int HW, int tiles, int slice,
                                                                  unsigned f_bytes,
                                                                  float *__restrict__ partials) {
    // two stages of [A tile | B tile] in LDS: stage s + 1 is written while stage s is multiplied,
    // one barrier per stage, and the loads of stage s + 2 are in flight for a whole stage
    constexpr int kStageFloats = 2 * kGT * kGLdW;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // XCD-aware order, as gram_partial_wide_kernel
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tile = L % tiles, split = L / tiles;
    int ti, tj;
    tile_coords(tile, ti, tj);
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
    const bool diag = ti == tj;
    const int p_begin = split * slice;
    const int p_end = min(HW, p_begin + slice);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging exactly as gram_partial_wide_kernel: fp32 tiles [64 channels][64 pixels + 4] in LDS,
    // coalesced 16-byte buffer loads (a wave covers 4 channel rows x 256 bytes)
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, f_bytes, 0x00020000);
    unsigned aoff[4], boff[4];
    int ldst[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;                 // float4 index in the 64 x 16 tile
        const int ch = e >> 4, px = (e & 15) * 4;
        const int ca = ti * kGT + ch, cb = tj * kGT + ch;
        aoff[n] = ca < C ? (unsigned)(ca * HW + px) * 4u : kOob;
        boff[n] = cb < C && !diag ? (unsigned)(cb * HW + px) * 4u : kOob;
        ldst[n] = ch * kGLdW + px;
    }
    u32x4g ra[4], rb[4];
    auto load = [&](int p0) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(p0 * 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) ra[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, aoff[n], so, 0);
        if (!diag) {
#pragma unroll
            for (int n = 0; n < 4; ++n) rb[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, boff[n], so, 0);
        }
    };
    auto store = [&](int p0, int buf) {
        float *At = lds + buf * kStageFloats, *Bt = At + kGT * kGLdW;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rem = p_end - (p0 + ((tid + 256 * n) & 15) * 4);
            u32x4g va = ra[n], vb = rb[n];
            if (rem < 4) {
                va.x = rem > 0 ? va.x : 0u, va.y = rem > 1 ? va.y : 0u, va.z = rem > 2 ? va.z : 0u, va.w = 0u;
                vb.x = rem > 0 ? vb.x : 0u, vb.y = rem > 1 ? vb.y : 0u, vb.z = rem > 2 ? vb.z : 0u, vb.w = 0u;
            }
            *reinterpret_cast<u32x4g *>(At + ldst[n]) = va;
            if (!diag) *reinterpret_cast<u32x4g *>(Bt + ldst[n]) = vb;
        }
    };

    // Fragments: wave w takes pixels 16 w .. 16 w + 15 of the stage as ONE 32x32x16 step; lane half h
    // holds pixels 8 h .. 8 h + 7 of it (two 16-byte LDS reads per 32-channel block), splits them
    // into the three bf16 pieces in registers and feeds all four output blocks of the tile.
    const int a_off = l31 * kGLdW + wave * 16 + half * 8;
    const int b_off = (diag ? 0 : kGT * kGLdW) + a_off;
    auto fragment = [&](const float *q, bf16x8 (&pc)[3]) {
        const f32x4g lo = *reinterpret_cast<const f32x4g *>(q);
        const f32x4g hi = *reinterpret_cast<const f32x4g *>(q + 4);
        const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (STX_GRAM_SKIP & 2) {
            typedef float f32x4s __attribute__((ext_vector_type(4)));
            pc[0] = __builtin_bit_cast(bf16x8, (f32x4s){x[0], x[1], x[2], x[3]});
            pc[1] = __builtin_bit_cast(bf16x8, (f32x4s){x[4], x[5], x[6], x[7]});
            pc[2] = pc[0];
        } else {
            split3_bf16(x, pc[0], pc[1], pc[2]);
        }
    };

    if (p_begin < p_end) {
        load(p_begin);
        store(p_begin, 0);
        if (p_begin + kGP < p_end) load(p_begin + kGP);
        __syncthreads();
        int buf = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += kGP, buf ^= 1) {
            const float *base = lds + buf * kStageFloats;
            bf16x8 pa[2][3], pb[2][3];
            fragment(base + a_off, pa[0]);
            fragment(base + a_off + 32 * kGLdW, pa[1]);
            if (!diag) {
                fragment(base + b_off, pb[0]);
                fragment(base + b_off + 32 * kGLdW, pb[1]);
            }
            // the other buffer was last read in the previous stage, which every wave has left
            if (p0 + kGP < p_end && !(STX_GRAM_SKIP & 8)) store(p0 + kGP, buf ^ 1);
            if (p0 + 2 * kGP < p_end && !(STX_GRAM_SKIP & 4)) load(p0 + 2 * kGP);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (diag && j > i) continue;      // upper block of a diagonal tile
                    if (STX_GRAM_SKIP & 1) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc[i][j][q] += (float)(pa[i][q][0] + (diag ? pa[j] : pb[j])[q][1]);
                        continue;
                    }
                    acc[i][j] = mfma_split6(pa[i], diag ? pa[j] : pb[j], acc[i][j]);
                }
            __syncthreads();
        }
    }
    // the four waves' partial tiles side by side in LDS (4 x 16 KB: the two stages are free now),
    // one barrier, then every thread adds its 16 elements in wave order -- ((w0 + w1) + w2) + w3,
    // the order of the fp32 kernel's four sequential rounds, bit for bit -- and writes them
    float *red = lds + wave * (kGT * kGT);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                red[row * kGT + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    static_assert(4 * kGT * kGT <= 2 * kStageFloats, "the four partial tiles must fit the stage buffers");
    float *out = partials + ((size_t)split * tiles + tile) * (kGT * kGT);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;
        const float4 w0 = reinterpret_cast<const float4 *>(lds)[e];
        const float4 w1 = reinterpret_cast<const float4 *>(lds + kGT * kGT)[e];
        const float4 w2 = reinterpret_cast<const float4 *>(lds + 2 * kGT * kGT)[e];
        const float4 w3 = reinterpret_cast<const float4 *>(lds + 3 * kGT * kGT)[e];
        float4 v;
        v.x = ((w0.x + w1.x) + w2.x) + w3.x;
        v.y = ((w0.y + w1.y) + w2.y) + w3.y;
        v.z = ((w0.z + w1.z) + w2.z) + w3.z;
        v.w = ((w0.w + w1.w) + w2.w) + w3.w;
        reinterpret_cast<float4 *>(out)[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same kernel on the fp16 matrix cores with two-piece operands (f16x2.h; round 5): three products
// per step instead of six, two vector instructions per element for the split instead of 5.5.  The
// scale is the power of two that puts the blob's maximum (f_amax: the kAmaxSlots words its producer
// left, or absmax_launch) into [2^13, 2^14).
__global__ __launch_bounds__(256, 2) void gram_partial_h2_kernel(const float *__restrict__ F, int C,
                                                                  int HW, int tiles, int slice,
                                                                 unsigned f_bytes,
                                                                 const unsigned *__restrict__ f_amax,
                                                                 float *__restrict__ partials) {
    // two stages of [A tile | B tile] in LDS: stage s + 1 is written while stage s is multiplied,
    // one barrier per stage, and the loads of stage s + 2 are in flight for a whole stage
    constexpr int kStageFloats = 2 * kGT * kGLdW;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // XCD-aware order, as gram_partial_wide_kernel
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tile = L % tiles, split = L / tiles;
    int ti, tj;
    tile_coords(tile, ti, tj);
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
    const bool diag = ti == tj;
    const int p_begin = split * slice;
    const int p_end = min(HW, p_begin + slice);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging exactly as gram_partial_wide_kernel: fp32 tiles [64 channels][64 pixels + 4] in LDS,
    // coalesced 16-byte buffer loads (a wave covers 4 channel rows x 256 bytes)
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, f_bytes, 0x00020000);
    unsigned aoff[4], boff[4];
    int ldst[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;                 // float4 index in the 64 x 16 tile
        const int ch = e >> 4, px = (e & 15) * 4;
        const int ca = ti * kGT + ch, cb = tj * kGT + ch;
        aoff[n] = ca < C ? (unsigned)(ca * HW + px) * 4u : kOob;
        boff[n] = cb < C && !diag ? (unsigned)(cb * HW + px) * 4u : kOob;
        ldst[n] = ch * kGLdW + px;
    }
    u32x4g ra[4], rb[4];
    auto load = [&](int p0) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(p0 * 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) ra[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, aoff[n], so, 0);
        if (!diag) {
#pragma unroll
            for (int n = 0; n < 4; ++n) rb[n] = __builtin_amdgcn_raw_buffer_load_b128(rf, boff[n], so, 0);
        }
    };
    auto store = [&](int p0, int buf) {
        float *At = lds + buf * kStageFloats, *Bt = At + kGT * kGLdW;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rem = p_end - (p0 + ((tid + 256 * n) & 15) * 4);
            u32x4g va = ra[n], vb = rb[n];
            if (rem < 4) {
                va.x = rem > 0 ? va.x : 0u, va.y = rem > 1 ? va.y : 0u, va.z = rem > 2 ? va.z : 0u, va.w = 0u;
                vb.x = rem > 0 ? vb.x : 0u, vb.y = rem > 1 ? vb.y : 0u, vb.z = rem > 2 ? vb.z : 0u, vb.w = 0u;
            }
            *reinterpret_cast<u32x4g *>(At + ldst[n]) = va;
            if (!diag) *reinterpret_cast<u32x4g *>(Bt + ldst[n]) = vb;
        }
    };

    // Fragments as in gram_partial_bf3_kernel, split into the two fp16 pieces in registers.
    const int a_off = l31 * kGLdW + wave * 16 + half * 8;
    const int b_off = (diag ? 0 : kGT * kGLdW) + a_off;
    // (both operands carry the scale: the partial tiles are 2^(2 es) times the products, undone by
    // gram_finish_kernel -- exact)
    const float sv = pow2f(h2_scale_exp(amax_of_slots(f_amax, kAmaxSlots)));
    auto fragment = [&](const float *q, f16x8h &hi, f16x8h &lo) {
        const f32x4g v0 = *reinterpret_cast<const f32x4g *>(q);
        const f32x4g v1 = *reinterpret_cast<const f32x4g *>(q + 4);
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        split2_f16(x, sv, hi, lo);
    };

    if (p_begin < p_end) {
        load(p_begin);
        store(p_begin, 0);
        if (p_begin + kGP < p_end) load(p_begin + kGP);
        __syncthreads();
        int buf = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += kGP, buf ^= 1) {
            const float *base = lds + buf * kStageFloats;
            f16x8h ah[2], al[2], bh[2], bl[2];
            fragment(base + a_off, ah[0], al[0]);
            fragment(base + a_off + 32 * kGLdW, ah[1], al[1]);
            if (!diag) {
                fragment(base + b_off, bh[0], bl[0]);
                fragment(base + b_off + 32 * kGLdW, bh[1], bl[1]);
            }
            // the other buffer was last read in the previous stage, which every wave has left
            if (p0 + kGP < p_end) store(p0 + kGP, buf ^ 1);
            if (p0 + 2 * kGP < p_end) load(p0 + 2 * kGP);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (diag && j > i) continue;      // upper block of a diagonal tile
                    acc[i][j] = mfma_split3(ah[i], al[i], diag ? ah[j] : bh[j], diag ? al[j] : bl[j], acc[i][j]);
                }
            __syncthreads();
        }
    }
    // the four waves' partial tiles side by side in LDS (4 x 16 KB: the two stages are free now),
    // one barrier, then every thread adds its 16 elements in wave order -- ((w0 + w1) + w2) + w3,
    // the order of the fp32 kernel's four sequential rounds, bit for bit -- and writes them
    float *red = lds + wave * (kGT * kGT);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                red[row * kGT + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    static_assert(4 * kGT * kGT <= 2 * kStageFloats, "the four partial tiles must fit the stage buffers");
    float *out = partials + ((size_t)split * tiles + tile) * (kGT * kGT);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;
        const float4 w0 = reinterpret_cast<const float4 *>(lds)[e];
        const float4 w1 = reinterpret_cast<const float4 *>(lds + kGT * kGT)[e];
        const float4 w2 = reinterpret_cast<const float4 *>(lds + 2 * kGT * kGT)[e];
        const float4 w3 = reinterpret_cast<const float4 *>(lds + 3 * kGT * kGT)[e];
        float4 v;
        v.x = ((w0.x + w1.x) + w2.x) + w3.x;
        v.y = ((w0.y + w1.y) + w2.y) + w3.y;
        v.z = ((w0.z + w1.z) + w2.z) + w3.z;
        v.w = ((w0.w + w1.w) + w2.w) + w3.w;
        reinterpret_cast<float4 *>(out)[e] = v;
    }
}

// STX_GRAM=fp32 (stx_reread_env after a change) keeps the fp32-MFMA kernels, =bf3 the three-piece bf16 kernel,
// for A/B measurements and tests.
static bool gram_use_bf3() {
    const char *env = sw_env("STX_GRAM");
    return !(env && !strcmp(env, "fp32"));
}

bool gram_h2_usable(const float *feat, int C, int HW) {
    const char *env = sw_env("STX_GRAM");
    if (env && (!strcmp(env, "fp32") || !strcmp(env, "bf3"))) return false;
    return (reinterpret_cast<uintptr_t>(feat) & 3) == 0 && 4.0 * C * (double)HW < 2147483648.0;
}

int gram_partials_launch(hipStream_t s, const float *feat, const GramPlan &plan, float *partials,
                         const unsigned *f_amax) {
    int slice = ceil_div(plan.HW, plan.splits);
    slice = ceil_div(slice, kGP) * kGP;
    const bool aligned = (reinterpret_cast<uintptr_t>(feat) & 15) == 0;
    const double bytes = 4.0 * plan.C * (double)plan.HW;
    if (f_amax) {        // the caller has asked gram_h2_usable
        gram_partial_h2_kernel<<<plan.tiles * plan.splits, 256, 0, s>>>(
            feat, plan.C, plan.HW, plan.tiles, slice, (unsigned)bytes, f_amax, partials);
        STX_CHECK_LAUNCH();
        return STX_OK;
    }
    if ((reinterpret_cast<uintptr_t>(feat) & 3) == 0 && bytes < 2147483648.0 && gram_use_bf3()) {
        gram_partial_bf3_kernel<<<plan.tiles * plan.splits, 256, 0, s>>>(
            feat, plan.C, plan.HW, plan.tiles, slice, (unsigned)bytes, partials);
        STX_CHECK_LAUNCH();
        return STX_OK;
    }
    // (buffer loads need dword alignment only: odd plane sizes take the wide kernel too)
    if ((reinterpret_cast<uintptr_t>(feat) & 3) == 0 && bytes < 2147483648.0 && !sw_env("STX_GRAM_NARROW")) {
        gram_partial_wide_kernel<<<plan.tiles * plan.splits, 256, 0, s>>>(
            feat, plan.C, plan.HW, plan.tiles, slice, (unsigned)bytes, partials);
        STX_CHECK_LAUNCH();
        return STX_OK;
    }
    // narrow kernel: one partial per slice, at stride `parts` (the other slots stay zero)
    if (plan.parts > 1)
        STX_HIP(hipMemsetAsync(partials, 0, plan.partial_floats * sizeof(float), s));
    if (plan.HW % 4 == 0 && aligned)
        gram_partial_kernel<true><<<plan.tiles * plan.splits, 256, 0, s>>>(
            feat, plan.C, plan.HW, plan.tiles, slice, plan.parts, partials);
    else
        gram_partial_kernel<false><<<plan.tiles * plan.splits, 256, 0, s>>>(
            feat, plan.C, plan.HW, plan.tiles, slice, plan.parts, partials);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// 64 consecutive (i, j) elements per workgroup, each summed by four "split lanes" that walk the
// slices with a stride of four (independent partial sums -> memory-level parallelism); the four
// lanes are then added in a fixed order, so the result does not depend on scheduling.
// f_amax (or null): the partial tiles came from gram_partial_h2_kernel, i.e. they are 2^(2 es) times
// the products, es the scale exponent of those slots -- undone here (exact).  block_amax (with
// target): max |G - Gs| of the block as float bits, for symm_h2_kernel's scale (plain stores: no
// memset, no atomics; blocks of the upper triangle leave 0).
__global__ __launch_bounds__(256) void gram_finish_kernel(const float *__restrict__ partials, int C,
                                                          int tiles, int splits, float scale,
                                                          float *__restrict__ gram,
                                                          const float *__restrict__ target,
                                                          float *__restrict__ dsym,
                                                          float *__restrict__ block_sumsq,
                                                          unsigned short *__restrict__ pieces,
                                                          const unsigned *__restrict__ f_amax,
                                                          unsigned *__restrict__ block_amax) {
    __shared__ float lane_sum[4][64];
    __shared__ float red[64];
    __shared__ float redm[64];
    float unscale = 1.f;
    if (f_amax) unscale = pow2f(-h2_scale_exp(amax_of_slots(f_amax, kAmaxSlots)));
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + el;
    const bool valid = idx < C * C;
    // Only the lower-triangle lanes (i >= j) add up partial sums: their 64 consecutive columns are
    // one 256-byte row segment of the partial tile per slice.  The mirrored lanes would read
    // column-wise (64 cache lines per wave load -- measured 27 us for C = 64 where this takes
    // ~10); the lower lane writes both (i, j) and (j, i) of the difference matrix instead.
    int i = 0, j = 0;
    float sum = 0.f;
    if (valid) {
        i = idx / C;
        j = idx % C;
    }
    const bool lower = valid && i >= j;
    if (lower) {
        const int ti = i / kGT, tj = j / kGT;
        const int tile = ti * (ti + 1) / 2 + tj;
        const size_t stride = (size_t)tiles * (kGT * kGT);
        const float *p = partials + (size_t)tile * (kGT * kGT) + (i % kGT) * kGT + (j % kGT);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int s = sl;
        for (; s + 12 < splits; s += 16) {
            s0 += p[(size_t)s * stride];
            s1 += p[(size_t)(s + 4) * stride];
            s2 += p[(size_t)(s + 8) * stride];
            s3 += p[(size_t)(s + 12) * stride];
        }
        for (; s < splits; s += 4) s0 += p[(size_t)s * stride];
        sum = (s0 + s1) + (s2 + s3);
    }
    lane_sum[sl][el] = sum;
    __syncthreads();
    float sq = 0.f, dmax = 0.f;
    if (sl == 0 && valid) {
        if (lower) {
            // (the two powers of two first: exact, and neither can leave the normal range alone)
            const float g = ((((lane_sum[0][el] + lane_sum[1][el]) + (lane_sum[2][el] + lane_sum[3][el])) *
                              unscale) * unscale) * scale;
            if (gram) gram[idx] = g;
            if (target) {
                const float d = g - target[i * C + j];
                dsym[idx] = d;
                if (i != j) dsym[j * C + i] = d;
                if (pieces) {       // the SYMM kernel's left operand, already split (bf16x3.h)
                    unsigned short s1, s2, s3;
                    split3_bf16_scalar(d, s1, s2, s3);
                    const size_t cc = (size_t)C * C;
                    pieces[idx] = s1, pieces[cc + idx] = s2, pieces[2 * cc + idx] = s3;
                    if (i != j) {
                        const int m = j * C + i;
                        pieces[m] = s1, pieces[cc + m] = s2, pieces[2 * cc + m] = s3;
                    }
                }
                sq = d * d;
                dmax = fabsf(d);
            }
        } else if (gram) {
            gram[idx] = 0.f;
        }
    }
    if (block_sumsq) {
        if (sl == 0) red[el] = sq, redm[el] = dmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int k = 0; k < 64; ++k) t += red[k];
            block_sumsq[blockIdx.x] = t;
        }
        if (threadIdx.x == 64 && block_amax) {
            float m = 0.f;
            for (int k = 0; k < 64; ++k) m = fmaxf(m, redm[k]);
            block_amax[blockIdx.x] = __builtin_bit_cast(unsigned, m);
        }
    }
}

__global__ void sum_partials_kernel(const float *__restrict__ partials, int n,
                                    float *__restrict__ out) {
    // single workgroup, fixed summation order: thread-strided sums, then a tree over 256 lanes
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

int sum_partials_launch(hipStream_t s, const float *partials, int n, float *out) {
    sum_partials_kernel<<<1, 256, 0, s>>>(partials, n, out);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// two independent sums in one launch (workgroup 0: a, workgroup 1: b), each in the fixed order of
// sum_partials_kernel: a dispatch costs ~4.5 us however little it does, and a style layer's two
// scalars (sum of squares of G - Gs, sum |S|) are both only needed after the SYMM product
__global__ void sum_partials2_kernel(const float *__restrict__ a, int na, float *__restrict__ out_a,
                                     const float *__restrict__ b, int nb, float *__restrict__ out_b) {
    __shared__ float red[256];
    const float *partials = blockIdx.x ? b : a;
    const int n = blockIdx.x ? nb : na;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) (blockIdx.x ? out_b : out_a)[0] = red[0];
}

int sum_partials2_launch(hipStream_t s, const float *a, int na, float *out_a, const float *b, int nb,
                         float *out_b) {
    sum_partials2_kernel<<<2, 256, 0, s>>>(a, na, out_a, b, nb, out_b);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// up to kSumJobs postponed sums in one launch: workgroup b adds job b's floats exactly as
// sum_partials_kernel / sum_partials2_kernel / sum_two_kernel do (thread-strided sums, then a tree
// over 256 lanes): the same bits, four to five launches fewer per tile evaluation
constexpr int kSumJobs = 16;
struct SumJobTable {
    const float *src[kSumJobs];
    float *dst[kSumJobs];
    int n[kSumJobs];
};
__global__ void sum_jobs_kernel(SumJobTable t) {
    __shared__ float red[256];
    const float *partials = t.src[blockIdx.x];
    const int n = t.n[blockIdx.x];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) t.dst[blockIdx.x][0] = red[0];
}

int sum_jobs_launch(hipStream_t s, const SumJob *jobs, int n_jobs) {
    for (int base = 0; base < n_jobs; base += kSumJobs) {
        SumJobTable t{};
        const int m = std::min(kSumJobs, n_jobs - base);
        for (int i = 0; i < m; ++i) t.src[i] = jobs[base + i].src, t.dst[i] = jobs[base + i].dst, t.n[i] = jobs[base + i].n;
        sum_jobs_kernel<<<m, 256, 0, s>>>(t);
        STX_CHECK_LAUNCH();
    }
    return STX_OK;
}

int gram_finish_blocks(const GramPlan &plan) { return ceil_div(plan.C * plan.C, 64); }

int gram_finish_launch(hipStream_t s, const float *partials, const GramPlan &plan, float *gram_out,
                       const float *target, float *dsym, float *sumsq, unsigned short *pieces,
                       const unsigned *f_amax, float *block_out) {
    const int blocks = gram_finish_blocks(plan);
    // block partial sums, and behind them the block maxima, live behind the Gram partials (the caller
    // sizes the buffer for all three) unless the caller names a place
    float *block_sumsq = !target ? nullptr : block_out ? block_out : const_cast<float *>(partials) + plan.partial_floats;
    unsigned *block_amax = target ? reinterpret_cast<unsigned *>(block_sumsq + blocks) : nullptr;
    const float scale = (float)(1.0 / ((double)plan.C * (double)plan.HW));
    gram_finish_kernel<<<blocks, 256, 0, s>>>(partials, plan.C, plan.tiles, plan.splits * plan.parts, scale,
                                              gram_out, target, dsym, block_sumsq,
                                              target ? pieces : nullptr, f_amax, block_amax);
    STX_CHECK_LAUNCH();
    // sumsq == null: the caller adds the block partials (behind the Gram partials) up itself
    if (target && sumsq) STX_TRY(sum_partials_launch(s, block_sumsq, blocks, sumsq));
    return STX_OK;
}

}  // namespace stx
