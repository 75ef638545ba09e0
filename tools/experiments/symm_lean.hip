// Experiment (not built into libstx.so): S = D F with |S| partial sums, as csrc/symm.hip computes it,
// in a form that fits beside a convolution workgroup on a CU -- at most 48 registers per lane and
// 24 KB of LDS -- so that it can run in the convolutions' shadow on a second stream (DESIGN.md
// section 7, tools/ubench/shadow.hip).
//
//   * v_mfma_f32_16x16x32_bf16 (4 accumulator registers per 16 x 16 block); a wave owns 64 channels
//     x 16 pixels (four blocks), a workgroup of four waves 64 x 64;
//   * D (pre-split pieces [3][Cp][Cp], as gram_finish_kernel writes them) comes chunk by chunk --
//     64 rows x 32 k x 3 pieces = 12 KB -- straight into LDS (`buffer_load ... lds`: no registers),
//     double buffered, XOR-swizzled by the source address so that the fragment reads are conflict free;
//   * F goes global -> registers (eight dword loads per lane: one pixel, eight channels), is split
//     there, and is NOT prefetched: the kernel is latency-bound by design -- beside a convolution
//     its stalls cost nothing.
// Same three-piece arithmetic as symm.hip (mfma_split6 order), other block shape: results agree to
// rounding, not bit for bit.

#include "bf16x3.h"
#include "common.h"

namespace stx {

namespace lean {
constexpr int kM = 64, kN = 64, kK = 32;
constexpr int kChunk = 3 * kM * 64;            // bytes: [piece][row][32 k bf16]
typedef float f32x4a __attribute__((ext_vector_type(4)));
}  // namespace lean

#ifndef STX_LEAN_PREFETCH
#define STX_LEAN_PREFETCH 0   // 1: the F fragment of the next step is requested before this step's MFMAs
#endif                       // (8 more registers: no longer fits beside a convolution workgroup)

#if STX_LEAN_PREFETCH
__global__ __launch_bounds__(256) void symm_lean_kernel(
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void symm_lean_kernel(
#endif
    const float *__restrict__ F, const unsigned short *__restrict__ Dp, float *__restrict__ S,
    float *__restrict__ partials, int C, int Cp, int HW, unsigned f_bytes, int m_tiles) {
    using namespace lean;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kChunk];
    __shared__ float wave_sum[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int mt = __builtin_amdgcn_readfirstlane(L % m_tiles), pt = __builtin_amdgcn_readfirstlane(L / m_tiles);
    const int m0 = mt * kM;
    const int px = pt * kN + wave * 16 + l15;

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(S, 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(Dp), 0, (unsigned)(3u * Cp * Cp * 2u), 0x00020000);
    const unsigned HW4 = (unsigned)HW * 4u;
    const unsigned fvoff = px < HW ? (unsigned)((g * 8) * HW + px) * 4u : kOob;
    const int n_chunks = Cp / kK;
    const unsigned a_off = (unsigned)(l15 * 64 + ((g ^ ((l15 >> 2) & 3)) * 16));   // (+ 1024 per row block)

    // D chunk: piece q of this wave's share = (piece, rows 16 r4 .. + 15): lane i moves row 16 r4 +
    // (i >> 2), segment (i & 3) ^ swizzle into LDS position i of that 1 KB
    // (the swizzle depends on the lane only: (row >> 2) & 3 = (lane >> 4) & 3 in every row quarter)
    const unsigned d_vo = (unsigned)(((lane >> 2) * Cp + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);
    auto d_dma = [&](int chunk, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int q = wave * 3 + n;                    // 0 .. 11: (piece, row quarter)
            const int pc = q >> 2, r4 = q & 3;
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(
                ((pc * Cp + m0 + r4 * 16) * Cp + chunk * kK) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rd, (__attribute__((address_space(3))) void *)(lds + buf * kChunk + pc * (kM * 64) + r4 * 1024),
                16, d_vo, so, 0, 0);
        }
    };

    f32x4a acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4a{0.f, 0.f, 0.f, 0.f};

    d_dma(0, 0);
    float raw[8];
    auto f_load = [&](int chunk) __attribute__((always_inline)) {
        // F fragment of a step: channels 32 chunk + 8 g .. + 7 of pixel px
        const int k0 = __builtin_amdgcn_readfirstlane(chunk * kK);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            raw[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rf, fvoff, (unsigned)min(k0 + e, C) * HW4, 0));
    };
    if (STX_LEAN_PREFETCH) f_load(0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int buf = chunk & 1;
        if (!STX_LEAN_PREFETCH) f_load(chunk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the D chunk and the fragment)
        __syncthreads();                                    // chunk `chunk` is in LDS; the other buffer is free
        if (chunk + 1 < n_chunks) d_dma(chunk + 1, buf ^ 1);
        bf16x8 pb[3];
        split3_bf16(raw, pb[0], pb[1], pb[2]);
        if (STX_LEAN_PREFETCH && chunk + 1 < n_chunks) f_load(chunk + 1);
        const unsigned char *base = lds + buf * kChunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = a_off + (unsigned)i * 1024u;
            bf16x8 pa[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) pa[n] = *reinterpret_cast<const bf16x8 *>(base + n * (kM * 64) + off);
            f32x4a a = acc[i];
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[1], pb[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0], pb[2], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[2], pb[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0], pb[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[1], pb[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0], pb[0], a, 0, 0, 0);
            acc[i] = a;
            __builtin_amdgcn_sched_barrier(0);      // (one block's fragments at a time: 48 registers)
        }
    }

    // ---- S out, |S| summed: register r of block i is row 16 i + 4 g + r, column = the lane's pixel
    float asum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + i * 16 + 4 * g + r;
            const bool ok = row < C && px < HW;
            const float v = acc[i][r];
            asum += ok ? fabsf(v) : 0.f;
            const unsigned vo = ok ? (unsigned)((4 * g + r) * HW + px) * 4u : kOob;
            const int row0 = min(__builtin_amdgcn_readfirstlane(m0 + i * 16), C);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, (unsigned)row0 * HW4, 0);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) asum += __shfl_down(asum, off, 64);
    if (lane == 0) wave_sum[wave] = asum;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
}

int symm_lean_num_workgroups(int C, int HW) { return ceil_div(C, lean::kM) * ceil_div(HW, lean::kN); }

int symm_lean_launch(hipStream_t s, const float *feat, const unsigned short *pieces, float *out,
                     float *partials, int C, int HW) {
    const int Cp = ceil_div(C, lean::kM) * lean::kM;
    symm_lean_kernel<<<symm_lean_num_workgroups(C, HW), 256, 0, s>>>(
        feat, pieces, out, partials, C, Cp, HW, (unsigned)(4.0 * C * (double)HW), Cp / lean::kM);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
