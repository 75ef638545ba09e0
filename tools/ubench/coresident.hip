// Microbenchmark: is it worth running TWO four-wave workgroups per CU instead of ONE eight-wave
// workgroup, so that the prologue / epilogue of one runs under the matrix work of the other?
//
// A synthetic work item with the instruction mix of conv_wino2's:
//   prologue   P buffer loads from a cold region -> LDS, barrier
//   C chunks   32 MFMAs per wave (8 accumulators x 4 k-steps), behind them 12 operand reads
//              (ds_read_b128), W LDS writes, W + 4 buffer loads, one burst of 16 v_pk_add_f32,
//              one barrier
//   epilogue   ~250 vector instructions, 16 + 16 LDS exchange accesses, 16 stores, barrier
// run as
//   mode 8: 512-thread workgroups, 128 KB of LDS, one per CU, two waves per SIMD (as shipped)
//   mode 4: 256-thread workgroups,  64 KB of LDS, two per CU (half the tiles each, so every
//           workgroup copies the whole filter image: W = 8 instead of 4)
// on the same total amount of matrix work.  Prints microseconds per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coresident.hip -o build_ubench/coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void item(float *out, const float *in, unsigned in_bytes,
                                                 int chunks, float a, float b) {
    constexpr int NT = WAVES * 64;
    constexpr int W = WAVES == 8 ? 4 : 8;                 // filter-image vectors per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)in_bytes, 0x00020000);
    // every workgroup streams its own part of a 16 MB window (the real kernel's loads mostly hit
    // the L2 / Infinity Cache: shared filter images, overlapping patches)
    const unsigned base = (unsigned)(((size_t)blockIdx.x * 2654435761u) % (16u << 20)) & ~1023u;
    f32x4 *l4 = reinterpret_cast<f32x4 *>(lds);
    constexpr int kVec = (WAVES == 8 ? 131072 : 65536) / 16;   // 16-byte slots of LDS
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 wreg[W], xreg[4];
    f32x2 t[16];
    // ---- prologue
#pragma unroll
    for (int n = 0; n < W; ++n) wreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u, base + n * NT * 16, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) xreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u + 4, base + (8 + n) * NT * 16, 0);
#pragma unroll
    for (int n = 0; n < W; ++n) l4[(n * NT + tid) % kVec] = __builtin_bit_cast(f32x4, wreg[n]);
#pragma unroll
    for (int n = 0; n < 4; ++n) l4[((W + n) * NT + tid) % kVec] = __builtin_bit_cast(f32x4, xreg[n]);
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = f32x2{a * i, b};
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- chunks
    f32x4 av[2], bv;
    av[0] = l4[tid % kVec], av[1] = l4[(tid + 64) % kVec], bv = l4[(tid + 128) % kVec];
    for (int c = 0; c < chunks; ++c) {
        const unsigned so = base + (unsigned)((c + 1) * 16 * NT * 16);
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            __builtin_amdgcn_sched_barrier(0);
            if (p == 24) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            acc[p & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][p & 3], bv[(p >> 1) & 3], acc[p & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((p & 7) < 3) {                       // operand reads of the next k-step
                const f32x4 v = l4[(tid + (p & 7) * 64 + (p >> 3) * 256 + c * 16) % kVec];
                if ((p & 7) == 0) av[0] = v;
                if ((p & 7) == 1) av[1] = v;
                if ((p & 7) == 2) bv = v;
            }
            if (p < W && p < 8) l4[((p * NT) + tid + 2048) % kVec] = __builtin_bit_cast(f32x4, wreg[p % W]);
            if (p == 9) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t[(q + 5) & 15]) : "v"(t[q]), "v"(__builtin_bit_cast(f32x4, xreg[q & 3]).xy));
            }
            if (p >= 10 && p < 14) l4[((p * NT) + tid + 4096) % kVec] = f32x4{t[p - 10].x, t[p - 9].y, t[p - 8].x, t[p - 7].y};
            if (p >= 14 && p < 14 + W) wreg[(p - 14) % W] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u, so + (p - 14) * NT * 16, 0);
            if (p >= 26 && p < 30) xreg[p - 26] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u + 4, so + (p - 18) * NT * 16, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: pairs, exchange, pairs, ~100 more vector instructions, stores
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x2 o[32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x2 c0 = {acc[i * 4 + 0][2 * q], acc[i * 4 + 0][2 * q + 1]}, c1 = {acc[i * 4 + 1][2 * q], acc[i * 4 + 1][2 * q + 1]};
            f32x2 c2 = {acc[i * 4 + 2][2 * q], acc[i * 4 + 2][2 * q + 1]}, c3 = {acc[i * 4 + 3][2 * q], acc[i * 4 + 3][2 * q + 1]};
            f32x2 u, v, w, x;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(u) : "v"(c0), "v"(c1));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(u), "v"(c2));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(w) : "v"(c1), "v"(c2));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(w), "v"(c3));
            l4[((wave * 16 + i * 8 + q) * 64 + lane) % kVec] = f32x4{v.x, v.y, x.x, x.y};
        }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 p[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) p[x] = l4[(((x % WAVES) * 16 + g * 2) * 64 + lane) % kVec];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x2 p0 = e ? p[0].zw : p[0].xy, p1 = e ? p[1].zw : p[1].xy, p2 = e ? p[2].zw : p[2].xy, p3 = e ? p[3].zw : p[3].xy;
            f32x2 u, v, w, x;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(u) : "v"(p0), "v"(p1));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(u), "v"(p2));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(w) : "v"(p1), "v"(p2));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(w), "v"(p3));
            o[g * 4 + e * 2] = v, o[g * 4 + e * 2 + 1] = x;
        }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        f32x2 v = o[n];
#pragma unroll
        for (int r = 0; r < 3; ++r) {               // bias / ReLU / selects: ~6 more per output pair
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(v.x) : "v"(a));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(v.y) : "v"(b));
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x + v.y), ro, tid * 4u,
                                              base + (unsigned)(n * NT * 4), 0);
    }
}

template <int WAVES> static float run(int items8, int chunks, int reps) {
    constexpr int NT = WAVES * 64;
    const unsigned bytes = 1u << 30;
    float *in, *out;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMemset(in, 0, bytes);
    auto kern = item<WAVES>;
    const int lds = WAVES == 8 ? 131072 : 65536;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = WAVES == 8 ? items8 : 2 * items8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, out, in, bytes, chunks, 0.f, 0.f);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, out, in, bytes, chunks, 0.f, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(in), hipFree(out);
    return ms / reps * 1e3f;
}

int main() {
    // (work items of the eight-wave form, chunks): the 64-, 128-, 256-, 512-channel layers of a 1024^2 tile
    const int cases[][2] = {{4096, 8}, {2048, 16}, {1024, 32}, {512, 64}};
    for (auto &cs : cases) {
        const float us8 = run<8>(cs[0], cs[1], 10), us4 = run<4>(cs[0], cs[1], 10);
        const double mfma_us = (double)cs[0] / 256 * cs[1] * 64 * 64 / 2.1e3;   // 64 MFMAs of 64 cycles per SIMD and chunk at 2.1 GHz
        printf("%5d items x %2d chunks: one 8-wave workgroup per CU %7.1f us, two 4-wave workgroups per CU %7.1f us  (x%.3f); pure matrix work %.1f us\n",
               cs[0], cs[1], us8, us4, us8 / us4, mfma_us);
    }
    return 0;
}
