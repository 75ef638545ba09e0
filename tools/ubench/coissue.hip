// Microbenchmark: how fast does a wave issue VALU / LDS-write / buffer-load instructions while its
// SIMD partner (wave w+4 of the same 512-thread workgroup) streams v_mfma_f32_32x32x2_f32?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ long long g_t[8][2];

template <int MODE, int MFMA_ON, int NOPS = 0>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters, float a, float b) {
    __shared__ f32x4 lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{a * i, b * i};
    f32x4 w4 = {a, b, a, b};
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        if (MFMA_ON)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    if (NOPS >= 1) asm volatile("s_nop 15");
                    if (NOPS >= 2) asm volatile("s_nop 15");
                    if (NOPS >= 3) asm volatile("s_nop 15");
                    if (NOPS >= 4) asm volatile("s_nop 7");
                }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0) {          // 32 independent packed adds
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            } else if (MODE == 1) {   // 32 plain adds
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(v[(i + 1) & 15].y));
            } else if (MODE == 2) {   // 32 LDS b128 writes
#pragma unroll
                for (int u = 0; u < 32; ++u) lds[(u & 7) * 512 + threadIdx.x] = w4;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 3) {   // 32 global b128 loads (L2 hits)
                f32x4 s = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 32; ++u) s += reinterpret_cast<const f32x4 *>(in)[(u * 512 + threadIdx.x) & 8191];
                w4 += s;
            }
        }
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) g_t[wave][0] = t1 - t0;
    float s = w4.x;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x].x;
}

template <int MODE, int MFMA_ON, int NOPS = 0> void run(const char *name) {
    float *out, *in;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&in, 8192 * 16);
    hipMemset(in, 0, 8192 * 16);
    const int iters = 200;
    k<MODE, MFMA_ON, NOPS><<<256, 512>>>(out, in, iters, 1.f, 1.f);
    k<MODE, MFMA_ON, NOPS><<<256, 512>>>(out, in, iters, 1.000001f, 0.999999f);
    hipDeviceSynchronize();
    long long t[8][2];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
    printf("%-28s mfma %d: MFMA wave %7.1f cycles per 32 MFMAs;  other wave %7.1f cycles per 32 instructions\n", name,
           MFMA_ON, (double)t[0][0] / iters, (double)t[4][0] / iters);
    hipFree(out), hipFree(in);
}
int main() {
    run<0, 1>("v_pk_add_f32");
    run<0, 1, 1>("v_pk_add_f32 + 1 nop16");
    run<0, 1, 2>("v_pk_add_f32 + 2 nop16");
    run<0, 1, 3>("v_pk_add_f32 + 3 nop16");
    run<0, 1, 4>("v_pk_add_f32 + 3.5 nop16");
    run<2, 1, 3>("ds_write_b128 + 3 nop16");
    run<3, 1, 3>("load b128 + 3 nop16");
    run<0, 0>("v_pk_add_f32");
    run<1, 1>("v_add_f32");
    run<1, 0>("v_add_f32");
    run<2, 1>("ds_write_b128");
    run<2, 0>("ds_write_b128");
    run<3, 1>("buffer/global load b128");
    run<3, 0>("buffer/global load b128");
    return 0;
}
