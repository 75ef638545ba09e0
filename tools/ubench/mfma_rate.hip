// Microbenchmark: fp32 MFMA issue rate as a function of waves per SIMD and accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu, int lds) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256, lds>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    k<NACC><<<grid, 256, lds>>>(out, iters, 1.000001f, 0.999999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("acc %d  wgs/CU %d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, wgs_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 5, 8}) run<2>(w, 0);
    for (int w : {1, 2, 5}) run<4>(w, 0);
    for (int w : {1, 2}) run<8>(w, 0);
    return 0;
}
