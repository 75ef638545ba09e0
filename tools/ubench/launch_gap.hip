// Microbenchmark: per-workgroup turnaround on a CU for a 512-thread, 128 KB LDS, ~220 VGPR kernel
// (one workgroup per CU): each workgroup spins for a fixed number of clocks; the launch takes
// rounds x (spin + gap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, long long spin, float a) {
    extern __shared__ float lds[];
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = a * i;
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    lds[threadIdx.x] = s;
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = lds[(threadIdx.x + 1) & 511];
}
int main() {
    float *out; hipMalloc(&out, 8192 * 512 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (long long spin : {24000LL, 48000LL, 96000LL}) {      // 10, 20, 40 us at 2.4 GHz
        for (int blocks : {256, 1024, 4096}) {
            k<<<blocks, 512, 131072>>>(out, spin, 1.f);
            hipEventRecord(e0);
            k<<<blocks, 512, 131072>>>(out, spin, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int rounds = blocks / 256;
            printf("spin %6lld clk (%.1f us)  blocks %5d: %.1f us total, %.2f us per round, gap %.2f us\n", spin,
                   spin / 2400.0, blocks, ms * 1e3, ms * 1e3 / rounds, ms * 1e3 / rounds - spin / 2400.0);
        }
    }
    return 0;
}
