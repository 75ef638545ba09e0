// How close to the bf16 matrix pipe's rate does a compiler-scheduled three-piece split loop get?
// (De-risking a direct bf16x3 convolution for the 64-channel layers: its main loop per k-step of
// 16 channels and tap would be -- per wave -- two B fragments read from an fp32 [pixel][channel]
// patch in LDS (2 x ds_read_b128 each) and split in registers (44 vector instructions each), two
// A blocks x three pre-split pieces read from LDS (6 x ds_read_b128), and 24 MFMAs.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I style_transfer_amd/csrc tools/ubench/bf3_loop.hip -o build_ubench/bf3_loop
// Prints cycles per k-step for: MFMAs only / + LDS reads / + split (the full step), one and two
// waves per SIMD.  768 cycles per step is the matrix pipe's floor for one wave per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bf16x3.h"

using namespace stx;

typedef float f32x4q __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const float *src, float *out, long long *cycles, int steps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // patch: 6 x 66 pixels x (64 + 4) floats; weights: 2 taps x 3 pieces x 64 rows x (16 k bf16 = 8 floats + pad)
    constexpr int CH = 68, PX = 6 * 66;
    float *patch = lds;
    float *wts = lds + PX * CH;
    for (int i = threadIdx.x; i < PX * CH + 2 * 3 * 64 * 12; i += blockDim.x) lds[i] = src[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, g = lane >> 5;
    f32x16b acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        const int tap = s % 9, ks = (s / 9) & 3;
        const int ky = tap / 3, kx = tap % 3;
        bf16x8 pb[2][3], pa[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // pixel (row wave % 4 + ky, column j * 32 + l31 + kx), channels ks * 16 + g * 8 .. + 7
            const float *q = patch + (((wave & 3) + ky) * 66 + j * 32 + l31 + kx) * CH + ks * 16 + g * 8;
            if (MODE >= 1) {
                const f32x4q lo = *reinterpret_cast<const f32x4q *>(q);
                const f32x4q hi = *reinterpret_cast<const f32x4q *>(q + 4);
                const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (MODE >= 2) {
                    split3_bf16(x, pb[j][0], pb[j][1], pb[j][2]);
                } else {
                    pb[j][0] = __builtin_bit_cast(bf16x8, lo);
                    pb[j][1] = __builtin_bit_cast(bf16x8, hi);
                    pb[j][2] = pb[j][0];
                }
            } else {
                pb[j][0] = pb[j][1] = pb[j][2] = __builtin_bit_cast(bf16x8, f32x4q{(float)s, 1.f, 2.f, (float)lane});
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                if (MODE >= 1)
                    pa[i][n] = *reinterpret_cast<const bf16x8 *>(wts + (((tap & 1) * 3 + n) * 64 + i * 32 + l31) * 12 + g * 4);
                else
                    pa[i][n] = __builtin_bit_cast(bf16x8, f32x4q{(float)s, 3.f, (float)n, (float)lane});
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_split6(pa[i], pb[j], acc[i][j]);
    }
    const long long t1 = clock64();
    float v = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) v += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    // the block's span: first start to last finish (two waves of a SIMD do not finish together)
    if (blockIdx.x == 0 && lane == 0) {
        atomicMin(reinterpret_cast<unsigned long long *>(cycles + 1), (unsigned long long)t0);
        atomicMax(reinterpret_cast<unsigned long long *>(cycles + 2), (unsigned long long)t1);
    }
}

template <int MODE>
static void run(int threads, const char *label) {
    float *src, *out;
    long long *cyc, h3[3] = {0, 0, 0};
    hipMalloc(&src, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 24);
    hipMemset(src, 0, 4096 * 4);
    const int steps = 9 * 4 * 8;
    const size_t lds = (6 * 66 * 68 + 2 * 3 * 64 * 12) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        const long long init[3] = {0, 0x7fffffffffffffffLL, 0};
        hipMemcpy(cyc, init, 24, hipMemcpyHostToDevice);
        loop_kernel<MODE><<<256, threads, lds>>>(src, out, cyc, steps);
        hipDeviceSynchronize();
    }
    hipMemcpy(h3, cyc, 24, hipMemcpyDeviceToHost);
    const long long h = h3[2] - h3[1];
    printf("%-34s %d waves/SIMD: %7.1f cycles per step and SIMD (24 MFMAs = 768 per wave)\n", label,
           threads / 256, (double)h / steps / (threads / 256));
}

int main() {
    run<0>(256, "MFMAs only");
    run<1>(256, "+ LDS fragment reads");
    run<2>(256, "+ three-piece split (full step)");
    run<0>(512, "MFMAs only");
    run<1>(512, "+ LDS fragment reads");
    run<2>(512, "+ three-piece split (full step)");
    return 0;
}
