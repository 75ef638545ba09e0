// What HBM delivers to plain streaming kernels on this part (ceilings for the first / last layer,
// pooling backward, Gram of conv1_1):  hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_bw.hip -o /tmp/stream_bw
//   write-only (dword / 16-byte stores), read-only (16-byte loads, sum), copy; 268 MB (one 64-channel
//   1024^2 blob) and 1 GB working sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void fill4(f4 *p, size_t n4, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = f4{v, v, v, v};
}
__global__ void fill1(float *p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void read4(const f4 *p, size_t n4, float *out) {
    f4 acc{0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
__global__ void copy4(const f4 *a, f4 *b, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <class F>
static double time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    for (size_t mb : {268, 1074}) {
        const size_t bytes = mb * 1000000ull / 16 * 16, n = bytes / 4, n4 = bytes / 16;
        float *a, *b, *out;
        (void)hipMalloc(&a, bytes), (void)hipMalloc(&b, bytes), (void)hipMalloc(&out, 4);
        (void)hipMemset(a, 0, bytes), (void)hipMemset(b, 0, bytes);
        for (int grid : {2048, 8192, 65536}) {
            const double w4 = time_us([&] { fill4<<<grid, 256>>>((f4 *)a, n4, 1.f); });
            const double w1 = time_us([&] { fill1<<<grid, 256>>>(a, n, 1.f); });
            const double r4 = time_us([&] { read4<<<grid, 256>>>((const f4 *)a, n4, out); });
            const double c4 = time_us([&] { copy4<<<grid, 256>>>((const f4 *)a, (f4 *)b, n4); });
            printf("%5zu MB grid %6d: write 16 B/lane %6.1f us = %.2f TB/s | write 4 B/lane %6.1f us = %.2f TB/s | read %6.1f us = %.2f TB/s | copy %6.1f us = %.2f TB/s (read + write)\n",
                   mb, grid, w4, bytes / w4 / 1e6, w1, bytes / w1 / 1e6, r4, bytes / r4 / 1e6, c4, 2.0 * bytes / c4 / 1e6);
        }
        const double ms = time_us([&] { (void)hipMemsetAsync(a, 0, bytes, 0); });
        printf("%5zu MB hipMemsetAsync: %6.1f us = %.2f TB/s\n", mb, ms, bytes / ms / 1e6);
        (void)hipFree(a), (void)hipFree(b), (void)hipFree(out);
    }
    return 0;
}
