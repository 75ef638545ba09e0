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

// The access pattern of every kernel that produces or consumes a [64][H][W] blob tile by tile: a
// workgroup writes (reads) a 64-plane x 128-pixel tile -- 64 pieces of 512 bytes, one per plane,
// 4 MB apart on a 1024^2 plane -- 16 bytes per lane, two planes per wave instruction; PLANES = how
// many planes a workgroup's tile spans (64: the layers' pattern; 1: one contiguous 32 KB run).
template <int PLANES, bool READ>
__global__ void tile_planes(f4 *p, size_t plane_f4, int tiles_per_plane, int n_tiles, float *out) {
    constexpr int kPix4 = 128 * 64 / PLANES / 4;          // 16-byte pieces per plane and tile
    f4 acc{0, 0, 0, 0};
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tp = t % tiles_per_plane, group = t / tiles_per_plane;     // (PLANES planes per group)
        for (int e = threadIdx.x; e < PLANES * kPix4; e += blockDim.x) {
            const int pl = e / kPix4, c = e % kPix4;
            f4 *q = p + (size_t)(group * PLANES + pl) * plane_f4 + (size_t)tp * kPix4 + c;
            if (READ) acc += *q;
            else *q = f4{1.f, 2.f, 3.f, 4.f};
        }
    }
    if (READ && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
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
        if (mb == 268) {      // a 64 x 1024 x 1024 blob
            const size_t plane_f4 = 1024 * 1024 / 4;
            auto run = [&](auto kern, int planes, const char *what) {
                const int kpix4 = 128 * 64 / planes / 4, tiles_per_plane = (int)(plane_f4 / kpix4);
                const int n_tiles = tiles_per_plane * (64 / planes);
                for (int grid : {512, 2048}) {
                    const double us = time_us([&] { kern<<<grid, 256>>>((f4 *)a, plane_f4, tiles_per_plane, n_tiles, out); });
                    printf("  tiles of %2d planes x %5d pixels, %s, grid %4d: %6.1f us = %.2f TB/s\n", planes,
                           128 * 64 / planes, what, grid, us, 64.0 * plane_f4 * 16 / us / 1e6);
                }
            };
            run(tile_planes<64, false>, 64, "write");
            run(tile_planes<16, false>, 16, "write");
            run(tile_planes<4, false>, 4, "write");
            run(tile_planes<1, false>, 1, "write");
            run(tile_planes<64, true>, 64, "read ");
            run(tile_planes<1, true>, 1, "read ");
        }
        const double ms = time_us([&] { (void)hipMemsetAsync(a, 0, bytes, 0); });
        printf("%5zu MB hipMemsetAsync: %6.1f us = %.2f TB/s\n", mb, ms, bytes / ms / 1e6);
        (void)hipFree(a), (void)hipFree(b), (void)hipFree(out);
    }
    return 0;
}
