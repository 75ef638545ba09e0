// Read bandwidth out of the L2 / the Infinity Cache by access width (what bounds kernels that
// re-read an L2-resident operand from many CUs: the Gram and SYMM kernels all level off near
// 6 TB/s of CU-side traffic).   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_bw.hip -o build_ubench/l2_bw
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T *buf, size_t n_elems, int iters, float *out) {
    // every block sweeps the whole buffer; lanes read consecutive elements (coalesced)
    const size_t stride = (size_t)blockDim.x;
    size_t idx = (threadIdx.x + (size_t)blockIdx.x * 4096) % n_elems;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            const T v = buf[idx];
            acc += reinterpret_cast<const float *>(&v)[0];
            idx += stride;
            if (idx >= n_elems) idx -= n_elems;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void write_kernel(T *buf, size_t n_elems, int iters) {
    // the grid writes the buffer once per iteration, lanes write consecutive elements
    T v;
    reinterpret_cast<float *>(&v)[0] = (float)threadIdx.x;
    for (int it = 0; it < iters; ++it)
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * 256) buf[i] = v;
}

template <typename T>
static void run_write(const char *label, size_t bytes, int blocks) {
    T *buf;
    hipMalloc(&buf, bytes);
    const size_t n = bytes / sizeof(T);
    const int iters = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    write_kernel<T><<<blocks, 256>>>(buf, n, 2);
    hipEventRecord(e0);
    write_kernel<T><<<blocks, 256>>>(buf, n, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("store %-14s buffer %7.1f MB, %5d blocks: %7.2f TB/s\n", label, bytes / 1e6, blocks,
           (double)bytes * iters / ms / 1e9);
    hipFree(buf);
}

template <typename T>
static void run(const char *label, size_t bytes, int blocks) {
    T *buf;
    float *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    const size_t n = bytes / sizeof(T);
    const int iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    read_kernel<T><<<blocks, 256>>>(buf, n, 4, out);
    hipEventRecord(e0);
    read_kernel<T><<<blocks, 256>>>(buf, n, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * 256 * iters * 64 * sizeof(T);
    printf("%-14s buffer %7.1f MB, %5d blocks: %7.2f TB/s\n", label, bytes / 1e6, blocks, total / ms / 1e9);
    hipFree(buf);
    hipFree(out);
}

int main() {
    for (size_t mb : {1, 2, 16, 64, 200}) {
        run<float>("4 B per lane", mb << 20, 2048);
        run<float2>("8 B per lane", mb << 20, 2048);
        run<float4>("16 B per lane", mb << 20, 2048);
    }
    for (size_t mb : {64, 268, 1024}) {
        run_write<float>("4 B per lane", mb << 20, 4096);
        run_write<float2>("8 B per lane", mb << 20, 4096);
        run_write<float4>("16 B per lane", mb << 20, 4096);
    }
    return 0;
}
