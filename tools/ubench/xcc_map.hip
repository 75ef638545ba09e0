// Which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID, gfx940+.)  The XCD-aware work
// orders of the kernels assume b & 7.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcc_map.hip -o build_ubench/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 0xf);
    // keep the CU busy for a while so that later workgroups queue behind earlier ones
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
int main() {
    for (int n : {64, 256, 552, 1024, 4099}) {
        for (int threads : {256, 512}) {
            int *d;
            hipMalloc(&d, n * 4);
            hipLaunchKernelGGL(k, dim3(n), dim3(threads), 65536, 0, d, 20000);
            std::vector<int> h(n);
            hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
            int bad = 0, first_bad = -1;
            for (int b = 0; b < n; ++b)
                if (h[b] != (h[0] + b) % 8 && h[b] != ((b & 7) + h[0]) % 8) { if (!bad) first_bad = b; ++bad; }
            printf("n %5d threads %3d: xcc of b=0..15:", n, threads);
            for (int b = 0; b < 16 && b < n; ++b) printf(" %d", h[b]);
            printf("   mismatches vs (b & 7) + const: %d (first %d)\n", bad, first_bad);
            hipFree(d);
        }
    }
    return 0;
}
