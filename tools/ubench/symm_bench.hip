// Standalone timing harness for symm_bf3_kernel (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         [-DSTX_SYMM_SKIP=n] tools/ubench/symm_bench.hip -o build_ubench/symm_bench[_n]
#include "../../style_transfer_amd/csrc/symm.hip"

#include <vector>

namespace stx {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace stx

int main() {
    using namespace stx;
    const int shapes[5][2] = {{64, 1 << 20}, {128, 1 << 18}, {256, 1 << 16}, {512, 1 << 14}, {512, 1 << 12}};
    float *f, *out, *dsym, *partials;
    unsigned short *pieces;
    hipMalloc(&f, (size_t)64 << 22);
    hipMalloc(&out, (size_t)64 << 22);
    hipMalloc(&dsym, 512 * 512 * 4);
    hipMalloc(&pieces, 3 * 512 * 512 * 2);
    hipMalloc(&partials, 1 << 20);
    std::vector<float> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f;
    hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsym, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (auto &sh : shapes) {
        for (int i = 0; i < 3; ++i) symm_bf3_launch(0, f, dsym, pieces, false, out, partials, sh[0], sh[1]);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) symm_bf3_launch(0, f, dsym, pieces, true, out, partials, sh[0], sh[1]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("C %3d HW %7d workgroups %4d: %6.1f us\n", sh[0], sh[1], symm_num_workgroups(sh[0], sh[1]),
               ms / reps * 1e3);
    }
    return 0;
}
