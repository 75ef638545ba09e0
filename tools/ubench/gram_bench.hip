// Standalone timing harness for the Gram kernels (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         [-DSTX_GRAM_SKIP=n] tools/ubench/gram_bench.hip -o build_ubench/gram_bench[_n]
// Times gram_partials_launch on the five style layers of a 1024^2 VGG-19 tile; STX_GRAM=fp32 in
// the environment selects the fp32-MFMA kernel, STX_GRAM_SKIP variants remove one ingredient.
#include "../../style_transfer_amd/csrc/gram.hip"

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
    float *f, *partials;
    hipMalloc(&f, (size_t)64 << 22);
    hipMalloc(&partials, (size_t)64 << 20);
    std::vector<float> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f;
    hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (auto &sh : shapes) {
        const GramPlan plan = gram_plan(sh[0], sh[1]);
        for (int i = 0; i < 3; ++i) gram_partials_launch(0, f, plan, partials);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) gram_partials_launch(0, f, plan, partials);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("C %3d HW %7d tiles %2d splits %3d: %6.1f us\n", sh[0], sh[1], plan.tiles, plan.splits,
               ms / reps * 1e3);
    }
    return 0;
}
