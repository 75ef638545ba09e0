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
const char *sw_env(const char *name) { return getenv(name); }     // (the harness reads the environment as it is)
void sw_reread() {}
}  // namespace stx

int main() {
    using namespace stx;
    const int shapes[5][2] = {{64, 1 << 20}, {128, 1 << 18}, {256, 1 << 16}, {512, 1 << 14}, {512, 1 << 12}};
    float *f, *partials, *g1, *g2;
    unsigned *amax;
    hipMalloc(&f, (size_t)64 << 22);
    hipMalloc(&partials, (size_t)64 << 20);
    hipMalloc(&g1, 512 * 512 * 4);
    hipMalloc(&g2, 512 * 512 * 4);
    hipMalloc(&amax, 64 * 4);
    std::vector<float> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 0.25f;
    hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> hm(64, 0u);
    const float fmax_ = 1023.f / 512.f;
    memcpy(&hm[9], &fmax_, 4);
    hipMemcpy(amax, hm.data(), 64 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    std::vector<float> o1(512 * 512), o2(512 * 512);
    for (auto &sh : shapes) {
        const GramPlan plan = gram_plan(sh[0], sh[1]);
        float us[2];
        for (int v = 0; v < 2; ++v) {      // 0: STX_GRAM's choice without maxima (bf16x3 / fp32), 1: fp16 two-piece
            const unsigned *am = v ? amax : nullptr;
            for (int i = 0; i < 3; ++i) gram_partials_launch(0, f, plan, partials, am);
            hipEventRecord(e0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) gram_partials_launch(0, f, plan, partials, am);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[v] = ms / reps * 1e3f;
            gram_finish_launch(0, partials, plan, v ? g2 : g1, nullptr, nullptr, nullptr, nullptr, am);
        }
        const size_t n = (size_t)sh[0] * sh[0];
        hipMemcpy(o1.data(), g1, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o2.data(), g2, n * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < n; ++i) md = std::max(md, (double)fabsf(o1[i] - o2[i])), mx = std::max(mx, (double)fabsf(o1[i]));
        printf("C %3d HW %7d tiles %2d splits %3d: %6.1f us   fp16x2 %6.1f us   max |diff| %.2e of max\n", sh[0], sh[1],
               plan.tiles, plan.splits, us[0], us[1], md / mx);
    }
    return 0;
}
