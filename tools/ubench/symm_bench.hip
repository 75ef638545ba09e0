// Standalone timing harness for symm_bf3_kernel / symm_h2_kernel (tuning aid, not part of the library).
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
const char *sw_env(const char *name) { return getenv(name); }     // (the harness reads the environment as it is)
void sw_reread() {}
}  // namespace stx

int main() {
    using namespace stx;
    const int shapes[5][2] = {{64, 1 << 20}, {128, 1 << 18}, {256, 1 << 16}, {512, 1 << 14}, {512, 1 << 12}};
    float *f, *out, *out2, *dsym, *partials;
    unsigned short *pieces;
    unsigned *amax;
    hipMalloc(&f, (size_t)64 << 22);
    hipMalloc(&out, (size_t)64 << 22);
    hipMalloc(&out2, (size_t)64 << 22);
    hipMalloc(&dsym, 512 * 512 * 4);
    hipMalloc(&pieces, 3 * 512 * 512 * 2);
    hipMalloc(&partials, 1 << 20);
    hipMalloc(&amax, (4096 + 64) * 4);
    std::vector<float> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 0.25f;
    hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsym, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice);
    // the maxima the fp16 two-piece kernel scales by: 4096 block words for D, 64 slots for F
    std::vector<unsigned> hm(4096 + 64, 0u);
    const float dmax = 1023.f / 512.f, fmax_ = 1023.f / 512.f;
    memcpy(&hm[17], &dmax, 4);
    memcpy(&hm[4096 + 5], &fmax_, 4);
    hipMemcpy(amax, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    std::vector<float> o1, o2;
    for (auto &sh : shapes) {
        const int C = sh[0], HW = sh[1];
        float us[2];
        for (int v = 0; v < 2; ++v) {
            auto run = [&](bool first) {
                if (v == 0) symm_bf3_launch(0, f, dsym, pieces, !first, out, partials, C, HW);
                else symm_h2_launch(0, f, dsym, amax, C * C / 64, amax + 4096, out2, partials, C, HW);
            };
            for (int i = 0; i < 3; ++i) run(i == 0);
            hipEventRecord(e0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) run(false);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[v] = ms / reps * 1e3f;
        }
        // the two forms against each other on a sample of the output
        const size_t n = std::min((size_t)C * HW, (size_t)1 << 22);
        o1.resize(n), o2.resize(n);
        hipMemcpy(o1.data(), out, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o2.data(), out2, n * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < n; ++i) {
            const double d = fabs((double)o1[i] - (double)o2[i]);
            md = d <= md ? md : d;        // (a NaN counts)
            mx = std::max(mx, (double)fabsf(o1[i]));
        }
        printf("C %3d HW %7d workgroups %4d: bf16x3 %6.1f us   fp16x2 %6.1f us   max |diff| %.2e of max\n", C, HW,
               symm_num_workgroups(C, HW), us[0], us[1], md / mx);
    }
    return 0;
}
