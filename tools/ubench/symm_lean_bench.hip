// The register-lean SYMM kernel (tools/experiments/symm_lean.hip) against the shipped one: results,
// time alone, and time beside a run of convolution launches on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         tools/ubench/symm_lean_bench.hip style_transfer_amd/csrc/conv_wino4.hip -o build_ubench/symm_lean_bench
#include "../../style_transfer_amd/csrc/conv_wino2.hip"
#include "../../style_transfer_amd/csrc/symm.hip"
#include "../experiments/symm_lean.hip"

#include <cstdarg>
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
int splitk_reduce_launch(hipStream_t, const ConvProblem &, int) { return 0; }
}  // namespace stx

static void run(int C, int HW) {
    using namespace stx;
    const size_t fn = (size_t)C * HW;
    float *F, *D, *S0, *S1, *P0, *P1;
    unsigned short *pieces;
    hipMalloc(&F, fn * 4), hipMalloc(&S0, fn * 4), hipMalloc(&S1, fn * 4), hipMalloc(&D, (size_t)C * C * 4);
    hipMalloc(&pieces, symm_pieces_elems(C) * 2);
    const int n0 = symm_num_workgroups(C, HW), n1 = symm_lean_num_workgroups(C, HW);
    hipMalloc(&P0, n0 * 4), hipMalloc(&P1, n1 * 4);
    std::vector<float> h(fn);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (size_t i = 0; i < fn; ++i) h[i] = std::max(0.f, rnd() * 40.f - 15.f);
    hipMemcpy(F, h.data(), fn * 4, hipMemcpyHostToDevice);
    std::vector<float> d((size_t)C * C);
    for (int i = 0; i < C; ++i)
        for (int j = 0; j <= i; ++j) d[(size_t)i * C + j] = d[(size_t)j * C + i] = rnd() - 0.5f;
    hipMemcpy(D, d.data(), d.size() * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking), hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    symm_bf3_launch(sb, F, D, pieces, false, S0, P0, C, HW);
    symm_lean_launch(sb, F, pieces, S1, P1, C, HW);
    hipDeviceSynchronize();
    std::vector<float> a(fn), b(fn), p0(n0), p1(n1);
    hipMemcpy(a.data(), S0, fn * 4, hipMemcpyDeviceToHost), hipMemcpy(b.data(), S1, fn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p0.data(), P0, n0 * 4, hipMemcpyDeviceToHost), hipMemcpy(p1.data(), P1, n1 * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0, s0 = 0, s1 = 0;
    for (size_t i = 0; i < fn; ++i) md = std::max(md, (double)std::fabs(a[i] - b[i])), mx = std::max(mx, (double)std::fabs(a[i]));
    for (float v : p0) s0 += v;
    for (float v : p1) s1 += v;
    // convolutions for the shadow test
    const int K = 512, M = 512, H = 128, W = 128;
    float *x, *y, *w;
    hipMalloc(&x, (size_t)K * H * W * 4), hipMalloc(&y, (size_t)M * H * W * 4), hipMalloc(&w, wino2_packed_floats(K, M) * 4);
    hipMemset(x, 0, (size_t)K * H * W * 4), hipMemset(w, 0, wino2_packed_floats(K, M) * 4);
    ConvProblem p{};
    p.x = x, p.w = w, p.y = y, p.K = K, p.M = M, p.H = H, p.W = W, p.ksize = 3, p.relu = 1, p.epilogue = kEpiForward;
    const ConvConfig cfg = wino2_config(0);
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0), hipEventCreate(&a1), hipEventCreate(&b0), hipEventCreate(&b1);
    const int convs = 10, reps = 10;
    auto conv_run = [&]() {
        hipEventRecord(a0, sa);
        for (int i = 0; i < convs; ++i) wino2_launch(sa, cfg, p, 1);
        hipEventRecord(a1, sa);
    };
    auto symm_run = [&](int lean) {
        hipEventRecord(b0, sb);
        for (int i = 0; i < reps; ++i) {
            if (lean) symm_lean_launch(sb, F, pieces, S1, P1, C, HW);
            else symm_bf3_launch(sb, F, D, pieces, true, S0, P0, C, HW);
        }
        hipEventRecord(b1, sb);
    };
    float tc, t0, t1, tc0, ts0, tc1, ts1;
    conv_run(), hipDeviceSynchronize();
    conv_run(), hipDeviceSynchronize(), hipEventElapsedTime(&tc, a0, a1);
    symm_run(0), hipDeviceSynchronize(), hipEventElapsedTime(&t0, b0, b1);
    symm_run(1), hipDeviceSynchronize(), hipEventElapsedTime(&t1, b0, b1);
    conv_run(), symm_run(0), hipDeviceSynchronize(), hipEventElapsedTime(&tc0, a0, a1), hipEventElapsedTime(&ts0, b0, b1);
    conv_run(), symm_run(1), hipDeviceSynchronize(), hipEventElapsedTime(&tc1, a0, a1), hipEventElapsedTime(&ts1, b0, b1);
    printf("C %3d HW %7d: lean vs shipped max |diff| %.2e of max, sum|S| %.9g vs %.9g\n", C, HW, md / mx, s1, s0);
    printf("   alone: convolutions %.3f ms, %d x shipped %.3f ms, %d x lean %.3f ms\n", tc, reps, t0, reps, t1);
    printf("   together with shipped: convolutions %.3f ms, symm %.3f ms;  with lean: convolutions %.3f ms, symm %.3f ms\n",
           tc0, ts0, tc1, ts1);
    hipFree(F), hipFree(D), hipFree(S0), hipFree(S1), hipFree(P0), hipFree(P1), hipFree(pieces), hipFree(x), hipFree(y), hipFree(w);
}

int main() {
    run(512, 16384);
    run(256, 65536);
    run(64, 1048576);
    return 0;
}
