// Standalone timing harness for conv_wino2.hip (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         [-DSTX_WINO2_TIMING] tools/ubench/wino2_bench.hip -o /tmp/wino2_bench
// (the four-wave form, ALGO=4, left the library in round 6: tools/experiments/conv_wino4.hip)
// With STX_WINO2_TIMING the kernel accumulates, per wave of workgroup 0, the core-clock cycles
// spent in its compute segments, hand-over segments and barrier waits; the harness prints them.
#include "../../style_transfer_amd/csrc/conv_wino2.hip"

#include <algorithm>
#include <cmath>
#include <map>
#include <cstring>
#include <vector>

namespace stx {
#ifdef STX_WINO4_TIMING
extern __device__ long long g_wino4_timing[4][4];
#endif
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

static void run(int K, int M, int H, int W, int epilogue) {
    using namespace stx;
    const size_t xn = (size_t)K * H * W, yn = (size_t)M * H * W, wn = wino2_packed_floats(K, M);
    float *x, *y, *w, *mask;
    hipMalloc(&x, xn * 4);
    hipMalloc(&y, yn * 4);
    hipMalloc(&mask, yn * 4);
    hipMalloc(&w, wn * 4);
    std::vector<float> h(std::max(xn, std::max(wn, yn)));
    // DATA=relu: post-ReLU-like inputs (a quarter zeros) and He-scaled weights, as tools/ubench/
    // bf3conv_bench.hip uses them -- the clock the chip sustains depends on the operands' bits
    if (getenv("DATA") && !strcmp(getenv("DATA"), "relu")) {
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
        const float thr = getenv("ZEROS") ? 0.4f * atoi(getenv("ZEROS")) : 10.f;
        for (size_t i = 0; i < xn; ++i) h[i] = std::max(0.f, rnd() * 40.f - thr);
        hipMemcpy(x, h.data(), xn * 4, hipMemcpyHostToDevice);
        const float ws = std::sqrt(2.f / (9.f * K));
        for (size_t i = 0; i < wn; ++i) h[i] = (rnd() + rnd() + rnd() + rnd() - 2.f) * 1.7f * ws;
        hipMemcpy(w, h.data(), wn * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < yn; ++i) h[i] = rnd() - 0.25f;
        hipMemcpy(mask, h.data(), yn * 4, hipMemcpyHostToDevice);
    } else if (getenv("DATA") && !strcmp(getenv("DATA"), "zero")) {
        hipMemset(x, 0, xn * 4), hipMemset(w, 0, wn * 4), hipMemset(mask, 0, yn * 4);
    } else {
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f - 0.5f;
        hipMemcpy(x, h.data(), xn * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, h.data(), wn * 4, hipMemcpyHostToDevice);
        hipMemcpy(mask, h.data(), yn * 4, hipMemcpyHostToDevice);
    }
    ConvProblem p{};
    p.x = x, p.w = w, p.y = y, p.bias = nullptr, p.mask = epilogue == kEpiDgrad ? mask : nullptr;
    p.K = K, p.M = M, p.H = H, p.W = W, p.ksize = 3, p.relu = 1, p.epilogue = epilogue;
    const int geo = getenv("GEO") ? atoi(getenv("GEO")) : wino2_pick_geometry(H, W);
    const ConvConfig cfg = wino2_config(geo);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) wino2_launch(0, cfg, p, 1);
#ifdef STX_WINO2_TIMING
    {
        unsigned long long zero[8] = {0};
        hipDeviceSynchronize();
        hipMemcpyToSymbol(HIP_SYMBOL(stx::g_wino2_sums), zero, sizeof(zero));
    }
#endif
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) wino2_launch(0, cfg, p, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * M * K * 9 * H * W;
    printf("K %4d M %4d %4dx%-4d epi %d: %.3f ms  %.1f TFLOP/s (direct-equivalent)  %.1f%% of MFMA time\n",
           K, M, H, W, epilogue, ms, flop / ms / 1e9, 100.0 * (flop * 16 / 36 / 157.3e12) / (ms * 1e-3));
#ifdef STX_WINO2_STAMPS
    if (cfg.id < 210) {      // per CU: how long a workgroup runs, and how long the CU stands empty before the next one starts
        static stx::Wino2Stamp st[8192][8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(stx::g_wino2_stamps), sizeof(st));
        const int tiles = ((H + 1) / 2) * ((W + 1) / 2);
        (void)tiles;
        struct Wg { unsigned long long t0, t1, last_start; };
        std::map<unsigned, std::vector<Wg>> per_cu;
        int n = 0;
        for (int b = 0; b < 8192; ++b) {
            if (st[b][0].t1 == 0) continue;
            Wg g{~0ull, 0, 0};
            for (int wv = 0; wv < 8; ++wv) {
                g.t0 = std::min(g.t0, st[b][wv].t0), g.t1 = std::max(g.t1, st[b][wv].t1);
                g.last_start = std::max(g.last_start, st[b][wv].t0);
            }
            per_cu[(st[b][0].xcc & 0xf) << 16 | (st[b][0].hw & 0xff00)].push_back(g);
            ++n;
        }
        std::vector<double> gaps, lens, skew;
        for (auto &kv : per_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end(), [](const Wg &x, const Wg &y) { return x.t0 < y.t0; });
            for (size_t i = 0; i < v.size(); ++i) {
                lens.push_back((v[i].t1 - v[i].t0) * 0.01);
                skew.push_back((v[i].last_start - v[i].t0) * 0.01);
                if (i + 1 < v.size()) gaps.push_back(((double)v[i + 1].t0 - (double)v[i].t1) * 0.01);
            }
        }
        auto pct = [](std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0. : v[(size_t)(p * (v.size() - 1))]; };
        printf("   stamps: %d workgroups on %zu CUs; workgroup first wave in -> last wave out: median %.2f us (p10 %.2f, p90 %.2f); "
               "CU empty between two workgroups: median %.2f us (p10 %.2f, p90 %.2f, max %.2f); last wave starts %.2f us after the first\n",
               n, per_cu.size(), pct(lens, .5), pct(lens, .1), pct(lens, .9), pct(gaps, .5), pct(gaps, .1), pct(gaps, .9), pct(gaps, 1.), pct(skew, .5));
        static stx::Wino2Stamp zero[8192][8];
        hipMemcpyToSymbol(HIP_SYMBOL(stx::g_wino2_stamps), zero, sizeof(zero));
    }
#endif
#ifdef STX_WINO4_TIMING
    if (cfg.id >= 210) {
        long long t4[4][4];
        hipMemcpyFromSymbol(t4, HIP_SYMBOL(stx::g_wino4_timing), sizeof(t4));
        for (int wv = 0; wv < 4; wv += 3)
            printf("   wave %d: prologue %6lld  chunk loop %7lld (%.0f per chunk)  epilogue %6lld cycles; workgroup %.2f us at %.0f MHz\n", wv,
                   t4[wv][0], t4[wv][1], (double)t4[wv][1] / ((K + 7) / 8), t4[wv][2], t4[wv][3] / 100.0,
                   (double)(t4[wv][0] + t4[wv][1] + t4[wv][2]) / (t4[wv][3] / 100.0));
    }
#endif
#ifdef STX_WINO2_TIMING
    if (cfg.id < 210) {
        unsigned long long sm[8];
        hipMemcpyFromSymbol(sm, HIP_SYMBOL(stx::g_wino2_sums), sizeof(sm));
        const double n = (double)sm[0];
        printf("   mean over %.0f workgroups: setup %5.0f  first loads -> LDS %5.0f  hand-over %5.0f  chunk loop %6.0f (%.0f per chunk)  "
               "last two chunks + epilogue %6.0f  = %6.0f cycles per workgroup; x %d workgroups / 256 CUs / %.3f ms = %.0f MHz\n",
               n, sm[1] / n, sm[2] / n, sm[3] / n, sm[4] / n, sm[4] / n / std::max(1, (K + 7) / 8 - 2), sm[5] / n, sm[6] / n,
               (int)(n / reps), ms, sm[6] / n * (n / reps) / 256.0 / (ms * 1e3));
    }
    long long t[8][8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(stx::g_wino2_timing), sizeof(t));
    const int chunks = (K + 7) / 8 - 2;
    for (int wv = 0; wv < 8; wv += 4)
        printf("   wave %d: per chunk  work %6.0f  |  prologue %6lld (setup %lld, loads -> LDS %lld)  chunk loop %7lld  last two chunks + epilogue %6lld cycles; chunk loop at %.0f MHz\n", wv,
               (double)t[wv][0] / chunks, t[wv][4], t[wv][6], t[wv][7], t[wv][3], t[wv][5],
               (double)t[wv][3] / (t[wv][2] / 100.0));
#endif
    hipFree(x), hipFree(y), hipFree(w), hipFree(mask);
}

int main() {
    run(512, 512, 64, 64, stx::kEpiForward);
    run(512, 512, 128, 128, stx::kEpiForward);
    run(256, 256, 256, 256, stx::kEpiForward);
    run(128, 128, 512, 512, stx::kEpiForward);
    run(64, 64, 1024, 1024, stx::kEpiForward);
    run(512, 512, 128, 128, stx::kEpiDgrad);
    return 0;
}
