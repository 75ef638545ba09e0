// Standalone timing + accuracy harness for conv_bf3.hip (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         -DSTX_EXPERIMENT_BF3 [-DSTX_BF3_TIMING] [-DSTX_BF3_SKIP=7] tools/ubench/bf3conv_bench.hip -o build_ubench/bf3conv_bench
// Prints, per shape, the time of the kernel and its error against a float64 direct convolution
// on a sample of output channels (max |err| / max |ref|).
#include "../experiments/conv_bf3.hip"

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
}  // namespace stx

__global__ void ref_conv_kernel(const float *x, const float *w, const float *bias, int K, int M, int H,
                                int W, const int *chans, int n_chans, int relu, double *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_chans * H * W) return;
    const int ci = idx / (H * W), pix = idx % (H * W), yy = pix / W, xx = pix % W;
    const int m = chans[ci];
    double s = bias ? (double)bias[m] : 0.0;
    for (int k = 0; k < K; ++k)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int y = yy + ky - 1, xq = xx + kx - 1;
                if (y < 0 || y >= H || xq < 0 || xq >= W) continue;
                s += (double)w[((size_t)m * K + k) * 9 + ky * 3 + kx] * (double)x[((size_t)k * H + y) * W + xq];
            }
    out[idx] = relu && s < 0 ? 0.0 : s;
}

static void run(int K, int M, int H, int W) {
    using namespace stx;
    const size_t xn = (size_t)K * H * W, yn = (size_t)M * H * W, wn = (size_t)M * K * 9;
    const size_t pn = bf3_packed_floats(K, M);
    float *x, *y, *w, *packed, *bias;
    hipMalloc(&x, xn * 4);
    hipMalloc(&y, yn * 4);
    hipMalloc(&w, wn * 4);
    hipMalloc(&bias, M * 4);
    hipMalloc(&packed, pn * 4);
    std::vector<float> h(std::max(xn, std::max(wn, yn)));
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    // ZEROS=<percent of zeros among the inputs> (default 25; 100: everything zero): the clock the chip
    // sustains, and with it the wall time, depends on the operands' bits
    const float thr = getenv("ZEROS") ? 0.4f * atoi(getenv("ZEROS")) : 10.f;
    for (size_t i = 0; i < xn; ++i) h[i] = std::max(0.f, rnd() * 40.f - thr);       // post-ReLU
    hipMemcpy(x, h.data(), xn * 4, hipMemcpyHostToDevice);
    const float ws = std::sqrt(2.f / (9.f * K));
    for (size_t i = 0; i < wn; ++i) h[i] = (rnd() + rnd() + rnd() + rnd() - 2.f) * 1.7f * ws;
    hipMemcpy(w, h.data(), wn * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < M; ++i) h[i] = rnd() - 0.5f;
    hipMemcpy(bias, h.data(), M * 4, hipMemcpyHostToDevice);
    hipMemset(y, 0xff, yn * 4);
    if (bf3_pack_weights(0, w, M, K, 0, packed) != 0) return;
    ConvProblem p{};
    p.x = x, p.w = packed, p.y = y, p.bias = bias, p.mask = nullptr;
    p.K = K, p.M = M, p.H = H, p.W = W, p.ksize = 3, p.relu = 1, p.epilogue = kEpiForward;
    const ConvConfig cfg = bf3_config();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    // PRE=1: the operand split once per layer (bf3_split_launch, timed separately below)
    void *split = nullptr;
    float split_ms = 0.f;
    if (getenv("PRE") && atoi(getenv("PRE"))) {
        hipMalloc(&split, bf3_split_bytes(K, H, W));
        for (int i = 0; i < 2; ++i) bf3_split_launch(0, x, K, H, W, split);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) bf3_split_launch(0, x, K, H, W, split);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&split_ms, e0, e1);
        split_ms /= 10;
        p.x_split = split;
    }
    for (int i = 0; i < 3; ++i)
        if (bf3_launch(0, cfg, p, 1) != 0) return;
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) bf3_launch(0, cfg, p, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    // accuracy on a sample of channels
    const int chans_h[8] = {0, 1, 31, 32, M / 2 + 5, M - 33, M - 2, M - 1};
    int *chans;
    double *ref;
    hipMalloc(&chans, sizeof(chans_h));
    hipMalloc(&ref, 8 * (size_t)H * W * 8);
    hipMemcpy(chans, chans_h, sizeof(chans_h), hipMemcpyHostToDevice);
    ref_conv_kernel<<<(8 * H * W + 255) / 256, 256>>>(x, w, bias, K, M, H, W, chans, 8, 1, ref);
    std::vector<double> rh(8 * (size_t)H * W);
    std::vector<float> yh(yn);
    hipMemcpy(rh.data(), ref, rh.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(yh.data(), y, yn * 4, hipMemcpyDeviceToHost);
    double max_err = 0, max_ref = 0;
    size_t bad = 0;
    for (int ci = 0; ci < 8; ++ci)
        for (size_t i = 0; i < (size_t)H * W; ++i) {
            const double r = rh[ci * (size_t)H * W + i], v = yh[(size_t)chans_h[ci] * H * W + i];
            if (!(std::fabs(v - r) <= 1e30)) ++bad;
            max_err = std::max(max_err, std::fabs(v - r));
            max_ref = std::max(max_ref, std::fabs(r));
        }
    const double flop = 2.0 * M * K * 9 * H * W;
    printf("K %4d M %4d %4dx%-4d: %.3f ms  %.1f TFLOP/s (direct-equivalent)  err %.2e of max (%zu bad)",
           K, M, H, W, ms, flop / ms / 1e9, max_err / max_ref, bad);
    if (split) printf("  + %.3f ms to split the operand (%.0f MB)", split_ms, bf3_split_bytes(K, H, W) / 1e6);
    printf("\n");
#ifdef STX_BF3_TIMING
    long long t[8][8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(stx::g_bf3_timing), sizeof(t));
    for (int wv = 0; wv < 8; wv += 7)
        printf("   wave %d: prologue %6lld  chunk loop %7lld (%.0f per chunk)  epilogue %6lld cycles;  wall %.2f / %.2f / %.2f us (loop at %.0f MHz)\n",
               wv, t[wv][0], t[wv][1], (double)t[wv][1] / (K / 16), t[wv][2], t[wv][3] / 100.0, t[wv][4] / 100.0,
               t[wv][5] / 100.0, (double)t[wv][1] / (t[wv][4] / 100.0));
#endif
    hipFree(x), hipFree(y), hipFree(w), hipFree(packed), hipFree(bias), hipFree(chans), hipFree(ref);
    if (split) hipFree(split);
}

int main(int argc, char **argv) {
    if (argc == 5) {
        run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
        return 0;
    }
    run(64, 64, 40, 50);
    run(128, 128, 91, 91);
    run(512, 512, 128, 128);
    run(256, 256, 256, 256);
    run(128, 128, 512, 512);
    run(64, 64, 1024, 1024);
    return 0;
}
