// Standalone timing + accuracy harness for csrc/conv_h2.hip (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         [-DSTX_H2_TIMING] [-DSTX_H2_SKIP=7] tools/ubench/h2conv_bench.hip -o tools/ubench/bin/h2conv_bench
// Prints, per shape and channel-tile variant, the time of the kernel, its error against a float64 direct
// convolution on a sample of output channels (max |err| / max |ref|) and whether the maximum it left
// for the next layer is the maximum of what it wrote.  First: does the fp16 MFMA keep fp16 subnormals?
#include "../../style_transfer_amd/csrc/conv_h2.hip"

#include <cstdarg>
#include <cmath>
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
int splitk_reduce_launch(hipStream_t, const ConvProblem &, int) { return STX_ERR_UNSUPPORTED; }
}  // namespace stx

// ---- probe: fp16 subnormal operands of v_mfma_f32_32x32x16_f16, and the split's own instructions
__global__ void probe_kernel(float *out) {
    using namespace stx;
    const int lane = threadIdx.x;
    f16x8 a, b;
    // A[m][k] = 2^-20 (an fp16 subnormal) for k = 0, else 0;  B[k][n] = 2^10 for k = 0
    for (int e = 0; e < 8; ++e) {
        a[e] = (lane < 32 && e == 0) ? (_Float16)9.5367431640625e-07f : (_Float16)0.f;
        b[e] = (lane < 32 && e == 0) ? (_Float16)1024.f : (_Float16)0.f;
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];               // 2^-10 if subnormals are kept, 0 if flushed
    // the split of one value: hi, lo and the residual
    const float v = 1234.56789f + lane, s = 4.f;
    unsigned hi, lo;
    float ra, rb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(-v), "v"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(v), "v"(s), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(-v), "v"(s), "v"(hi));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
    if (lane == 0) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = __builtin_bit_cast(h2, hi), l = __builtin_bit_cast(h2, lo);
        out[1] = (float)h[0], out[2] = (float)h[1], out[3] = ra, out[4] = rb, out[5] = (float)l[0], out[6] = (float)l[1];
        out[7] = v * s;
    }
}

__global__ void ref_conv_kernel(const float *x, const float *w, const float *bias, int K, int M, int H,
                                int W, const int *chans, int n_chans, int relu, int flip, double *out) {
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
                // flip: backward-to-data, w is [K][M][3][3] and the taps are rotated
                const double wv = flip ? w[((size_t)k * M + m) * 9 + (8 - (ky * 3 + kx))]
                                       : w[((size_t)m * K + k) * 9 + ky * 3 + kx];
                s += wv * (double)x[((size_t)k * H + y) * W + xq];
            }
    out[idx] = relu && s < 0 ? 0.0 : s;
}

static void run(int K, int M, int H, int W, int mb, int backward) {
    using namespace stx;
    const size_t xn = (size_t)K * H * W, yn = (size_t)M * H * W, wn = (size_t)M * K * 9;
    const size_t pn = h2_packed_floats(K, M);
    float *x, *y, *w, *packed, *bias, *mask;
    unsigned *amax;
    hipMalloc(&x, xn * 4);
    hipMalloc(&y, yn * 4);
    hipMalloc(&mask, yn * 4);
    hipMalloc(&w, wn * 4);
    hipMalloc(&bias, M * 4);
    hipMalloc(&packed, pn * 4);
    hipMalloc(&amax, 2 * kAmaxSlots * 4);
    hipMemset(amax, 0, 2 * kAmaxSlots * 4);
    std::vector<float> h(std::max(xn, std::max(wn, yn)));
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    const float thr = getenv("ZEROS") ? 0.4f * atoi(getenv("ZEROS")) : 10.f;
    const float gscale = getenv("GSCALE") ? (float)atof(getenv("GSCALE")) : 1.f;
    if (!backward) {
        for (size_t i = 0; i < xn; ++i) h[i] = std::max(0.f, rnd() * 40.f - thr);       // post-ReLU
    } else {   // a gradient: signed, heavy-tailed
        for (size_t i = 0; i < xn; ++i) {
            const float g = (rnd() + rnd() + rnd() + rnd() - 2.f) * 1.7f;
            h[i] = gscale * 0.1f * g * std::exp(1.5f * (rnd() + rnd() + rnd() + rnd() - 2.f) * 1.7f);
        }
    }
    // PIN=1 (MAX) / 2 (AVE), backward only: the gradient arrives pooled, with window codes (PINMASK=0: the
    // blob under the pooling layer is not rectified); x = what pool_bwd_codes_kernel would have written
    const int pin = backward && getenv("PIN") ? atoi(getenv("PIN")) : 0;
    const bool pin_mask = !getenv("PINMASK") || atoi(getenv("PINMASK"));
    const int ph = (H + 1) / 2, pw = (W + 1) / 2;
    float *dyp = nullptr;
    unsigned char *codes = nullptr;
    if (pin) {
        std::vector<float> dy((size_t)K * ph * pw);
        std::vector<unsigned char> cd(dy.size() + 4, 0);
        for (size_t i = 0; i < dy.size(); ++i) {
            dy[i] = h[i];
            cd[i] = (unsigned char)(rnd() * (pin == 1 ? 8.f : 16.f));
        }
        for (int k = 0; k < K; ++k)
            for (int oy = 0; oy < ph; ++oy)
                for (int ox = 0; ox < pw; ++ox) {
                    const size_t i = ((size_t)k * ph + oy) * pw + ox;
                    const bool hx = 2 * ox + 1 < W, hy = 2 * oy + 1 < H;
                    const float g = dy[i];
                    const unsigned code = cd[i];
                    float o[4];
                    if (pin == 1) {
                        const float gs = (!pin_mask || (code & 4u)) ? g : 0.f;
                        for (int e = 0; e < 4; ++e) o[e] = (code & 3u) == (unsigned)e ? gs : 0.f;
                    } else {
                        const float gq = g / ((hx ? 2.f : 1.f) * (hy ? 2.f : 1.f));
                        for (int e = 0; e < 4; ++e) o[e] = (!pin_mask || (code >> e & 1u)) ? gq : 0.f;
                    }
                    for (int e = 0; e < 4; ++e) {
                        const int yy = 2 * oy + (e >> 1), xx = 2 * ox + (e & 1);
                        if (yy < H && xx < W) h[((size_t)k * H + yy) * W + xx] = o[e];
                    }
                }
        hipMalloc(&dyp, dy.size() * 4);
        hipMalloc(&codes, cd.size());
        hipMemcpy(dyp, dy.data(), dy.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(codes, cd.data(), cd.size(), hipMemcpyHostToDevice);
    }
    hipMemcpy(x, h.data(), xn * 4, hipMemcpyHostToDevice);
    const float ws = std::sqrt(2.f / (9.f * K));
    for (size_t i = 0; i < wn; ++i) h[i] = (rnd() + rnd() + rnd() + rnd() - 2.f) * 1.7f * ws;
    hipMemcpy(w, h.data(), wn * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < M; ++i) h[i] = rnd() - 0.5f;
    hipMemcpy(bias, h.data(), M * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < yn; ++i) h[i] = rnd() - 0.3f;
    hipMemcpy(mask, h.data(), yn * 4, hipMemcpyHostToDevice);
    hipMemset(y, 0xff, yn * 4);
    // forward: w is [M][K][3][3]; backward: the layer's bank is [Mo = K][Ko = M] and the kernel's
    // output channels are the layer's inputs
    if (h2_pack_weights(0, w, backward ? K : M, backward ? M : K, backward, packed) != 0) return;
    if (absmax_launch(0, pin ? dyp : x, pin ? (size_t)K * ph * pw : xn, amax) != 0) return;
    ConvProblem p{};
    p.x = x, p.w = packed, p.y = y, p.bias = backward ? nullptr : bias, p.mask = backward ? mask : nullptr;
    p.K = K, p.M = M, p.H = H, p.W = W, p.ksize = 3, p.relu = backward ? 0 : 1;
    p.epilogue = backward ? kEpiDgrad : kEpiForward;
    p.x_amax = amax, p.y_amax = amax + kAmaxSlots;
    // INJECT=1 (backward only): the style term of a tapped blob rides in the epilogue (timing only: the
    // reference below does not add it)
    float *sg = nullptr, *sabs = nullptr;
    if (backward && getenv("INJECT") && atoi(getenv("INJECT"))) {
        hipMalloc(&sg, yn * 4);
        hipMalloc(&sabs, 4);
        hipMemcpy(sg, mask, yn * 4, hipMemcpyDeviceToDevice);
        const float one = 1.f * yn;
        hipMemcpy(sabs, &one, 4, hipMemcpyHostToDevice);
        p.inject.sgrad = sg, p.inject.s_abs_sum = sabs, p.inject.s_coef = 0.f;
    }
    if (pin) p.x = dyp, p.pin_codes = codes, p.pin_mode = pin == 1 ? STX_POOL_MAX : STX_POOL_AVE, p.pin_mask = pin_mask;
    const ConvConfig cfg = mb == 3 ? h2_config(1, 2) : h2_config(mb);       // 1: 64 ch, 2: 128 ch, 3: 64 ch x two patches
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        if (h2_launch(0, cfg, p, 1) != 0) return;
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) h2_launch(0, cfg, p, 1);
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
    ref_conv_kernel<<<(8 * H * W + 255) / 256, 256>>>(x, w, backward ? nullptr : bias, K, M, H, W, chans, 8,
                                                      backward ? 0 : 1, backward, ref);
    std::vector<double> rh(8 * (size_t)H * W);
    std::vector<float> yh(yn), mh(yn);
    unsigned ah[2 * kAmaxSlots];
    hipMemcpy(rh.data(), ref, rh.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(yh.data(), y, yn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(mh.data(), mask, yn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ah, amax, sizeof(ah), hipMemcpyDeviceToHost);
    double max_err = 0, max_ref = 0;
    size_t bad = 0;
    for (int ci = 0; ci < 8; ++ci)
        for (size_t i = 0; i < (size_t)H * W; ++i) {
            double r = rh[ci * (size_t)H * W + i];
            const size_t yi = (size_t)chans_h[ci] * H * W + i;
            if (backward && !(mh[yi] > 0.f)) r = 0;
            const double v = yh[yi];
            if (!(std::fabs(v - r) <= 1e30)) ++bad;
            max_err = std::max(max_err, std::fabs(v - r));
            max_ref = std::max(max_ref, std::fabs(r));
        }
    float ymax = 0.f, kmax = 0.f;
    for (size_t i = 0; i < yn; ++i) ymax = std::max(ymax, std::fabs(yh[i]));
    for (int i = 0; i < kAmaxSlots; ++i) kmax = std::max(kmax, *reinterpret_cast<float *>(&ah[kAmaxSlots + i]));
    const double flop = 2.0 * M * K * 9 * H * W;
    if (pin) printf("(pooled input, %s%s) ", pin == 1 ? "MAX" : "AVE", pin_mask ? ", rectified" : "");
    printf("%s K %4d M %4d %4dx%-4d MB %d: %.3f ms  %.1f TFLOP/s (direct-equivalent)  err %.2e of max (%zu bad)  max |y| %s (%g / %g)\n",
           backward ? "bwd" : "fwd", K, M, H, W, mb, ms, flop / ms / 1e9, max_err / max_ref, bad,
           ymax == kmax ? "ok" : "WRONG", kmax, ymax);
#ifdef STX_H2_TIMING
    long long t[8][8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(stx::g_h2_timing), sizeof(t));
    for (int wv = 0; wv < 8; wv += 7)
        printf("   wave %d: prologue %6lld  chunk loop %7lld (%.0f per chunk)  epilogue %6lld cycles;  wall %.2f / %.2f / %.2f us (loop at %.0f MHz)\n",
               wv, t[wv][0], t[wv][1], (double)t[wv][1] / (K / 16), t[wv][2], t[wv][3] / 100.0, t[wv][4] / 100.0,
               t[wv][5] / 100.0, (double)t[wv][1] / (t[wv][4] / 100.0));
    long long te[8][8];
    hipMemcpyFromSymbol(te, HIP_SYMBOL(stx::g_h2_epi), sizeof(te));
    printf("   prologue: setup %lld, loads requested + epilogue addresses %lld, wait for the loads %lld, first staging %lld, barrier + first B read %lld\n",
           te[0][4], te[0][5], te[0][6], te[0][7], t[0][0] - te[0][4] - te[0][5] - te[0][6] - te[0][7]);
    printf("   epilogue pass 0, cycles after the loop: loads issued %lld, through the first barrier %lld, exchange written + barrier %lld, pass done %lld\n",
           te[0][0], te[0][1], te[0][2], te[0][3]);
#endif
    hipFree(x), hipFree(y), hipFree(w), hipFree(packed), hipFree(bias), hipFree(chans), hipFree(ref);
    hipFree(mask), hipFree(amax);
}

int main(int argc, char **argv) {
    float *out, oh[8];
    hipMalloc(&out, 32);
    probe_kernel<<<1, 64>>>(out);
    hipMemcpy(oh, out, 32, hipMemcpyDeviceToHost);
    printf("fp16 MFMA, subnormal A operand 2^-20 x 2^10: %g (%s)\n", oh[0], oh[0] != 0.f ? "kept" : "FLUSHED");
    printf("split of %.7g: hi %.7g / %.7g  residual %.7g / %.7g  lo %.7g / %.7g\n", oh[7], oh[1], oh[2], oh[3], oh[4], oh[5], oh[6]);
    if (argc == 7) {
        run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
        return 0;
    }
    for (int mb = 1; mb <= 3; ++mb) {
        run(64, 64, 40, 50, mb, 0);
        run(128, 128, 91, 91, mb, 0);
        run(128, 128, 91, 91, mb, 1);
        run(512, 512, 128, 128, mb, 0);
        run(512, 512, 128, 128, mb, 1);
        run(256, 256, 256, 256, mb, 0);
        run(256, 256, 256, 256, mb, 1);
        run(512, 256, 128, 128, mb, 1);
        run(128, 128, 512, 512, mb, 0);
    }
    return 0;
}
