// 2x2 stride-2 ceil-mode pooling, forward and backward (HBM-bound, one pass each).
//
// Caffe Pooling layer semantics as used by vgg19.prototxt:50-60 / vgg16_avgpool.prototxt:
// output size ceil((n-2)/2)+1; windows are clipped at the bottom/right edge; MAX keeps the FIRST
// maximum in row-major window order (strict '>' scan -- post-ReLU inputs have many all-zero
// windows, so ties are common); AVE divides by the clipped window size.  The backward kernel
// recomputes the argmax from the pool input instead of storing it, fuses the ReLU mask of the
// conv blob below (diff *= data > 0, style_transfer.py:608-610 via relu layers), and since
// windows do not overlap (kernel == stride) every input gradient is written exactly once.
//
// Window codes.  A forward pass that also leaves one byte per window behind (pool_fwd_kernel with
// `codes`, or the convolution epilogue that produced the pooled blob: conv_wino2.hip) lets the
// backward pass run without the pool input at all -- 1/16 of its bytes (bwd pool1 of a 1024^2
// tile: 336 -> 84 MB fetched).  MAX: bits 0-1 = index of the first maximum, bit 2 = that maximum
// is > 0 (the ReLU mask of the only element that receives gradient).  AVE: bits 0-3 = element > 0.

#include "common.h"

namespace stx {

template <int MODE>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float *__restrict__ x, int C, int H,
                                                       int W, int Ho, int Wo,
                                                       float *__restrict__ y,
                                                       unsigned char *__restrict__ codes) {
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % Wo;
        const int oy = (i / Wo) % Ho;
        const int c = i / ((size_t)Wo * Ho);
        const int y0 = 2 * oy, x0 = 2 * ox;
        const float *p = x + ((size_t)c * H + y0) * W + x0;
        const bool hx = x0 + 1 < W, hy = y0 + 1 < H;
        const float v00 = p[0];
        const float v01 = hx ? p[1] : 0.f;
        const float v10 = hy ? p[W] : 0.f;
        const float v11 = (hx && hy) ? p[W + 1] : 0.f;
        float out;
        unsigned code;
        if (MODE == STX_POOL_MAX) {
            out = v00;
            if (hx) out = fmaxf(out, v01);
            if (hy) out = fmaxf(out, v10);
            if (hx && hy) out = fmaxf(out, v11);
            code = pool_max_code(v00, v01, v10, v11, hx, hy);
        } else {
            const float cnt = (hx ? 2.f : 1.f) * (hy ? 2.f : 1.f);
            out = (v00 + v01 + v10 + v11) / cnt;
            code = pool_ave_code(v00, v01, v10, v11, hx, hy);
        }
        y[i] = out;
        if (codes) codes[i] = (unsigned char)code;
    }
}

template <int MODE, bool MASK>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float *__restrict__ dy,
                                                       const float *__restrict__ x, int C, int H,
                                                       int W, int Ho, int Wo,
                                                       float *__restrict__ dx) {
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % Wo;
        const int oy = (i / Wo) % Ho;
        const int c = i / ((size_t)Wo * Ho);
        const int y0 = 2 * oy, x0 = 2 * ox;
        const size_t base = ((size_t)c * H + y0) * W + x0;
        const bool hx = x0 + 1 < W, hy = y0 + 1 < H;
        const float g = dy[i];
        const float v00 = x[base];
        const float v01 = hx ? x[base + 1] : 0.f;
        const float v10 = hy ? x[base + W] : 0.f;
        const float v11 = (hx && hy) ? x[base + W + 1] : 0.f;
        float g00, g01, g10, g11;
        if (MODE == STX_POOL_MAX) {
            int arg = 0;
            float best = v00;
            if (hx && v01 > best) { best = v01; arg = 1; }
            if (hy && v10 > best) { best = v10; arg = 2; }
            if (hx && hy && v11 > best) { best = v11; arg = 3; }
            g00 = arg == 0 ? g : 0.f;
            g01 = arg == 1 ? g : 0.f;
            g10 = arg == 2 ? g : 0.f;
            g11 = arg == 3 ? g : 0.f;
        } else {
            const float cnt = (hx ? 2.f : 1.f) * (hy ? 2.f : 1.f);
            g00 = g01 = g10 = g11 = g / cnt;
        }
        if (MASK) {
            g00 = v00 > 0.f ? g00 : 0.f;
            g01 = v01 > 0.f ? g01 : 0.f;
            g10 = v10 > 0.f ? g10 : 0.f;
            g11 = v11 > 0.f ? g11 : 0.f;
        }
        dx[base] = g00;
        if (hx) dx[base + 1] = g01;
        if (hy) dx[base + W] = g10;
        if (hx && hy) dx[base + W + 1] = g11;
    }
}

// The same backward pass from the window codes of the forward pass (no pool input).
template <int MODE, bool MASK>
__global__ __launch_bounds__(256) void pool_bwd_codes_kernel(const float *__restrict__ dy,
                                                             const unsigned char *__restrict__ codes,
                                                             int C, int H, int W, int Ho, int Wo,
                                                             float *__restrict__ dx) {
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % Wo;
        const int oy = (i / Wo) % Ho;
        const int c = i / ((size_t)Wo * Ho);
        const int y0 = 2 * oy, x0 = 2 * ox;
        const size_t base = ((size_t)c * H + y0) * W + x0;
        const bool hx = x0 + 1 < W, hy = y0 + 1 < H;
        const float g = dy[i];
        const unsigned code = codes[i];
        float g00, g01, g10, g11;
        if (MODE == STX_POOL_MAX) {
            const unsigned arg = code & 3u;
            const float gs = (!MASK || (code & 4u)) ? g : 0.f;
            g00 = arg == 0 ? gs : 0.f;
            g01 = arg == 1 ? gs : 0.f;
            g10 = arg == 2 ? gs : 0.f;
            g11 = arg == 3 ? gs : 0.f;
        } else {
            const float cnt = (hx ? 2.f : 1.f) * (hy ? 2.f : 1.f);
            const float gq = g / cnt;
            g00 = (!MASK || (code & 1u)) ? gq : 0.f;
            g01 = (!MASK || (code & 2u)) ? gq : 0.f;
            g10 = (!MASK || (code & 4u)) ? gq : 0.f;
            g11 = (!MASK || (code & 8u)) ? gq : 0.f;
        }
        if (hx && ((W & 1) == 0)) {       // even widths: the two columns are one aligned store
            *reinterpret_cast<float2 *>(dx + base) = make_float2(g00, g01);
            if (hy) *reinterpret_cast<float2 *>(dx + base + W) = make_float2(g10, g11);
        } else {
            dx[base] = g00;
            if (hx) dx[base + 1] = g01;
            if (hy) dx[base + W] = g10;
            if (hx && hy) dx[base + W + 1] = g11;
        }
    }
}

static int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 256 * 16); }

int pool_forward_launch(hipStream_t s, const float *x, int C, int H, int W, int mode, float *y,
                        unsigned char *codes) {
    const int Ho = pooled_len(H), Wo = pooled_len(W);
    const size_t total = (size_t)C * Ho * Wo;
    if (mode == STX_POOL_MAX)
        pool_fwd_kernel<STX_POOL_MAX><<<grid_for(total), 256, 0, s>>>(x, C, H, W, Ho, Wo, y, codes);
    else
        pool_fwd_kernel<STX_POOL_AVE><<<grid_for(total), 256, 0, s>>>(x, C, H, W, Ho, Wo, y, codes);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int pool_backward_launch(hipStream_t s, const float *dy, const float *x, int C, int H, int W,
                         int mode, bool relu_mask, float *dx) {
    const int Ho = pooled_len(H), Wo = pooled_len(W);
    const size_t total = (size_t)C * Ho * Wo;
    const int g = grid_for(total);
    if (mode == STX_POOL_MAX) {
        if (relu_mask)
            pool_bwd_kernel<STX_POOL_MAX, true><<<g, 256, 0, s>>>(dy, x, C, H, W, Ho, Wo, dx);
        else
            pool_bwd_kernel<STX_POOL_MAX, false><<<g, 256, 0, s>>>(dy, x, C, H, W, Ho, Wo, dx);
    } else {
        if (relu_mask)
            pool_bwd_kernel<STX_POOL_AVE, true><<<g, 256, 0, s>>>(dy, x, C, H, W, Ho, Wo, dx);
        else
            pool_bwd_kernel<STX_POOL_AVE, false><<<g, 256, 0, s>>>(dy, x, C, H, W, Ho, Wo, dx);
    }
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int pool_backward_codes_launch(hipStream_t s, const float *dy, const unsigned char *codes, int C,
                               int H, int W, int mode, bool relu_mask, float *dx) {
    const int Ho = pooled_len(H), Wo = pooled_len(W);
    const size_t total = (size_t)C * Ho * Wo;
    const int g = grid_for(total);
    if (mode == STX_POOL_MAX) {
        if (relu_mask)
            pool_bwd_codes_kernel<STX_POOL_MAX, true><<<g, 256, 0, s>>>(dy, codes, C, H, W, Ho, Wo, dx);
        else
            pool_bwd_codes_kernel<STX_POOL_MAX, false><<<g, 256, 0, s>>>(dy, codes, C, H, W, Ho, Wo, dx);
    } else {
        if (relu_mask)
            pool_bwd_codes_kernel<STX_POOL_AVE, true><<<g, 256, 0, s>>>(dy, codes, C, H, W, Ho, Wo, dx);
        else
            pool_bwd_codes_kernel<STX_POOL_AVE, false><<<g, 256, 0, s>>>(dy, codes, C, H, W, Ho, Wo, dx);
    }
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
