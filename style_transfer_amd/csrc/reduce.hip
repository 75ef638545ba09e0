// Loss reductions and gradient injection for the tapped layers of a tile (HBM-bound passes).
//
// Reference: CaffeModel.eval_sc_grad_tile, style_transfer.py:575-593 with num_utils.py
// normalize (85-87), norm2 (69-71), saxpy (26-29):
//   content:  c = F - Fc[:, window];  loss += lw*cw * 1/2 sum c^2;  diff += lw*cw * c / (mean|c| + EPS)
//   style:    S = sym(tril(G - Gs)) F;                              diff += lw*sw/n * S / (mean|S| + EPS)
// The content residual is never materialised: one pass reduces sum c^2 and sum |c|, the
// injection pass recomputes F - Fc.  The content map Fc is the FULL-image map; the worker's
// physical roll of it (style_transfer.py:234,647-655) is applied here as an index offset with
// wrap-around.  All reductions write per-workgroup partials that are then added in a fixed
// order, so results do not depend on scheduling.

#include <algorithm>

#include "common.h"

namespace stx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int kRedBlocks = 1024;

__global__ __launch_bounds__(256) void content_sums_kernel(const float *__restrict__ feat,
                                                           const float *__restrict__ content,
                                                           ContentWindow w,
                                                           float *__restrict__ partials) {
    __shared__ float red[2][4];
    // One wave per 64-column segment of a feature row: the wrapped row / first column of the
    // content map are wave-uniform and computed once per segment (per ELEMENT, with its 64-bit
    // divisions, this kernel took 38 us for the 33 MB of conv4_2 where the two streams it reads
    // need 13).  Segments are dealt out in a fixed order: the sums do not depend on timing.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int segs = (w.fw + 63) >> 6;
    const int total_segs = w.C * w.fh * segs;
    const int origin_y = content_origin_y(w);
    int x_first = content_origin_x(w) % w.cw;
    if (x_first < 0) x_first += w.cw;
    // four segments per trip: eight loads in flight per lane (with one segment per trip the kernel
    // was bound by load latency: 2 MB in flight on the whole chip)
    float sq4[4] = {0.f, 0.f, 0.f, 0.f}, ab4[4] = {0.f, 0.f, 0.f, 0.f};
    const int step = gridDim.x * 4;
    for (int g0 = blockIdx.x * 4 + wave; g0 < total_segs; g0 += 4 * step) {
        float f[4], t[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * step;
            const int gg = g < total_segs ? g : g0;
            const int row = gg / segs, x0 = (gg - row * segs) * 64;
            const int c = row / w.fh, y = row - c * w.fh;
            int yy = (origin_y + y) % w.ch;
            if (yy < 0) yy += w.ch;
            const int x = x0 + lane;
            ok[u] = g < total_segs && x < w.fw;
            const int xc = x < w.fw ? x : 0;
            const int xx = (x_first + xc) % w.cw;
            f[u] = feat[((size_t)c * w.fh + y) * w.fw + xc];
            t[u] = content ? content[((size_t)c * w.ch + yy) * w.cw + xx] : 0.f;   // null map: Deep Dream
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = ok[u] ? f[u] - t[u] : 0.f;
            sq4[u] += d * d;
            ab4[u] += fabsf(d);
        }
    }
    float sq = (sq4[0] + sq4[1]) + (sq4[2] + sq4[3]), ab = (ab4[0] + ab4[1]) + (ab4[2] + ab4[3]);
    sq = wave_sum(sq);
    ab = wave_sum(ab);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sq;
        red[1][threadIdx.x >> 6] = ab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partials[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void sum_two_kernel(const float *__restrict__ partials, int n, float *__restrict__ out) {
    __shared__ float red[256];
    for (int which = 0; which < 2; ++which) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += partials[which * n + i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[which] = red[0];
        __syncthreads();
    }
}

// `sums` doubles as scratch: sums[0..1] receive the results, the partials live behind them
// (the engine reserves 2 + 2*kRedBlocks floats per content tap).
int content_sums_launch(hipStream_t s, const float *feat, const float *content,
                        const ContentWindow &win, float *sums, std::vector<SumJob> *defer) {
    const size_t total = (size_t)win.C * win.fh * win.fw;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, kRedBlocks);
    float *partials = sums + 2;
    content_sums_kernel<<<blocks, 256, 0, s>>>(feat, content, win, partials);
    STX_CHECK_LAUNCH();
    if (defer) {        // (sum_jobs_kernel adds in sum_two_kernel's order)
        defer->push_back(SumJob{partials, blocks, sums});
        defer->push_back(SumJob{partials + blocks, blocks, sums + 1});
        return STX_OK;
    }
    sum_two_kernel<<<1, 256, 0, s>>>(partials, blocks, sums);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// max |value written| of a launch into kAmaxSlots words of float bits (ConvProblem::y_amax; conv_h2.hip
// reads it as the scale of its input): one atomic per block
__device__ __forceinline__ void block_amax(float amax, unsigned *y_amax) {
    if (!y_amax) return;            // (uniform)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(y_amax + (blockIdx.x & (kAmaxSlots - 1)),
                  __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

template <bool ACC>
__global__ __launch_bounds__(256) void inject_style_kernel(float *__restrict__ diff,
                                                           const float *__restrict__ sgrad, size_t n,
                                                           const float *__restrict__ abs_sum,
                                                           float coef, unsigned *y_amax) {
    // normalize(): x *= 1 / (sum|x| / size + EPS)   (num_utils.py:85-87)
    const float scale = coef * (1.0f / (abs_sum[0] / (float)n + kEps));
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = scale * sgrad[i];
        const float o = ACC ? diff[i] + v : v;
        diff[i] = o;
        amax = fmaxf(amax, fabsf(o));
    }
    block_amax(amax, y_amax);
}

int inject_style_launch(hipStream_t s, float *diff, const float *sgrad, size_t n,
                        const float *abs_sum, float coef, bool accumulate, unsigned *y_amax) {
    // (with y_amax every block ends in an atomic: fewer, longer blocks)
    const int blocks = (int)std::min<size_t>((n + 255) / 256, y_amax ? 512 : 256 * 16);
    if (accumulate)
        inject_style_kernel<true><<<blocks, 256, 0, s>>>(diff, sgrad, n, abs_sum, coef, y_amax);
    else
        inject_style_kernel<false><<<blocks, 256, 0, s>>>(diff, sgrad, n, abs_sum, coef, y_amax);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

template <bool ACC>
__global__ __launch_bounds__(256) void inject_content_kernel(float *__restrict__ diff,
                                                             const float *__restrict__ feat,
                                                             const float *__restrict__ content,
                                                             ContentWindow w,
                                                             const float *__restrict__ sums,
                                                             float coef, unsigned *y_amax) {
    const size_t total = (size_t)w.C * w.fh * w.fw;
    const float scale = coef * (1.0f / (sums[1] / (float)total + kEps));
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = i % w.fw;
        const int y = (i / w.fw) % w.fh;
        const int c = i / ((size_t)w.fw * w.fh);
        const float v = scale * (feat[i] - (content ? content[content_index(w, c, y, x)] : 0.f));
        const float o = ACC ? diff[i] + v : v;
        diff[i] = o;
        amax = fmaxf(amax, fabsf(o));
    }
    block_amax(amax, y_amax);
}

int inject_content_launch(hipStream_t s, float *diff, const float *feat, const float *content,
                          const ContentWindow &win, const float *sums, float coef,
                          bool accumulate, unsigned *y_amax) {
    const size_t total = (size_t)win.C * win.fh * win.fw;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, y_amax ? 512 : 256 * 16);
    if (accumulate)
        inject_content_kernel<true><<<blocks, 256, 0, s>>>(diff, feat, content, win, sums, coef, y_amax);
    else
        inject_content_kernel<false><<<blocks, 256, 0, s>>>(diff, feat, content, win, sums, coef, y_amax);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

__global__ __launch_bounds__(256) void relu_kernel(float *__restrict__ x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x[i] = fmaxf(x[i], 0.f);
}

int relu_inplace_launch(hipStream_t s, float *x, size_t n) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
    relu_kernel<<<blocks, 256, 0, s>>>(x, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
