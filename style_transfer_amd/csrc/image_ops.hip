// Full-image kernels: tile cut / put with the seam-suppression shift as an index offset, the
// TV / p-norm / auxiliary regularizers, the fused Adam step, the BLAS-1 pieces of L-BFGS, the
// per-step statistics and the float -> uint8 conversion.  All are single-pass and HBM-bound;
// reductions produce per-workgroup partials that a second tiny kernel adds in a fixed order.
//
// Reference: style_transfer.py:700-736 (eval_loss_and_grad), 777-815 (roll, statistics),
// 378-386 (get_image); num_utils.py:74-82 (p_norm), 136-140 (roll2), 150-162 (tv_norm);
// optimizers.py:26-42 (Adam), 74-121 (L-BFGS vector algebra).
// This file is compiled with -ffp-contract=off so that elementwise float32 arithmetic rounds
// like the reference's numpy expressions (one rounding per operation).

#include <algorithm>

#include "common.h"

namespace stx {

constexpr int kBlocks = 1024;   // upper bound on reduction workgroups (scratch is sized for it)

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block-reduces up to NV values and writes partials[v * gridDim.x + blockIdx.x].
template <int NV>
__device__ __forceinline__ void block_partials(float (&v)[NV], float *partials) {
    __shared__ float red[NV][4];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float s = wave_sum_f(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            partials[k * gridDim.x + blockIdx.x] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    }
}

// (c, y, x) of flat index i of a [3][H][W] array, carried along a grid-stride loop with adds and
// compares: the three 64-bit divisions per element that `i % W`, `i / W % H`, `i / plane` cost were
// most of the regularizer and statistics kernels' instructions (93 and 50 us on a 2048^2 image).
// Same elements in the same order per thread: the sums are bit-identical to the division form's.
struct Pos3 {
    int x, y, c;
};
struct Stride3 {
    int sx, sy, sc;      // the loop stride gridDim.x * 256 as (columns, rows, planes)
};
__device__ __forceinline__ Pos3 pos3_at(size_t i, int H, int W) {
    const size_t row = i / (size_t)W;
    return Pos3{(int)(i - row * (size_t)W), (int)(row % (size_t)H), (int)(row / (size_t)H)};
}
__device__ __forceinline__ Stride3 stride3_of(size_t stride, int H, int W) {
    const size_t rows = stride / (size_t)W;
    return Stride3{(int)(stride - rows * (size_t)W), (int)(rows % (size_t)H), (int)(rows / (size_t)H)};
}
__device__ __forceinline__ void pos3_advance(Pos3 &p, const Stride3 &s, int H, int W) {
    p.x += s.sx;
    const int cx = p.x >= W ? 1 : 0;
    p.x -= cx ? W : 0;
    p.y += s.sy + cx;
    const int cy = p.y >= H ? 1 : 0;
    p.y -= cy ? H : 0;
    p.c += s.sc + cy;
}

// out[k] = sum over n partials of value k, accumulated in double in a fixed order.
template <int NV>
__global__ void finish_partials_kernel(const float *__restrict__ partials, int n,
                                       double *__restrict__ out) {
    __shared__ double red[256];
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[k * n + i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

static int blocks_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, kBlocks); }

// ------------------------------------------------------------------------------ cut / put ---
__device__ __forceinline__ int wrap(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// One workgroup row per (channel, tile row): the wrapped source row is a scalar, a lane's column
// one add and one conditional subtract (the flat-index form spent its time in 64-bit divisions:
// 15 us for a 12.6 MB tile).
template <bool PUT>
__global__ __launch_bounds__(256) void tile_move_kernel(float *__restrict__ full, int H, int W,
                                                        int rx, int ry, int y0, int x0, int th,
                                                        int tw, float *__restrict__ tile) {
    const int cy = blockIdx.y, c = cy / th, y = cy - c * th;
    const float *const frow_c = full + ((size_t)c * H + wrap(y0 + y - ry, H)) * W;
    float *const frow = const_cast<float *>(frow_c);
    float *const trow = tile + (size_t)cy * tw;
    const int xs = wrap(x0 - rx, W);                       // source column of the tile's column 0
    for (int x = blockIdx.x * 256 + threadIdx.x; x < tw; x += gridDim.x * 256) {
        int xx = xs + x % W;                               // (a tile is never wider than the image,
        xx = xx >= W ? xx - W : xx;                        //  but the arithmetic does not rely on it)
        if (PUT)
            frow[xx] = trow[x];
        else
            trow[x] = frow[xx];
    }
}

static dim3 tile_move_grid(int th, int tw) { return dim3((unsigned)std::min(8, (tw + 255) / 256), (unsigned)(3 * th)); }

// The same move four columns at a time (16-byte accesses): widths, the tile's first source column
// and both bases multiples of four floats -- which the seam-suppression shifts are (multiples of
// the deepest content layer's scale, 8 or 16: style_transfer.py:777-779) -- so a group of four
// never straddles the wrap.  Four rows per workgroup: a quarter of a million one-dword threads
// per 1024^2 tile made the copy launch-bound (13 us for 25 MB).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool PUT>
__global__ __launch_bounds__(256) void tile_move4_kernel(float *__restrict__ full, int H, int W,
                                                         int rx, int ry, int y0, int x0, int th,
                                                         int tw, float *__restrict__ tile) {
    const int tw4 = tw >> 2, per_row = (tw4 + 63) / 64 * 64;      // lanes per row, whole waves
    const int rows = 3 * th;
    const int xs = wrap(x0 - rx, W);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * per_row; e += gridDim.x * 256) {
        const int cy = e / per_row, x4 = e - cy * per_row;
        if (x4 >= tw4) continue;
        const int c = cy / th, y = cy - c * th;
        float *const frow = full + ((size_t)c * H + wrap(y0 + y - ry, H)) * W;
        float *const trow = tile + (size_t)cy * tw;
        int xx = xs + 4 * x4;
        xx = xx >= W ? xx - W : xx;
        if (PUT)
            *reinterpret_cast<f32x4_t *>(frow + xx) = *reinterpret_cast<const f32x4_t *>(trow + 4 * x4);
        else
            *reinterpret_cast<f32x4_t *>(trow + 4 * x4) = *reinterpret_cast<const f32x4_t *>(frow + xx);
    }
}

static bool tile_move_vec(const float *full, int W, int rx, int x0, int tw, const float *tile) {
    const int xs = ((x0 - rx) % W + W) % W;      // (host-side twin of wrap())
    return tw % 4 == 0 && W % 4 == 0 && tw <= W && xs % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(full) | reinterpret_cast<uintptr_t>(tile)) & 15) == 0;
}
static int tile_move4_blocks(int th, int tw) {
    const long lanes = 3L * th * (((tw >> 2) + 63) / 64 * 64);
    return (int)std::min<long>((lanes + 255) / 256, 4096);
}

int cut_tile_launch(hipStream_t s, const float *img, int H, int W, int rx, int ry, int y0, int x0,
                    int th, int tw, float *tile) {
    if (3 * (long)th > 65535) {
        set_error("cut_tile: tile of %d rows is too tall", th);
        return STX_ERR_UNSUPPORTED;
    }
    if (tile_move_vec(img, W, rx, x0, tw, tile))
        tile_move4_kernel<false><<<tile_move4_blocks(th, tw), 256, 0, s>>>(const_cast<float *>(img), H, W, rx,
                                                                            ry, y0, x0, th, tw, tile);
    else
        tile_move_kernel<false><<<tile_move_grid(th, tw), 256, 0, s>>>(const_cast<float *>(img), H, W, rx, ry,
                                                                      y0, x0, th, tw, tile);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int put_tile_launch(hipStream_t s, float *grad, int H, int W, int rx, int ry, int y0, int x0,
                    int th, int tw, const float *tile) {
    if (3 * (long)th > 65535) {
        set_error("put_tile: tile of %d rows is too tall", th);
        return STX_ERR_UNSUPPORTED;
    }
    if (tile_move_vec(grad, W, rx, x0, tw, tile))
        tile_move4_kernel<true><<<tile_move4_blocks(th, tw), 256, 0, s>>>(grad, H, W, rx, ry, y0, x0, th, tw,
                                                                           const_cast<float *>(tile));
    else
        tile_move_kernel<true><<<tile_move_grid(th, tw), 256, 0, s>>>(grad, H, W, rx, ry, y0, x0, th, tw,
                                                                     const_cast<float *>(tile));
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ------------------------------------------------------- feature-map stitching on the device ---
// dst[c][y0 + y][x0 + x] = src[c][y][x]: places one tile's feature map into the full-image map
// (the stitch of eval_features_once, style_transfer.py:457-461).
__global__ __launch_bounds__(256) void place_window_kernel(float *__restrict__ dst, int dh, int dw,
                                                           int y0, int x0,
                                                           const float *__restrict__ src, int C,
                                                           int h, int w) {
    const size_t total = (size_t)C * h * w;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = i % w;
        const int y = (i / w) % h;
        const int c = i / ((size_t)w * h);
        dst[((size_t)c * dh + y0 + y) * dw + x0 + x] = src[i];
    }
}

int place_window_launch(hipStream_t s, float *dst, int dh, int dw, int y0, int x0, const float *src,
                        int C, int h, int w) {
    const size_t total = (size_t)C * h * w;
    place_window_kernel<<<(int)std::min<size_t>((total + 255) / 256, 8192), 256, 0, s>>>(
        dst, dh, dw, y0, x0, src, C, h, w);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// acc (= | +=) alpha * roll2(src, (sx, sy)): one pass of prepare_features' averaging
// (style_transfer.py:479-485) without physically rolling the accumulator back and forth.
template <bool INIT>
__global__ __launch_bounds__(256) void roll_add_kernel(float *__restrict__ acc,
                                                       const float *__restrict__ src, int C, int h,
                                                       int w, int sx, int sy, float alpha) {
    const size_t total = (size_t)C * h * w;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = i % w;
        const int y = (i / w) % h;
        const int c = i / ((size_t)w * h);
        const float v = src[((size_t)c * h + wrap(y - sy, h)) * w + wrap(x - sx, w)];
        acc[i] = INIT ? v / alpha : alpha * v + acc[i];   // first pass divides (feats / passes)
    }
}

// init: acc = roll(src) / alpha;  otherwise acc += alpha * roll(src)
int roll_add_launch(hipStream_t s, float *acc, const float *src, int C, int h, int w, int sx, int sy,
                    float alpha, bool init) {
    const size_t total = (size_t)C * h * w;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    if (init)
        roll_add_kernel<true><<<blocks, 256, 0, s>>>(acc, src, C, h, w, sx, sy, alpha);
    else
        roll_add_kernel<false><<<blocks, 256, 0, s>>>(acc, src, C, h, w, sx, sy, alpha);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ------------------------------------------------------------------------------- resampling ---
// One separable pass of Pillow's 'F'-mode resampler (num_utils.resize, num_utils.py:90-108):
// out[c][y][xx] = float( sum_x double(in[c][y][xmin + x]) * k[xx][x] ) along the last axis
// (AXIS 0) or the row axis (AXIS 1); the windows and normalised weights come from the host.
template <int AXIS>
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ src, int C, int H,
                                                       int W, float *__restrict__ dst, int OH,
                                                       int OW, const int *__restrict__ bounds,
                                                       const double *__restrict__ k, int ksize,
                                                       int clamp) {
    const size_t total = (size_t)C * OH * OW;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = i % OW;
        const int oy = (i / OW) % OH;
        const int c = i / ((size_t)OW * OH);
        const int o = AXIS == 0 ? ox : oy;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const double *kk = k + (size_t)o * ksize;
        double acc = 0.0;
        if (AXIS == 0) {
            const float *row = src + ((size_t)c * H + oy) * W + first;
            for (int t = 0; t < n; ++t) acc += (double)row[t] * kk[t];
        } else {
            const float *col = src + ((size_t)c * H + first) * W + ox;
            for (int t = 0; t < n; ++t) acc += (double)col[(size_t)t * W] * kk[t];
        }
        float v = (float)acc;
        if (clamp) v = fmaxf(v, 0.f);
        dst[i] = v;
    }
}

int resample_launch(hipStream_t s, int axis, const float *src, int C, int H, int W, float *dst,
                    int OH, int OW, const int *bounds, const double *k, int ksize, int clamp) {
    const size_t total = (size_t)C * OH * OW;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    if (axis == 0)
        resample_kernel<0><<<blocks, 256, 0, s>>>(src, C, H, W, dst, OH, OW, bounds, k, ksize, clamp);
    else
        resample_kernel<1><<<blocks, 256, 0, s>>>(src, C, H, W, dst, OH, OW, bounds, k, ksize, clamp);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// --------------------------------------------------------------------------- regularizers ---
struct RegArgs {
    const float *img;
    float *grad;
    const float *aux;
    int H, W;
    float mean[3];
    float tv_scale, tv_half_beta, p_scale, p_power, aux_scale;
    int p_int;              // p_power - 1 when that is an integer in 1..8, else 0
    int aux_rx, aux_ry;     // the iteration's shift: the auxiliary image is NOT rolled with the image
};

__global__ __launch_bounds__(256) void regularizers_kernel(RegArgs a, float *__restrict__ partials) {
    const size_t plane = (size_t)a.H * a.W, total = 3 * plane;
    float sums[3] = {0.f, 0.f, 0.f};   // TV, P, AUX (unscaled)
    const size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x;
    Pos3 pos = pos3_at(i0, a.H, a.W);
    const Stride3 step = stride3_of((size_t)gridDim.x * 256, a.H, a.W);
    for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256, pos3_advance(pos, step, a.H, a.W)) {
        const int x = pos.x, y = pos.y, c = pos.c;
        const float *p = a.img + (size_t)c * plane;
        const float v = p[(size_t)y * a.W + x];
        float g = a.grad[i];   // each term is added like a separate saxpy (style_transfer.py:713-733)
        if (a.tv_scale != 0.f) {
            const int xp = x + 1 == a.W ? 0 : x + 1, xm = x == 0 ? a.W - 1 : x - 1;
            const int yp = y + 1 == a.H ? 0 : y + 1, ym = y == 0 ? a.H - 1 : y - 1;
            const float s = 127.5f;
            const float x00 = v / s;
            const float x0p = p[(size_t)y * a.W + xp] / s, xp0 = p[(size_t)yp * a.W + x] / s;
            const float x0m = p[(size_t)y * a.W + xm] / s, xm0 = p[(size_t)ym * a.W + x] / s;
            const float xpm = p[(size_t)yp * a.W + xm] / s, xmp = p[(size_t)ym * a.W + xp] / s;
            const float hb = a.tv_half_beta;
            // this pixel
            const float dx = x00 - x0p, dy = x00 - xp0;
            const float n2 = dx * dx + dy * dy + kEps;
            sums[0] += hb == 1.f ? n2 : powf(n2, hb);
            const float dn = hb == 1.f ? 1.f : hb * powf(n2, hb - 1.f);
            // left neighbour's x-difference and upper neighbour's y-difference point at this pixel
            const float dxl = x0m - x00, dyl = x0m - xpm;
            const float n2l = dxl * dxl + dyl * dyl + kEps;
            const float dnl = hb == 1.f ? 1.f : hb * powf(n2l, hb - 1.f);
            const float dxu = xm0 - xmp, dyu = xm0 - x00;
            const float n2u = dxu * dxu + dyu * dyu + kEps;
            const float dnu = hb == 1.f ? 1.f : hb * powf(n2u, hb - 1.f);
            const float gx = 2.f * dx * dn, gy = 2.f * dy * dn;
            float tv = gx + gy;
            tv -= 2.f * dxl * dnl;
            tv -= 2.f * dyu * dnu;
            g = a.tv_scale * tv + g;
        }
        if (a.p_scale != 0.f) {
            const float z = (v + a.mean[c] - 127.5f) / 127.5f;
            const float az = fabsf(z);
            // |z|^(p-1): small integer exponents (the default p = 6) by multiplication, a few ulp
            // like powf itself and a tenth of its instructions
            float ap1;
            if (a.p_int > 0) {
                ap1 = az;
                for (int e = 1; e < a.p_int; ++e) ap1 *= az;
            } else {
                ap1 = powf(az, a.p_power - 1.f);
            }
            sums[1] += ap1 * az;
            const float sg = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
            g = a.p_scale * (a.p_power * sg * ap1) + g;
        }
        if (a.aux != nullptr && a.aux_scale != 0.f) {
            // the reference rolls the image by the iteration's shift but never its auxiliary
            // image (style_transfer.py:729-733 against 777-786): in the un-rolled frame kept here,
            // pixel (y, x) meets aux[(y + ry) mod H][(x + rx) mod W]
            int ay = (y + a.aux_ry) % a.H, ax = (x + a.aux_rx) % a.W;
            if (ay < 0) ay += a.H;
            if (ax < 0) ax += a.W;
            const float d = (v - a.aux[(size_t)c * plane + (size_t)ay * a.W + ax]) / 127.5f;
            sums[2] += d * d;
            g = a.aux_scale * d + g;
        }
        a.grad[i] = g;
    }
    block_partials<3>(sums, partials);
}

int regularizers_launch(hipStream_t s, const float *img, float *grad, int H, int W,
                        const float mean[3], float tv_scale, float tv_power, float p_scale,
                        float p_power, const float *aux, float aux_scale, int aux_rx, int aux_ry,
                        double *loss_terms, float *scratch, size_t scratch_floats) {
    RegArgs a;
    a.img = img;
    a.grad = grad;
    a.aux = aux;
    a.H = H;
    a.W = W;
    for (int i = 0; i < 3; ++i) a.mean[i] = mean[i];
    a.tv_scale = tv_scale;
    a.tv_half_beta = tv_power / 2.f;
    a.p_scale = p_scale;
    a.p_power = p_power;
    const float pm1 = p_power - 1.f;
    a.p_int = (pm1 >= 1.f && pm1 <= 8.f && pm1 == (float)(int)pm1) ? (int)pm1 : 0;
    a.aux_scale = aux_scale;
    a.aux_rx = aux_rx;
    a.aux_ry = aux_ry;
    const int blocks = blocks_for((size_t)3 * H * W);
    if (scratch_floats < (size_t)3 * blocks) {
        set_error("regularizers: scratch too small");
        return STX_ERR_STATE;
    }
    regularizers_kernel<<<blocks, 256, 0, s>>>(a, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<3><<<1, 256, 0, s>>>(scratch, blocks, loss_terms);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ----------------------------------------------------------------------------------- SWT ---
// The SWT regularizer of the reference for its defaults (wavelet 'haar', one level;
// num_utils.py:179-196, style_transfer.py:716-720): every channel of img / 127.5 is padded
// symmetrically to a square of side N = 2^ceil(log2(max(H, W))) (an odd amount puts the extra row
// / column behind), transformed with the stationary Haar transform (periodic on the square), the
// approximation band is dropped and the rest transformed back: what comes out is
//     D = x - B x,   B = [1 2 1]/4 along rows x [1 2 1]/4 along columns, circular on the square
// (oracle/num_ops.py has the derivation and a band-by-band check).  loss = sum |D|^p over the
// H x W crop; the reference adds the p-norm's OWN gradient at D to the image gradient, not its
// chain through D, and so does this.  The reference transforms the image rolled by the
// iteration's shift; here the image stays un-rolled: pixel (y, x) sits at ((y + ry) mod H,
// (x + rx) mod W) of the rolled picture.
__device__ __forceinline__ int swt_source(int q, int pad_lo, int n) {
    // padded coordinate q in [0, N) -> coordinate of the rolled picture (numpy.pad 'symmetric')
    int t = (q - pad_lo) % (2 * n);
    if (t < 0) t += 2 * n;
    return t < n ? t : 2 * n - 1 - t;
}

__global__ __launch_bounds__(256) void swt_haar_kernel(const float *__restrict__ img,
                                                       float *__restrict__ grad, int H, int W, int N,
                                                       int rx, int ry, float scale, float power,
                                                       float *__restrict__ partials) {
    const size_t plane = (size_t)H * W, total = 3 * plane;
    const int pad_y = (N - H) / 2, pad_x = (N - W) / 2;
    float sums[1] = {0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = i % W;
        const int y = (i / W) % H;
        const int c = i / plane;
        const float *p = img + (size_t)c * plane;
        // position in the rolled picture and on the padded square
        int Y = (y + ry) % H, X = (x + rx) % W;
        if (Y < 0) Y += H;
        if (X < 0) X += W;
        const int qy = Y + pad_y, qx = X + pad_x;
        // rows first (vertical [1 2 1]/4), then columns, like the oracle's two passes
        int uy[3];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int sy = swt_source((qy + dy + N) % N, pad_y, H);
            uy[dy + 1] = (sy - ry) % H;
            if (uy[dy + 1] < 0) uy[dy + 1] += H;
        }
        float cols[3], centre = 0.f;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int sx = swt_source((qx + dx + N) % N, pad_x, W);
            int ux = (sx - rx) % W;
            if (ux < 0) ux += W;
            const float up = p[(size_t)uy[0] * W + ux] / 127.5f, mid = p[(size_t)uy[1] * W + ux] / 127.5f;
            const float down = p[(size_t)uy[2] * W + ux] / 127.5f;
            cols[dx + 1] = (up + 2.f * mid + down) / 4.f;
            if (dx == 0) centre = mid;
        }
        const float blur = (cols[0] + 2.f * cols[1] + cols[2]) / 4.f;
        const float d = centre - blur;
        const float ad = fabsf(d);
        float g;
        if (power == 2.f) {
            sums[0] += d * d;
            g = 2.f * d;
        } else if (power == 1.f) {
            sums[0] += ad;
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else {
            const float ap1 = powf(ad, power - 1.f);
            sums[0] += ap1 * ad;
            g = power * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * ap1;
        }
        grad[i] = scale * g + grad[i];
    }
    block_partials<1>(sums, partials);
}

int swt_haar_launch(hipStream_t s, const float *img, float *grad, int H, int W, int rx, int ry,
                    float scale, float power, double *loss_term, float *scratch,
                    size_t scratch_floats) {
    int N = 1;
    while (N < std::max(H, W)) N *= 2;
    const int blocks = blocks_for((size_t)3 * H * W);
    if (scratch_floats < (size_t)blocks) {
        set_error("swt_haar: scratch too small");
        return STX_ERR_STATE;
    }
    swt_haar_kernel<<<blocks, 256, 0, s>>>(img, grad, H, W, N, rx, ry, scale, power, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<1><<<1, 256, 0, s>>>(scratch, blocks, loss_term);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ----------------------------------------------------------------------------------- Adam ---
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ params,
                                                   const float *__restrict__ grad,
                                                   float *__restrict__ g1, float *__restrict__ g2,
                                                   float *__restrict__ p1, float *__restrict__ avg,
                                                   size_t n, float lr, float b1, float b2, float bp1,
                                                   float omb1, float omb2, float ombp1, float c1,
                                                   float c2, float cp) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = grad[i];
        const float m1 = g1[i] * b1 + omb1 * g;            // EWMA.update: v *= beta; v += (1-beta)*x
        const float m2 = g2[i] * b2 + omb2 * (g * g);
        g1[i] = m1;
        g2[i] = m2;
        const float step = (m1 / c1) / (sqrtf(m2 / c2) + kEps);   // optimizers.py:37
        const float p = params[i] + (-lr) * step;                  // saxpy(-step_size, step, params)
        params[i] = p;
        const float a = p1[i] * bp1 + ombp1 * p;
        p1[i] = a;
        avg[i] = a / cp;
    }
}

int adam_launch(hipStream_t s, float *params, const float *grad, float *g1, float *g2, float *p1,
                float *avg, size_t n, double lr, double b1, double b2, double bp1, double c1,
                double c2, double cp) {
    // every scalar is formed in double like the reference's Python floats and rounded to float32
    // once, exactly where numpy casts it (beta, 1 - beta, 1 - beta^t, -lr)
    adam_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(
        params, grad, g1, g2, p1, avg, n, (float)lr, (float)b1, (float)b2, (float)bp1,
        (float)(1.0 - b1), (float)(1.0 - b2), (float)(1.0 - bp1), (float)c1, (float)c2, (float)cp);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// --------------------------------------------------------------------------------- BLAS-1 ---
template <bool ABS>
__global__ __launch_bounds__(256) void dot_kernel(const float *__restrict__ x,
                                                  const float *__restrict__ y, size_t n,
                                                  float *__restrict__ partials) {
    float acc[1] = {0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc[0] += ABS ? fabsf(x[i]) : x[i] * y[i];
    block_partials<1>(acc, partials);
}

int dot_launch(hipStream_t s, const float *x, const float *y, size_t n, double *out_dev,
               float *scratch, size_t scratch_floats) {
    const int blocks = blocks_for(n);
    if (scratch_floats < (size_t)blocks) return STX_ERR_STATE;
    dot_kernel<false><<<blocks, 256, 0, s>>>(x, y, n, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<1><<<1, 256, 0, s>>>(scratch, blocks, out_dev);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int abs_sum_launch(hipStream_t s, const float *x, size_t n, double *out_dev, float *scratch,
                   size_t scratch_floats) {
    const int blocks = blocks_for(n);
    if (scratch_floats < (size_t)blocks) return STX_ERR_STATE;
    dot_kernel<true><<<blocks, 256, 0, s>>>(x, x, n, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<1><<<1, 256, 0, s>>>(scratch, blocks, out_dev);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float *__restrict__ x,
                                                   float *__restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = a * x[i] + y[i];
}

int axpy_launch(hipStream_t s, float a, const float *x, float *y, size_t n) {
    axpy_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(a, x, y, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

__global__ __launch_bounds__(256) void scale_kernel(float a, float *__restrict__ x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x[i] = a * x[i];
}

int scale_launch(hipStream_t s, float a, float *x, size_t n) {
    scale_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(a, x, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// Coefficients that live on the device (the L-BFGS two-loop recursion: every dot product feeds the
// next axpy, optimizers.py:105-121).  The coefficient is formed in double exactly as the host
// form does -- a / da * c1 + b / db * c2 -- and rounded to float once.
__global__ __launch_bounds__(256) void axpy_dev_kernel(double c1, const double *__restrict__ a,
                                                       double da, double c2,
                                                       const double *__restrict__ b, double db,
                                                       const float *__restrict__ x,
                                                       float *__restrict__ y, size_t n) {
    double coef = a[0] / da * c1;
    if (b) coef += b[0] / db * c2;
    const float f = (float)coef;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = f * x[i] + y[i];
}

int axpy_dev_launch(hipStream_t s, double c1, const double *a, double da, double c2, const double *b,
                    double db, const float *x, float *y, size_t n) {
    axpy_dev_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(c1, a, da, c2, b, db,
                                                                                 x, y, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// x *= c / (den[0] / den_div)
__global__ __launch_bounds__(256) void scale_dev_kernel(double c, const double *__restrict__ den,
                                                        double den_div, float *__restrict__ x,
                                                        size_t n) {
    const float f = (float)(c / (den[0] / den_div));
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x[i] = f * x[i];
}

int scale_dev_launch(hipStream_t s, double c, const double *den, double den_div, float *x, size_t n) {
    scale_dev_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(c, den, den_div, x, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// Fused passes of the L-BFGS step (round 5).  Each is, value for value, what the separate launches
// computed -- the same float operations in the same order per element (this file is compiled with
// -ffp-contract=off), the dot products accumulated by dot_kernel's threads in dot_kernel's order
// (same grid) and finished by the same kernel -- so a trajectory does not change by one bit; what
// changes is that an array is read once where it was read two or three times.
//   y = [g] (f x + src)  (src may be y; g = c_s / (den_s[0] / div_s) when den_s is given),
//   partial sums of <z, y>                                  [axpy_dev (+ scale_dev) + dot]
__global__ __launch_bounds__(256) void axpy_dot_dev_kernel(double c1, const double *__restrict__ a,
                                                           double da, double c2,
                                                           const double *__restrict__ b, double db,
                                                           double c_s, const double *__restrict__ den_s,
                                                           double div_s, const float *__restrict__ x,
                                                           const float *src, float *y,
                                                           const float *__restrict__ z, size_t n,
                                                           float *__restrict__ partials) {
    double coef = a[0] / da * c1;
    if (b) coef += b[0] / db * c2;
    const float f = (float)coef;
    const bool scaled = den_s != nullptr;
    const float g = scaled ? (float)(c_s / (den_s[0] / div_s)) : 1.f;
    float acc[1] = {0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = f * x[i] + src[i];
        if (scaled) v = g * v;
        y[i] = v;
        acc[0] += z[i] * v;
    }
    block_partials<1>(acc, partials);
}

int axpy_dot_dev_launch(hipStream_t s, double c1, const double *a, double da, double c2, const double *b,
                        double db, double c_s, const double *den_s, double div_s, const float *x,
                        const float *src, float *y, const float *z, size_t n, double *out_dev, float *scratch,
                        size_t scratch_floats) {
    const int blocks = blocks_for(n);
    if (scratch_floats < (size_t)blocks) return STX_ERR_STATE;
    axpy_dot_dev_kernel<<<blocks, 256, 0, s>>>(c1, a, da, c2, b, db, c_s, den_s, div_s, x, src, y, z, n, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<1><<<1, 256, 0, s>>>(scratch, blocks, out_dev);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

//   y = -g_old + g_new,  g_old = g_new,  partial sums of <s, y> and <y, y>
//                                               [copy + axpy(-1) + dot(s, y) + dot(y, y) + copy]
__global__ __launch_bounds__(256) void lbfgs_pair_kernel(const float *__restrict__ g_new,
                                                         float *__restrict__ g_old,
                                                         const float *__restrict__ sv,
                                                         float *__restrict__ y, size_t n,
                                                         float *__restrict__ partials) {
    float acc[2] = {0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gn = g_new[i];
        const float v = -1.0f * g_old[i] + gn;
        y[i] = v;
        g_old[i] = gn;
        acc[0] += sv[i] * v;
        acc[1] += v * v;
    }
    block_partials<2>(acc, partials);
}

int lbfgs_pair_launch(hipStream_t s, const float *g_new, float *g_old, const float *sv, float *y, size_t n,
                      double *out_dev2, float *scratch, size_t scratch_floats) {
    const int blocks = blocks_for(n);
    if (scratch_floats < (size_t)2 * blocks) return STX_ERR_STATE;
    lbfgs_pair_kernel<<<blocks, 256, 0, s>>>(g_new, g_old, sv, y, n, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<2><<<1, 256, 0, s>>>(scratch, blocks, out_dev2);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

//   s = c2 (c1 s),  params = 1 s + params                          [scale + scale + axpy]
__global__ __launch_bounds__(256) void scale2_axpy_kernel(float c1, float c2, float *__restrict__ sv,
                                                          float *__restrict__ params, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = c2 * (c1 * sv[i]);
        sv[i] = v;
        params[i] = 1.0f * v + params[i];
    }
}

int scale2_axpy_launch(hipStream_t s, float c1, float c2, float *sv, float *params, size_t n) {
    scale2_axpy_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(c1, c2, sv, params, n);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ------------------------------------------------------------------------------ statistics ---
__global__ __launch_bounds__(256) void step_stats_kernel(const float *__restrict__ avg,
                                                         float *__restrict__ old, int H, int W,
                                                         float *__restrict__ partials) {
    const size_t plane = (size_t)H * W, total = 3 * plane;
    float sums[2] = {0.f, 0.f};
    const size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x;
    Pos3 pos = pos3_at(i0, H, W);
    const Stride3 step = stride3_of((size_t)gridDim.x * 256, H, W);
    for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256, pos3_advance(pos, step, H, W)) {
        const int x = pos.x, y = pos.y;
        const size_t base = i - (size_t)y * W - x;
        const float v = avg[i];
        sums[0] += fabsf(v - old[i]);
        const float xd = v - avg[base + (size_t)y * W + (x + 1 == W ? 0 : x + 1)];
        const float yd = v - avg[base + (size_t)(y + 1 == H ? 0 : y + 1) * W + x];
        sums[1] += xd * xd + yd * yd;
        old[i] = v;
    }
    block_partials<2>(sums, partials);
}

int step_stats_launch(hipStream_t s, const float *avg, float *old, int H, int W, double *out_dev,
                      float *scratch, size_t scratch_floats) {
    const int blocks = blocks_for((size_t)3 * H * W);
    if (scratch_floats < (size_t)2 * blocks) return STX_ERR_STATE;
    step_stats_kernel<<<blocks, 256, 0, s>>>(avg, old, H, W, scratch);
    STX_CHECK_LAUNCH();
    finish_partials_kernel<2><<<1, 256, 0, s>>>(scratch, blocks, out_dev);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

// ----------------------------------------------------------------------------------- uint8 ---
__global__ __launch_bounds__(256) void to_u8_kernel(const float *__restrict__ img, int H, int W,
                                                    float m0, float m1, float m2,
                                                    uint8_t *__restrict__ out) {
    const size_t plane = (size_t)H * W;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
        const float b = fminf(fmaxf(img[i] + m0, 0.f), 255.f);
        const float g = fminf(fmaxf(img[plane + i] + m1, 0.f), 255.f);
        const float r = fminf(fmaxf(img[2 * plane + i] + m2, 0.f), 255.f);
        out[3 * i + 0] = (uint8_t)r;    // np.uint8() truncates toward zero
        out[3 * i + 1] = (uint8_t)g;
        out[3 * i + 2] = (uint8_t)b;
    }
}

int to_u8_launch(hipStream_t s, const float *img, int H, int W, const float mean[3], uint8_t *out) {
    const size_t plane = (size_t)H * W;
    to_u8_kernel<<<(int)std::min<size_t>((plane + 255) / 256, 8192), 256, 0, s>>>(
        img, H, W, mean[0], mean[1], mean[2], out);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
