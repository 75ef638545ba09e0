// Internal declarations shared by the libstx translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "stx.h"

namespace stx {

// float32 machine epsilon: num_utils.py:14 (EPS), used by normalize / tv_norm / Adam.
constexpr float kEps = 1.1920928955078125e-07f;

void set_error(const char *fmt, ...);

// The library's STX_* switches (A/B levers and test hooks: INTEGRATION.md lists them).  They are read from a
// SNAPSHOT of the process environment -- taken when the library is first asked for one and again at every
// stx_reread_env() -- never with getenv() at a call site: the launch paths run on several host threads (one per
// engine of a farm), and getenv() beside a setenv() of the host program is a data race.  A caller that changes a
// switch inside a process (the tests do, bench.py's fp32 leg does) calls stx_reread_env() afterwards.  Returns
// the value or null, like getenv(); the pointer stays valid for the life of the process.
const char *sw_env(const char *name);
void sw_reread();

#define STX_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t err__ = (call);                                                          \
        if (err__ != hipSuccess) {                                                          \
            ::stx::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,            \
                             hipGetErrorString(err__));                                     \
            return STX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define STX_CHECK_LAUNCH()                                                                  \
    do {                                                                                    \
        hipError_t err__ = hipGetLastError();                                               \
        if (err__ != hipSuccess) {                                                          \
            ::stx::set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__,        \
                             hipGetErrorString(err__));                                     \
            return STX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define STX_TRY(expr)                                                                       \
    do {                                                                                    \
        int rc__ = (expr);                                                                  \
        if (rc__ != STX_OK) return rc__;                                                    \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int pooled_len(int n) { return n >= 2 ? (n - 2 + 1) / 2 + 1 : 1; }  // ceil((n-2)/2)+1
#ifdef __HIPCC__
// Window codes of the 2x2/2 pooling (pool.hip): what the backward pass needs to know about a window.
// MAX: index of the FIRST maximum in row-major order (strict '>' scan, as Caffe's) | 4 if it is > 0.
__device__ __forceinline__ unsigned pool_max_code(float v00, float v01, float v10, float v11, bool hx,
                                                  bool hy) {
    unsigned arg = 0;
    float best = v00;
    if (hx && v01 > best) best = v01, arg = 1;
    if (hy && v10 > best) best = v10, arg = 2;
    if (hx && hy && v11 > best) best = v11, arg = 3;
    return arg | (best > 0.f ? 4u : 0u);
}
// AVE: bit k set if element k of the window is > 0 (the ReLU mask of the blob below).
__device__ __forceinline__ unsigned pool_ave_code(float v00, float v01, float v10, float v11, bool hx,
                                                  bool hy) {
    return (v00 > 0.f ? 1u : 0u) | (hx && v01 > 0.f ? 2u : 0u) | (hy && v10 > 0.f ? 4u : 0u) |
           (hx && hy && v11 > 0.f ? 8u : 0u);
}
#endif

// ------------------------------------------------------------------------------------------------
// Kernel launchers (each enqueues on `stream` and returns STX_OK / STX_ERR_*).
// ------------------------------------------------------------------------------------------------

// Epilogue selection for the implicit-GEMM MFMA kernel.
enum ConvEpilogue {
    kEpiForward = 0,      // y = [relu](acc + bias)
    kEpiDgrad = 1,        // y = acc * (mask > 0)      (mask optional)
    kEpiSymm = 2,         // y = acc, plus per-workgroup sum|y| partials
    kEpiDgradInject = 3,  // kEpiDgrad plus the loss-gradient terms of the produced blob (internal)
    kEpiPartial = 4       // raw accumulators of one K slice (split-K, internal)
};

struct ContentWindow {
    int C, fh, fw;          // tile feature size
    int ch, cw;             // full content map size
    int oy, ox;             // window origin in the rolled map
    int sy, sx;             // roll shifts (rows, cols)
};

// Loss-gradient terms added by the backward-data epilogue to the gradient it produces (the
// reference's saxpy(lw*w, normalize(x), diff[layer]), style_transfer.py:580,592-593):
//   y += c_coef / (c_sums[1]/n + EPS) * (feat - content[window])     (content term, first)
//   y += s_coef / (s_abs_sum[0]/n + EPS) * sgrad                     (style term)
struct ConvInject {
    const float *sgrad = nullptr, *s_abs_sum = nullptr;
    float s_coef = 0.f;
    const float *content = nullptr, *c_sums = nullptr, *feat = nullptr;
    float c_coef = 0.f;
    ContentWindow win{};
};

struct ConvProblem {
    const float *x;        // [K][H][W] input planes (activations or upstream gradient)
    const float *w;        // weight rows, see conv_mfma.hip (packed tiles or a dense symmetric matrix)
    float *y;              // [M][H][W]
    const float *bias;     // kEpiForward: [M] or null
    const float *mask;     // kEpiDgrad: [M][H][W] post-ReLU data of the output blob, or null
    float *partials;       // kEpiSymm: one float per workgroup
    int K, M, H, W;        // reduction channels, output channels, plane size
    int ksize;             // 3 (pad 1) or 1 (pad 0)
    int relu;              // kEpiForward
    int epilogue;
    ConvInject inject;     // kEpiDgrad, optional
    unsigned char *pool_codes = nullptr;   // with pool_out: one window code per pooled element (pool.hip)
    // ReLU sign nibbles of a [C][H][W] blob, one byte per 2x2 window ([C][ceil(H/2)][ceil(W/2)],
    // bit 2*dy+dx = element (2y+dy, 2x+dx) > 0).  kEpiForward: in_codes (optional) receives the
    // nibbles of the INPUT planes x -- the forward pass of the layer that consumes a rectified blob
    // writes them as a by-product of staging its patches; kEpiDgrad: mask_codes (optional) replaces
    // the fp32 `mask` array, 1 byte instead of 16 per lane and channel.  Kernels that cannot use
    // them ignore them (conv_uses_relu_codes).
    unsigned char *in_codes = nullptr;
    const unsigned char *mask_codes = nullptr;
    // kEpiForward with relu: out_codes (optional) receives the nibbles of the OUTPUT planes y -- the
    // producer's epilogue holds exactly one 2x2 window per lane and channel (round 5: the fp16-split
    // kernel and the eight-wave fp32 kernel; unsplit launches only: conv_writes_out_codes)
    unsigned char *out_codes = nullptr;
    // the caller attaches ReLU nibbles to this launch wherever the kernel takes them (it does not under
    // STX_WINO_BIG=1, for one): launches that want them keep out of the tail split either way, so
    // that the schedule -- and with it the rounding -- does not depend on whether they were taken
    bool wants_codes = false;
    // kEpiForward with a fused pooling that also writes window codes: nobody will read y itself
    // (the pooled blob feeds the next layer, the backward pooling runs from the codes) -- skip
    // its stores.  Only honoured by the kernel that fuses (wino2_launch); ignored otherwise.
    bool skip_y = false;
    float *pool_out = nullptr;       // kEpiForward: also write the 2x2/2 ceil-mode pooling of y here
    int pool_mode = 0;               // (kernels that cannot do it leave it to the caller: see
                                     // wino2_fuses_pool)
    float *splitk_ws = nullptr;      // scratch for split-K partial sums (optional)
    size_t splitk_ws_floats = 0;
    // stx_clock_marks: one workgroup of the launch (the eight-wave Winograd kernel only) stores
    // {core-clock cycles, 100 MHz ticks} spent in its chunk loop here -- the shader clock the part
    // sustains INSIDE the kernel that does 80 % of the work (null: nothing is read or stored)
    long long *clock_out = nullptr;
    // conv_h2.hip (fp16 two-piece split): max |x| of the input planes as kAmaxSlots words of float
    // bits (the largest one counts), left by the kernel that wrote them or by absmax_launch; y_amax
    // (optional, zeroed by the caller): where this launch leaves max |y| of its own output
    const unsigned *x_amax = nullptr;
    unsigned *y_amax = nullptr;
    // kEpiDgrad, conv_h2.hip only (h2_takes_pooled_input): the upstream gradient arrives as the gradient
    // of a 2x2/2 pooling layer's OUTPUT -- x is [K][ceil(H/2)][ceil(W/2)] -- together with that layer's
    // window codes (pool.hip; the array must be readable 3 bytes past its end).  The patch staging
    // routes / spreads it exactly as pool_bwd_codes_kernel would have: the gradient of the pooling
    // layer's input (four times the size) is never written or read.
    const unsigned char *pin_codes = nullptr;
    int pin_mode = 0;            // STX_POOL_MAX / STX_POOL_AVE
    bool pin_mask = false;       // the blob under the pooling layer is rectified (the codes carry its signs)
#ifdef STX_EXPERIMENT_BF3
    const void *x_split = nullptr;   // tools/experiments/conv_bf3.hip only
#endif
};

// Tile configuration chosen for a problem; weights must be packed for the same (bm, kc).
struct ConvConfig {
    int id;        // index into the instantiation table
    int bm;        // output channels per workgroup
    int kc;        // reduction channels per LDS stage
    int pr, pc;    // pixel tile rows x cols
    int threads;
    size_t lds_bytes;
};

ConvConfig conv_pick_config(int ksize, int K, int M, int H, int W);
ConvConfig conv_config_by_id(int id);
int conv_num_workgroups(const ConvConfig &cfg, int M, int H, int W);
// Packed weight buffer size (floats) for a [M][K][ks][ks] filter bank under cfg.
size_t conv_packed_floats(const ConvConfig &cfg, int K, int M, int ksize);
// Packs Caffe-layout weights w[Mo][Ko][ks][ks] (device) into the kernel's tile layout.
// transpose_flip = 0: forward (M = Mo, K = Ko).  1: backward-data (M = Ko, K = Mo, taps rotated 180).
int conv_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int ksize,
                      int transpose_flip, const ConvConfig &cfg, float *packed);
int conv_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, bool packed_weights);
// Number of K slices conv_launch will use for this problem (1 = no split) and the scratch floats
// that requires.  Small planes yield fewer workgroups than the chip has CUs; slicing the
// reduction over several workgroups and adding the slices in a fixed order fills the machine.
int conv_splitk_factor(const ConvConfig &cfg, const ConvProblem &p, bool packed_weights);
size_t conv_splitk_floats(const ConvConfig &cfg, const ConvProblem &p, bool packed_weights);

int splitk_reduce_launch(hipStream_t s, const ConvProblem &p, int ksplit);
// The same for the work items item_base .. item_base + items - 1 of a 2-D Winograd launch only (64
// channels x one pr x pc pixel patch each, in the kernel's own item order): conv_wino2's tail split.
int splitk_reduce_items_launch(hipStream_t s, const ConvProblem &p, const ConvConfig &cfg, int slices,
                               int item_base, int items);
// Work item lt of a 2-D Winograd launch -> (pixel patch, channel tile); conv_wino2.hip's order.
__host__ __device__ inline void wino2_item_tiles(int lt, int m_tiles, int n_patches, int &pt, int &mt) {
    pt = lt / m_tiles;
    mt = lt - pt * m_tiles;
    if (m_tiles == 8 && (n_patches & 7) == 0) {
        const int g = lt >> 5, r = lt & 31;
        mt = (g & 1) * 4 + (r & 3);
        pt = (g >> 1) * 8 + (r >> 2);
    }
}

// Kernel arguments shared by the Winograd kernels.
struct WinoArgs {
    const float *x;
    const float *w;        // transformed filter bank in the kernel's own tile layout
    float *y;
    const float *bias;
    const float *mask;
    int K, M, H, W;
    int n_chunks, tiles_x, tiles_y, m_tiles, ksplit;
    int w_tile_stride;     // floats between consecutive output-channel tiles
    int x_bytes, w_bytes;
    int relu;
    ConvInject inj;
    float *pool_out;       // forward only: 2x2/2 pooling of the output, or null
    int pool_mode;
    unsigned char *pool_codes;   // with pool_out: window codes for the backward pass, or null
    int skip_y = 0;                             // forward + fused pooling with codes: y is not stored
    unsigned char *out_codes = nullptr;         // forward: ReLU nibbles of y to write (ConvProblem)
    unsigned char *in_codes = nullptr;          // forward: ReLU nibbles of x to write (ConvProblem)
    const unsigned char *mask_codes = nullptr;  // backward: ReLU nibbles of the output blob to read
    long long *clock_out = nullptr;             // ConvProblem::clock_out
    int item_base = 0;                          // conv_wino2, K slices: first work item of the launch (tail split)
    const unsigned *x_amax = nullptr;           // conv_h2: ConvProblem::x_amax / y_amax
    unsigned *y_amax = nullptr;
    const unsigned char *pin_codes = nullptr;   // conv_h2: ConvProblem::pin_codes / pin_mode / pin_mask
    int pin_mode = 0, pin_mask = 0;
#ifdef STX_EXPERIMENT_BF3
    int vp_rows = 0, vp_tp = 0;                 // tools/experiments/conv_bf3.hip only
#endif
};

// The Winograd configurations (cfg.id 200-202: conv_wino2.hip, fp32 2-D F(2x2,3x3); 300-302: conv_h2.hip,
// fp16-split 1-D F(2,3)): these three dispatch on cfg.id (conv_wino2.hip).
size_t wino_packed_floats(const ConvConfig &cfg, int K, int M);
int wino_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                      const ConvConfig &cfg, float *packed);
int wino_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit);


// First layer (at most 3 planes -> 64 channels, 3x3) with the Gram partials of its own output
// (conv_first.hip).  gram_partials (or null): conv_first_workgroups(H, W) partial tiles of 64 x 64
// floats, finished by gram_finish_launch with {C 64, HW, splits = that count, tiles 1, parts 1}.
bool conv_first_usable(int K, int M, int ksize);
int conv_first_workgroups(int H, int W);
int conv_first_launch(hipStream_t s, const float *x, const float *w_caffe, const float *bias, float *y,
                      int K, int H, int W, int relu, float *gram_partials, unsigned *y_amax = nullptr);

// 2-D Winograd F(2x2,3x3) variant (conv_wino2.hip); config id 200.
ConvConfig wino2_config(int geometry = 0);     // 0: 4 x 64 pixel patches, 1: 16 x 16, 2: 8 x 32
int wino2_pick_geometry(int H, int W);
size_t wino2_packed_floats(int K, int M);
int wino2_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                       float *packed);
int wino2_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit);
bool wino2_fuses_pool(const ConvProblem &p);
// True if a launch of p under cfg writes p.in_codes (forward) / reads p.mask_codes (backward).
bool conv_uses_relu_codes(const ConvConfig &cfg, const ConvProblem &p, int ksplit);
// True if a forward launch of p under cfg writes p.out_codes (an unsplit launch of the fp16-split or the
// eight-wave fp32 kernel; shape and epilogue only)
bool conv_writes_out_codes(const ConvConfig &cfg, const ConvProblem &p, int ksplit);
int wino2_splitk_factor(const ConvConfig &cfg, const ConvProblem &p);
// Tail split: a launch of n = 256 q + r work items (q >= 1) runs its last r items as r x slices K
// slices -- one short round instead of a mostly empty full one -- and a reduce pass over those r
// patches only.  {0, 1}: not for this problem.  Shape and epilogue only, like the K split.
struct Wino2Tail { int items, slices; };
Wino2Tail wino2_tail_split(const ConvConfig &cfg, const ConvProblem &p);
int wino2_max_slices(const ConvConfig &cfg, const ConvProblem &p);

// 1-D Winograd F(2,3) on the fp16 matrix cores with two-piece operands (conv_h2.hip): fp32-class
// accuracy at 0.28 of the fp32 2-D form's matrix time.  Config ids 300 (64 channels x 8 x 32 pixels per
// workgroup), 301 (128 channels) and 302 (64 channels x 16 x 32 pixels); all read the same packed bank.
constexpr int kAmaxSlots = 64;
ConvConfig h2_config(int mb, int pb = 1);     // (1, 1), (2, 1) or (1, 2): ids 300 / 301 / 302
ConvConfig h2_pick_config(const ConvProblem &p);    // the cheaper tiling by the round model (shape only)
bool h2_usable(const ConvProblem &p);       // what the kernel takes (shape, epilogue, addressing)
size_t h2_packed_floats(int K, int M);
int h2_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip, float *packed);
int h2_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit);
int h2_splitk_factor(const ConvConfig &cfg, const ConvProblem &p);
bool h2_fuses_pool(const ConvProblem &p);
bool h2_takes_pooled_input(const ConvConfig &cfg, const ConvProblem &p);   // ConvProblem::pin_codes
// slots[0 .. kAmaxSlots) = 0, then max |x| as float bits into them (the largest slot counts)
int absmax_launch(hipStream_t s, const float *x, size_t n, unsigned *slots);

// 3x3 convolution with <= 4 output channels (backward into the image) on the 4x4x1 MFMA.
size_t conv_small_packed_floats(int K);
int conv_small_pack(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                    float *packed);
int conv_small_launch(hipStream_t s, const float *x, const float *packed, float *y,
                      const float *mask, int K, int M, int H, int W);

int pool_forward_launch(hipStream_t s, const float *x, int C, int H, int W, int mode, float *y,
                        unsigned char *codes = nullptr);
int pool_backward_codes_launch(hipStream_t s, const float *dy, const unsigned char *codes, int C,
                               int H, int W, int mode, bool relu_mask, float *dx);
// dx = route(dy) [* (x > 0) when masked]; x is the pool input data (post-ReLU).
int pool_backward_launch(hipStream_t s, const float *dy, const float *x, int C, int H, int W,
                         int mode, bool relu_mask, float *dx);

// Gram: partial products over split-K slices, then a fixed-order reduction.
struct GramPlan {
    int C, HW, splits, tiles;      // tiles = lower-triangular 64x64 tiles
    int parts;                     // partial tiles written per slice (4: one per wave)
    size_t partial_floats;
};
GramPlan gram_plan(int C, int HW);
// f_amax (or null): kAmaxSlots words of float bits bounding |feat| -> the fp16 two-piece kernel
// (f16x2.h; ask gram_h2_usable first), whose partial tiles carry the square of the input scale:
// hand the same slots to gram_finish_launch.
bool gram_h2_usable(const float *feat, int C, int HW);
int gram_partials_launch(hipStream_t s, const float *feat, const GramPlan &plan, float *partials,
                         const unsigned *f_amax = nullptr);
// gram_out (lower-tri, upper zero) = sum_s partials * 1/(C*HW).  If target != null also writes
// dsym = sym(tril(gram - target)) and sumsq[0] = sum over the lower triangle of (gram-target)^2.
// `pieces` (optional, with target): dsym split into three bf16 matrices [3][C][C] for symm_bf3_launch
// (C a multiple of 64).  sumsq == null: the per-block sums of squares are left behind the Gram
// partials (gram_finish_blocks(plan) floats at partials + plan.partial_floats) for the caller to add.
int gram_finish_blocks(const GramPlan &plan);
// With a target, gram_finish_blocks(plan) words of float bits follow the block sums: max |dsym| per
// block (symm_h2_launch's d_amax).  f_amax: see gram_partials_launch.
// block_out (or null: behind the Gram partials): where the 2 x gram_finish_blocks(plan) block words go.
int gram_finish_launch(hipStream_t s, const float *partials, const GramPlan &plan, float *gram_out,
                       const float *target, float *dsym, float *sumsq, unsigned short *pieces = nullptr,
                       const unsigned *f_amax = nullptr, float *block_out = nullptr);

// S = dsym F (+ per-workgroup partial sums of |S|) on the bf16 matrix cores, three-piece split
// (symm.hip).  `pieces` is scratch for the split dsym: symm_pieces_elems(C) 16-bit words.
size_t symm_pieces_elems(int C);
int symm_num_workgroups(int C, int HW);
bool symm_bf3_usable(const float *feat, const float *out, int C, int HW);
// pieces_ready: gram_finish_launch already wrote them (else they are made from dsym here)
// The same product with two fp16 pieces per operand, three products (f16x2.h): dsym is read as it
// is; d_amax = n_damax words of float bits whose largest is max |dsym|, f_amax = kAmaxSlots words
// bounding |feat|.
bool symm_h2_usable(const float *feat, const float *out, int C, int HW);
int symm_h2_launch(hipStream_t s, const float *feat, const float *dsym, const unsigned *d_amax, int n_damax,
                   const unsigned *f_amax, float *out, float *partials, int C, int HW);
int symm_bf3_launch(hipStream_t s, const float *feat, const float *dsym, unsigned short *pieces,
                    bool pieces_ready, float *out, float *partials, int C, int HW);

// sums[0] = sum (F - Fc)^2, sums[1] = sum |F - Fc| over the tile window of the (virtually rolled)
// content map.
#ifdef __HIPCC__
// Window origin in the un-rolled content map, before wrapping: (oy - sy, ox - sx).
__device__ __forceinline__ int content_origin_y(const ContentWindow &w) { return w.oy - w.sy; }
__device__ __forceinline__ int content_origin_x(const ContentWindow &w) { return w.ox - w.sx; }
// Rolled content value at tile-feature position (c, y, x):
// roll2(Fc, (sx, sy))[c][oy + y][ox + x] = Fc[c][(oy + y - sy) mod ch][(ox + x - sx) mod cw]
__device__ __forceinline__ size_t content_index(const ContentWindow &w, int c, int y, int x) {
    int yy = (content_origin_y(w) + y) % w.ch;
    int xx = (content_origin_x(w) + x) % w.cw;
    if (yy < 0) yy += w.ch;
    if (xx < 0) xx += w.cw;
    return ((size_t)c * w.ch + yy) * w.cw + xx;
}
#endif

// A final sum a caller may postpone: dst[0] = the n floats at src, added in the fixed order of
// sum_partials_kernel.  The tile path collects the jobs of all its loss terms and runs them as ONE
// launch behind the forward pass (sum_jobs_launch) -- a dispatch costs ~5 us however little it does.
struct SumJob {
    const float *src;
    int n;
    float *dst;
};
int sum_jobs_launch(hipStream_t s, const SumJob *jobs, int n_jobs);
// defer (or null: the two sums are launched here): receives the two jobs instead
int content_sums_launch(hipStream_t s, const float *feat, const float *content,
                        const ContentWindow &win, float *sums /*[2]*/, std::vector<SumJob> *defer = nullptr);
// diff (=|+=) coef / (abs_sum/n + EPS) * term, term = S (style) or F - Fc (content).
// y_amax (optional, zeroed by the caller): max |value written| for a conv_h2 launch that reads diff next
int inject_style_launch(hipStream_t s, float *diff, const float *sgrad, size_t n,
                        const float *abs_sum, float coef, bool accumulate, unsigned *y_amax = nullptr);
int inject_content_launch(hipStream_t s, float *diff, const float *feat, const float *content,
                          const ContentWindow &win, const float *sums, float coef, bool accumulate,
                          unsigned *y_amax = nullptr);
int relu_inplace_launch(hipStream_t s, float *x, size_t n);
int sum_partials_launch(hipStream_t s, const float *partials, int n, float *out);
int sum_partials2_launch(hipStream_t s, const float *a, int na, float *out_a, const float *b, int nb,
                         float *out_b);

// image_ops.hip
int cut_tile_launch(hipStream_t s, const float *img, int H, int W, int rx, int ry, int y0, int x0,
                    int th, int tw, float *tile);
int put_tile_launch(hipStream_t s, float *grad, int H, int W, int rx, int ry, int y0, int x0,
                    int th, int tw, const float *tile);
int place_window_launch(hipStream_t s, float *dst, int dh, int dw, int y0, int x0, const float *src,
                        int C, int h, int w);
int roll_add_launch(hipStream_t s, float *acc, const float *src, int C, int h, int w, int sx, int sy,
                    float alpha, bool init);
int resample_launch(hipStream_t s, int axis, const float *src, int C, int H, int W, float *dst,
                    int OH, int OW, const int *bounds, const double *k, int ksize, int clamp);
int swt_haar_launch(hipStream_t s, const float *img, float *grad, int H, int W, int rx, int ry,
                    float scale, float power, double *loss_term, float *scratch,
                    size_t scratch_floats);
int regularizers_launch(hipStream_t s, const float *img, float *grad, int H, int W,
                        const float mean[3], float tv_scale, float tv_power, float p_scale,
                        float p_power, const float *aux, float aux_scale, int aux_rx, int aux_ry,
                        double *loss_terms /*[3]*/, float *scratch, size_t scratch_floats);
int adam_launch(hipStream_t s, float *params, const float *grad, float *g1, float *g2, float *p1,
                float *avg, size_t n, double lr, double b1, double b2, double bp1, double c1,
                double c2, double cp);
int dot_launch(hipStream_t s, const float *x, const float *y, size_t n, double *out_dev,
               float *scratch, size_t scratch_floats);
int abs_sum_launch(hipStream_t s, const float *x, size_t n, double *out_dev, float *scratch,
                   size_t scratch_floats);
int axpy_launch(hipStream_t s, float a, const float *x, float *y, size_t n);
int scale_launch(hipStream_t s, float a, float *x, size_t n);
int axpy_dev_launch(hipStream_t s, double c1, const double *a, double da, double c2, const double *b,
                    double db, const float *x, float *y, size_t n);
int scale_dev_launch(hipStream_t s, double c, const double *den, double den_div, float *x, size_t n);
// fused passes of the L-BFGS step (image_ops.hip): bit for bit the separate launches they replace
int axpy_dot_dev_launch(hipStream_t s, double c1, const double *a, double da, double c2, const double *b,
                        double db, double c_s, const double *den_s, double div_s, const float *x,
                        const float *src, float *y, const float *z, size_t n, double *out_dev, float *scratch,
                        size_t scratch_floats);
int lbfgs_pair_launch(hipStream_t s, const float *g_new, float *g_old, const float *sv, float *y, size_t n,
                      double *out_dev2, float *scratch, size_t scratch_floats);
int scale2_axpy_launch(hipStream_t s, float c1, float c2, float *sv, float *params, size_t n);
int step_stats_launch(hipStream_t s, const float *avg, float *old, int H, int W,
                      double *out_dev /*[2]*/, float *scratch, size_t scratch_floats);
int to_u8_launch(hipStream_t s, const float *img, int H, int W, const float mean[3], uint8_t *out);

}  // namespace stx
