/*
 * stx.h -- C ABI of libstx, the MI355X (gfx950) tile engine for tiled neural style transfer.
 *
 * This is the drop-in boundary for the hot path of crowsonkb/style_transfer.  The reference has
 * no native plugin ABI; its hot path sits behind the tile-worker message protocol
 * (style_transfer.py:156-166), served by TileWorker.process_one_request (style_transfer.py:215-259)
 * which calls CaffeModel.eval_features_tile / eval_sc_grad_tile (style_transfer.py:421-427,
 * 556-612), and behind the host-side numpy in eval_loss_and_grad (style_transfer.py:700-736) and
 * optimizers.py.  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - every function returns STX_OK (0) or a negative stx_status; nothing throws or aborts across
 *     the ABI; stx_last_error() gives the message of the calling thread's last failure;
 *   - plain pointers and sizes only.  A pointer argument that may live on either side carries an
 *     stx_mem tag: STX_HOST (pageable or pinned host memory) or STX_DEVICE (memory of the engine's
 *     GPU, e.g. from stx_malloc or any HIP allocation of this process);
 *   - buffers are caller-owned; float tensors are dense row-major float32 in the reference's
 *     layouts: images [3][H][W] (BGR, mean-subtracted), feature maps [C][h][w], Grams [C][C];
 *   - an engine owns one GPU and one HIP stream.  Calls on one engine are issued in order on that
 *     stream and are ASYNCHRONOUS w.r.t. the host unless stated; stx_sync() waits and then
 *     publishes pending scalar results (losses).  Engines on different GPUs are independent --
 *     that independence is the whole multi-GPU story (tiles never exchange data);
 *   - one host thread per engine at a time.
 */
#ifndef STX_H_
#define STX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stx_engine stx_engine;

typedef enum stx_status {
    STX_OK = 0,
    STX_ERR_ARG = -1,          /* bad argument (null pointer, unknown layer name, bad shape) */
    STX_ERR_HIP = -2,          /* a HIP runtime call or kernel launch failed */
    STX_ERR_STATE = -3,        /* call sequence error (e.g. weights or targets not set) */
    STX_ERR_UNSUPPORTED = -4,  /* graph / parameter outside what the kernels implement */
    STX_ERR_NOMEM = -5
} stx_status;

typedef enum stx_mem { STX_HOST = 0, STX_DEVICE = 1 } stx_mem;

typedef enum stx_layer_type {
    STX_LAYER_INPUT = 0, STX_LAYER_CONV = 1, STX_LAYER_RELU = 2, STX_LAYER_POOL = 3
} stx_layer_type;

typedef enum stx_pool_mode { STX_POOL_MAX = 0, STX_POOL_AVE = 1 } stx_pool_mode;

/* One layer of a Caffe deploy.prototxt (vgg19.prototxt:3-342): same names, same graph. */
typedef struct stx_layer_desc {
    const char *name;       /* layer name, e.g. "conv1_1", "relu1_1", "pool1" */
    int type;               /* stx_layer_type */
    const char *bottom;     /* input blob name (NULL for the input layer) */
    const char *top;        /* output blob name; ReLU must be in-place (top == bottom) */
    int num_output;         /* conv: output channels; input layer: image channels (3) */
    int kernel_size;        /* conv: 3 (pad 1) or 1 (pad 0); pool: 2 */
    int pad;                /* conv */
    int stride;             /* pool: 2 */
    int pool_mode;          /* stx_pool_mode */
} stx_layer_desc;

/* ---------------------------------------------------------------- library / device queries */
const char *stx_version(void);
/* Message of the calling thread's most recent failing call ("" if none). */
const char *stx_last_error(void);
/* The library's STX_* switches (A/B levers, test hooks) are read from a snapshot of
 * the environment taken at first use, not with getenv() at every launch; a program that changes one while it
 * runs calls this to take a new snapshot.  (No counterpart in the reference: its only run-time switches are the
 * command line's, config_system.py:26-118.) */
int stx_reread_env(void);
/* Number of visible GPUs; replaces detect_devices() (config_system.py:17-24, nvidia-smi -L). */
int stx_device_count(int *count);
/* gcnArchName of a device (e.g. "gfx950:sramecc+:xnack-") into buf. */
int stx_device_name(int device, char *buf, size_t buf_len);

/* ------------------------------------------------------------------------ engine lifetime */
/* Replaces TileWorker.run's setup (style_transfer.py:187-207: pick device, caffe.Net(deploy, 1,
 * weights)).  The graph is copied.  Weights are supplied separately with stx_set_conv_weights. */
int stx_engine_create(int device, const stx_layer_desc *layers, int n_layers, stx_engine **out);
/* Another worker on the primary's GPU: its own HIP stream and activation buffers, but the
 * primary's network, weights, packed filter banks and targets (one copy per GPU).  The reference
 * gives every TileWorker process its own caffe.Net (style_transfer.py:187-207) and sends the
 * targets to each (style_transfer.py:309-332); engines created here see what is set through any
 * engine of the group -- stx_set_conv_weights / stx_set_contents_and_styles need to be called
 * once per GPU. */
int stx_engine_create_shared(stx_engine *primary, stx_engine **out);
void stx_engine_destroy(stx_engine *e);
/* Caffe blob layout: weights [Cout][Cin][k][k], bias [Cout], float32 (Appendix C of SURVEY.md). */
int stx_set_conv_weights(stx_engine *e, const char *conv_layer, const float *weights,
                         const float *bias, int mem);
/* Waits for all work queued on the engine, then writes pending loss values. */
int stx_sync(stx_engine *e);
/* A step loop that runs AHEAD of the GPU.  The reference's loop blocks on every iteration's loss
 * and statistics before it starts the next one (style_transfer.py:799-815: resp_q.get(), then the
 * callback); here the host may queue iteration i + 1 first and collect iteration i afterwards.
 * stx_fence closes the engine's current set of pending loss values behind an event on its stream
 * and returns a ticket; calls queued after it collect theirs in a second set.  stx_fence_wait
 * waits for that event only -- not for the work queued since -- and writes the values of the set
 * the ticket names (a ticket that was already published, by stx_sync or because its set had to
 * be reused, is a no-op).  At most one closed set exists per engine: a second stx_fence before
 * the first was waited for publishes the older set itself (after waiting for it). */
int stx_fence(stx_engine *e, unsigned long long *ticket);
int stx_fence_wait(stx_engine *e, unsigned long long ticket);
/* Orders e's stream behind everything queued so far on other's stream (an event; no host wait).
 * The engines may sit on different GPUs.  This is what replaces the reference's blocking
 * resp_q.get() between handing out tiles and stitching their gradients (style_transfer.py:
 * 634-643): the master's stream waits for a worker's gradient, the host does not. */
int stx_engine_wait(stx_engine *e, stx_engine *other);
/* Counters for tests and bench.py. */
typedef enum stx_query {
    STX_Q_SHARED_ENGINES = 0,  /* engines sharing this engine's weights / targets (>= 1) */
    STX_Q_TARGET_UPLOADS = 1,  /* stx_set_contents_and_styles calls served by this group */
    STX_Q_TARGET_BYTES = 2,    /* bytes those calls copied, cumulative */
    STX_Q_WEIGHT_BYTES = 3,    /* device bytes of weights + packed banks held by this group */
    STX_Q_TILE_EVALS = 4,      /* stx_sc_grad_tile calls enqueued by this engine */
    STX_Q_PEERS_WITHOUT_ACCESS = 5  /* GPUs of the node this engine's GPU has no direct (xGMI peer)
                                     * access to: copies to / from them are staged by the runtime
                                     * (reported once on stderr) */
} stx_query;
int stx_engine_query(stx_engine *e, int what, double *value);
int stx_engine_device(stx_engine *e, int *device);
/* The engine's hipStream_t as an opaque pointer (for callers that enqueue their own copies). */
int stx_engine_stream(stx_engine *e, void **hip_stream);

/* ---------------------------------------------------------------------------- raw memory */
int stx_malloc(stx_engine *e, size_t bytes, void **dev_ptr);
int stx_free(stx_engine *e, void *dev_ptr);
int stx_memset_async(stx_engine *e, void *dev_ptr, int value, size_t bytes);
/* dst/src tagged with stx_mem; device<->device copies may cross GPUs (xGMI peer copy). */
int stx_memcpy_async(stx_engine *e, void *dst, int dst_mem, const void *src, int src_mem,
                     size_t bytes);

/* ------------------------------------------------------------------- targets (per scale) */
/* One content feature map: ContentData.features[layer] (style_transfer.py:165,246-249), the
 * FULL-image map [C][h][w] with h = ceil(H/scale), w = ceil(W/scale) (style_transfer.py:440). */
typedef struct stx_content_target {
    int content_index;      /* index into model.contents (normally 0) */
    const char *layer;
    int channels, height, width;
    const float *features;
    int mem;
} stx_content_target;

/* One style Gram: StyleData.grams[layer] (style_transfer.py:166,250-253), [C][C], only the lower
 * triangle is read (num_utils.py:53-66). */
typedef struct stx_style_target {
    int style_index;        /* index into model.styles (normally 0) */
    const char *layer;
    int channels;
    const float *gram;
    int mem;
} stx_style_target;

/* Replaces the SetContentsAndStyles message (style_transfer.py:163,243-254,309-332): replaces all
 * targets held by the engine.  Data is copied before the call returns control of the buffers
 * (host sources: synchronously; device sources: ordered on the engine stream). */
int stx_set_contents_and_styles(stx_engine *e, const stx_content_target *contents, int n_contents,
                                const stx_style_target *styles, int n_styles);

/* --------------------------------------------------------------------------- the hot path */
/* Replaces CaffeModel.eval_features_tile via FeatureMapRequest (style_transfer.py:156,221-228,
 * 421-427): forward the tile and return the post-ReLU maps of the requested blobs.
 * out[i] receives [C_i][ceil(th/scale_i)][ceil(tw/scale_i)] floats. */
int stx_features_tile(stx_engine *e, const float *img, int img_mem, int th, int tw,
                      const char *const *layers, int n_layers, float *const *out, int out_mem);

/* One tapped blob of an SCGradRequest: layer_weights[layer], content_weight[layer] (0 = not a
 * content layer), style_weight[layer] (0 = not a style layer) -- style_transfer.py:158-161,570,
 * 579-580,591-593 -- and dd_weight[layer] (0 = not a Deep-Dream layer): loss -= lw*dd*1/2|F|^2,
 * diff -= lw*dd*normalize(F), style_transfer.py:602-604. */
typedef struct stx_tap {
    const char *layer;
    double layer_weight;
    int is_content;
    double content_weight;
    int is_style;
    double style_weight;
    int is_dd;
    double dd_weight;
} stx_tap;

/* Replaces the SCGradRequest branch of TileWorker.process_one_request + CaffeModel.
 * eval_sc_grad_tile (style_transfer.py:230-241,556-612).
 *   img        [3][th][tw] tile of the (rolled) image;
 *   roll_xy    the request's roll: the engine addresses its content maps as if rolled by
 *              roll_xy // scale with numpy.roll(axis=(-1,-2)) semantics, i.e. roll_xy[0] shifts
 *              the W axis and roll_xy[1] the H axis (num_utils.py:136-140); no copy is made;
 *   start_yx   tile origin in the rolled image (y, x) (style_transfer.py:571-573);
 *   loss_out   host double; written by the next stx_sync() (or immediately if sync_now != 0);
 *   grad_out   [3][th][tw], the reference's diff['data'].
 */
int stx_sc_grad_tile(stx_engine *e, const float *img, int img_mem, int th, int tw,
                     const int roll_xy[2], const int start_yx[2], const stx_tap *taps, int n_taps,
                     double *loss_out, float *grad_out, int grad_mem, int sync_now);

/* Zero-copy hand-off for callers that produce the tile on the engine's own GPU (the tile farm
 * cuts tiles with stx_image_cut_tile): *tile_in is the engine's input blob sized for a
 * [3][th][tw] tile, *grad_out the blob its gradient is left in.  Passing exactly these pointers
 * as `img` / `grad_out` of stx_sc_grad_tile (same th, tw) skips the two device-to-device copies
 * of the call.  The pointers are valid until a larger tile is evaluated on this engine: query
 * them again before every use.  (The reference copies every tile into and out of POSIX shared
 * memory, style_transfer.py:634-643.) */
int stx_tile_buffers(stx_engine *e, int th, int tw, float **tile_in, float **grad_out);

/* Lower-triangular Gram of a feature map, F F^T / (C*h*w), upper triangle zero: gram_matrix
 * (num_utils.py:143-147) as used for the style targets (style_transfer.py:534). */
int stx_gram_matrix(stx_engine *e, const float *feat, int feat_mem, int channels, int hw,
                    float *gram_out, int gram_mem);

/* ------------------------------------------------- full-image ops (device-resident state) */
/* All pointers in this group are STX_DEVICE memory on the engine's GPU, images are [3][H][W].
 * The image and the optimizer state are kept UN-rolled; the per-iteration random shift
 * (style_transfer.py:777-806) is applied as an index offset when tiles are cut and when tile
 * gradients are put back, which is equivalent to the reference's roll / un-roll pair. */

/* tile[c][y][x] = img[c][(y0 + y - roll_xy[1]) mod H][(x0 + x - roll_xy[0]) mod W]: the window
 * [y0,y0+th) x [x0,x0+tw) of roll2(img, roll_xy) (style_transfer.py:632,661). */
int stx_image_cut_tile(stx_engine *e, const float *img, int H, int W, const int roll_xy[2],
                       int y0, int x0, int th, int tw, float *tile);
/* Inverse placement of a tile gradient into the un-rolled full gradient
 * (style_transfer.py:642 followed by the un-roll at 805-806). */
int stx_image_put_tile(stx_engine *e, float *grad, int H, int W, const int roll_xy[2],
                       int y0, int x0, int th, int tw, const float *tile_grad);

/* Feature-map preprocessing on the device (eval_features_once / prepare_features,
 * style_transfer.py:429-486), so that per-scale targets never visit the host:
 * stx_map_place puts one tile's [C][h][w] map into the full-image map [C][dst_h][dst_w] at
 * (y0, x0) (the stitch at style_transfer.py:457-461); stx_map_roll_add does one pass of the
 * rolled-tiling average (style_transfer.py:476-485) on [C][h][w] maps:
 * init_divisor != 0:  acc  = roll2(src, roll_xy) / init_divisor   (features = feats / passes)
 * init_divisor == 0:  acc += alpha * roll2(src, roll_xy)          (saxpy(1 / passes, ...)) */
int stx_map_place(stx_engine *e, float *dst, int channels, int dst_h, int dst_w, int y0, int x0,
                  const float *src, int h, int w);
int stx_map_roll_add(stx_engine *e, float *acc, const float *src, int channels, int h, int w,
                     const int roll_xy[2], double alpha, double init_divisor);

/* Resampling between pyramid scales (num_utils.resize, num_utils.py:90-108, used by
 * style_transfer.py:399-401 and optimizers.py:53-61): Pillow's separable 'F'-mode resampler --
 * horizontal pass, then vertical pass on the float32 intermediate, double accumulation.  The
 * caller supplies, per axis, the window table bounds[out][2] = (first input index, tap count) and
 * the normalised weights[out][ksize] (host memory); src [C][H][W] and dst [C][out_h][out_w] are
 * STX_DEVICE.  clamp_min_zero applies max(0, .) to the result (the Adam g2 state).  Synchronous. */
int stx_image_resample(stx_engine *e, const float *src, int channels, int H, int W, float *dst,
                       int out_h, int out_w, const int *bounds_x, const double *weights_x,
                       int ksize_x, const int *bounds_y, const double *weights_y, int ksize_y,
                       int clamp_min_zero);

/* TV + p-norm + auxiliary-image terms of eval_loss_and_grad (style_transfer.py:709-733,
 * num_utils.py:74-82,150-162): grad += tv_scale*d tv_norm(img/127.5, tv_power)
 *                                    + p_scale*d p_norm((img+mean-127.5)/127.5, p_power)
 *                                    + aux_scale*(img-aux')/127.5;
 * *loss_out (host, written at stx_sync) = tv_scale*TV + p_scale*P + aux_scale*A.
 * A scale of 0 disables a term; aux may be NULL.  mean_bgr is a host array of 3 floats.
 * aux_roll_xy (or NULL = no shift): the reference rolls the image by the iteration's shift but
 * not its auxiliary image (style_transfer.py:729-733,777-786), so in the un-rolled frame used here
 * aux'[y][x] = aux[(y + roll_xy[1]) mod H][(x + roll_xy[0]) mod W]. */
int stx_image_regularizers(stx_engine *e, const float *img, float *grad, int H, int W,
                           const float mean_bgr[3], double tv_scale, double tv_power,
                           double p_scale, double p_power, const float *aux, double aux_scale,
                           const int aux_roll_xy[2], double *loss_out);

/* The SWT term of eval_loss_and_grad (style_transfer.py:716-720, num_utils.py:179-196) for the
 * reference's defaults --swt-wavelet haar --swt-levels 1:
 *   D = detail part (approximation band zeroed) of the one-level stationary Haar transform of
 *       roll(img)/127.5, taken on its symmetric padding to a power-of-two square and cropped back;
 *   grad += scale * d p_norm(D, power) (the p-norm's own gradient at D, as the reference adds it);
 *   *loss_out (host, written at stx_sync) = scale * sum |D|^power.
 * roll_xy (or NULL): the iteration's shift; the image itself stays un-rolled.  PyWavelets is not
 * part of the reference tree: the transform is restated (oracle/num_ops.py), parity unpinned. */
int stx_image_swt_haar(stx_engine *e, const float *img, float *grad, int H, int W,
                       const int roll_xy[2], double scale, double power, double *loss_out);

/* AdamOptimizer.update after the gradient is known (optimizers.py:35-42), fused:
 *   g1 = b1*g1 + (1-b1)*grad; g2 = b2*g2 + (1-b2)*grad^2; p1 likewise on the new params;
 *   params -= lr * (g1/c1) / (sqrt(g2/c2) + EPS);  avg_out = p1/cp
 * where c1, c2, cp are the EWMA bias corrections (1 - beta^t, or 1 when uncorrected). */
int stx_adam_step(stx_engine *e, float *params, const float *grad, float *g1, float *g2, float *p1,
                  float *avg_out, size_t n, double lr, double b1, double b2, double bp1,
                  double corr1, double corr2, double corrp);

/* BLAS-1 pieces of LBFGSOptimizer (optimizers.py:74-121; num_utils.py:20-42). */
int stx_vec_dot(stx_engine *e, const float *x, const float *y, size_t n, double *out_host_sync);
int stx_vec_axpy(stx_engine *e, double a, const float *x, float *y, size_t n);
int stx_vec_scale(stx_engine *e, double a, float *x, size_t n);
int stx_vec_mean_abs(stx_engine *e, const float *x, size_t n, double *out_host_sync);
/* The same recursion with its scalars kept on the device (no host round trip per dot product;
 * LBFGSOptimizer.inv_hv, optimizers.py:105-121).  *_dev arguments point to STX_DEVICE doubles.
 *   stx_vec_dot_async:      *out_dev = <x, y>            stx_vec_abs_sum_async: *out_dev = sum |x|
 *   stx_vec_axpy_dev:       y += (float)(*a_dev / da * c1 [+ *b_dev / db * c2]) * x   (b_dev may be NULL)
 *   stx_vec_scale_dev:      x *= (float)(c / (*den_dev / den_div))
 * All asynchronous, ordered on the engine stream. */
int stx_vec_dot_async(stx_engine *e, const float *x, const float *y, size_t n, double *out_dev);
int stx_vec_abs_sum_async(stx_engine *e, const float *x, size_t n, double *out_dev);
int stx_vec_axpy_dev(stx_engine *e, double c1, const double *a_dev, double da, double c2,
                     const double *b_dev, double db, const float *x, float *y, size_t n);
int stx_vec_scale_dev(stx_engine *e, double c, const double *den_dev, double den_div, float *x,
                      size_t n);
/* Fused passes of the same step (round 5): each computes, value for value and bit for bit, what the
 * separate calls above compute, and reads an array once where they read it two or three times.
 *   stx_vec_axpy_dot_dev:   y = coef * x + src (coef as stx_vec_axpy_dev; src may be y); with scale_den_dev
 *                           != NULL then y *= (float)(scale_c / (*scale_den_dev / scale_div)); *out_dev =
 *                           <z, y> -- the axpy of one iteration of inv_hv's loops (and the scaling between
 *                           them) with the dot product of the next (optimizers.py:108-120)
 *   stx_vec_lbfgs_pair:     y = g_new - g_old, g_old = g_new, out_dev2[0] = <s, y>, out_dev2[1] = <y, y>;
 *                           *sy_host_sync = out_dev2[0] after a stream synchronisation -- the curvature
 *                           pair and its test (optimizers.py:84-85,97-103)
 *   stx_vec_scale2_axpy:    s = c2 * (c1 * s), params += s        (optimizers.py:76-82)              */
int stx_vec_axpy_dot_dev(stx_engine *e, double c1, const double *a_dev, double da, double c2,
                         const double *b_dev, double db, double scale_c, const double *scale_den_dev,
                         double scale_div, const float *x, const float *src, float *y, const float *z,
                         size_t n, double *out_dev);
int stx_vec_lbfgs_pair(stx_engine *e, const float *g_new, float *g_old, const float *s, float *y, size_t n,
                       double *out_dev2, double *sy_host_sync);
int stx_vec_scale2_axpy(stx_engine *e, double c1, double c2, float *s, float *params, size_t n);

/* Per-step statistics of transfer() (style_transfer.py:808-815):
 * stats[0] = mean|avg - old|, stats[1] = sqrt(mean(xdiff^2 + ydiff^2)) with circular forward
 * differences; then old <- avg.  Synchronous (returns after the values are on the host). */
int stx_image_step_stats(stx_engine *e, const float *avg, float *old, int H, int W,
                         double stats[2]);
/* The same kernel without the host wait: raw_sums[0] = sum|avg - old|, raw_sums[1] = sum(xdiff^2 +
 * ydiff^2) over the 3*H*W elements are written (host doubles) when the engine's pending values are
 * published -- stx_sync, or stx_fence_wait on a later fence; the caller divides by 3*H*W and
 * takes the root (style_transfer.py:808-812). */
int stx_image_step_stats_async(stx_engine *e, const float *avg, float *old, int H, int W,
                               double raw_sums[2]);

/* get_image (style_transfer.py:378-386): out_rgb_u8[H][W][3] = uint8(clip(img + mean, 0, 255))
 * with BGR->RGB flip and truncation toward zero.  out is STX_DEVICE memory. */
int stx_image_to_u8(stx_engine *e, const float *img, int H, int W, const float mean_bgr[3],
                    uint8_t *out_rgb_u8);

/* ------------------------------------------------------------- single-kernel test hooks */
/* Direct entry points to the individual kernels, used by tests/ to check each against the
 * oracle (x, w, b, y: STX_DEVICE).  w is the Caffe layout [Cout][Cin][k][k]. */
int stx_op_conv_forward(stx_engine *e, const float *x, int Cin, int H, int W, const float *w,
                        const float *b, int Cout, int ksize, int relu, float *y);
int stx_op_conv_backward_data(stx_engine *e, const float *dy, int Cout, int H, int W,
                              const float *w, int Cin, int ksize, const float *relu_mask_data,
                              float *dx);
int stx_op_pool_forward(stx_engine *e, const float *x, int C, int H, int W, int mode, float *y);
int stx_op_pool_backward(stx_engine *e, const float *dy, const float *x, int C, int H, int W,
                         int mode, const float *relu_mask_data, float *dx);

/* The loss terms of one tapped blob, launched exactly as stx_sc_grad_tile launches them
 * (style_transfer.py:575-593), for direct comparison with num_utils.gram_matrix / ssymm / norm2 /
 * normalize (num_utils.py:53-71,85-87,143-147).  All arrays STX_DEVICE.
 * stx_op_style_terms:   G = tril(F F^T)/(C*h*w);  D = G - gram_target (lower triangle);
 *                       half_sumsq = 1/2 sum D^2;  s_out = sym(D) F;  abs_sum = sum |s_out|;
 *                       normalized_out = s_out / (abs_sum/n + EPS)      (either output may be NULL)
 * stx_op_content_terms: c = F - roll2(content, roll_xy)[:, oy:oy+h, ox:ox+w];
 *                       sums[0] = sum c^2, sums[1] = sum |c|;  normalized_out = c / (sums[1]/n + EPS)
 * Both synchronise before returning the host scalars. */
int stx_op_style_terms(stx_engine *e, const float *feat, int channels, int h, int w,
                       const float *gram_target, float *s_out, float *normalized_out,
                       double *half_sumsq, double *abs_sum);
int stx_op_content_terms(stx_engine *e, const float *feat, int channels, int h, int w,
                         const float *content, int content_h, int content_w, int oy, int ox,
                         const int roll_xy[2], float *normalized_out, double sums[2]);

/* Per-kernel-group timing for tuning and for bench.py's roofline figures: while enabled, every
 * launch group of the tile path (one conv / pool / Gram / SYMM / injection) is bracketed by HIP
 * events on the engine stream.  stx_profile_read synchronises, writes one line per group
 * "label<TAB>milliseconds<TAB>algorithmic_flops<TAB>MHz" into buf (NUL-terminated, truncated to
 * buf_len; *needed receives the full size) and clears the record.  MHz: the shader clock inside the
 * group's convolution kernel while stx_clock_marks is on (0 otherwise / for other groups). */
int stx_profile_enable(stx_engine *e, int on);
int stx_profile_read(stx_engine *e, char *buf, size_t buf_len, size_t *needed);

/* Timing: ms spent by the GPU between the first and last kernel of the most recent FINISHED
 * stx_sc_grad_tile / stx_features_tile on this engine (HIP events on the engine stream; after
 * stx_sync that is the last call; while the host runs ahead it is the newest of the last four
 * that has completed, and only if none has does the call wait, for the newest). */
int stx_last_tile_ms(stx_engine *e, float *ms);

/* Matrix-core work of the convolutions of the same call: `algorithmic` counts every layer as a
 * direct convolution (2 * Cout * Cin * k * k * H * W, the figure SURVEY.md section 8d uses),
 * `issued` what the kernels actually put on the MFMA units (the Winograd kernels issue 2/3 or
 * 4/9 of it).  Gram / SYMM products are not included. */
int stx_last_tile_flops(stx_engine *e, double *algorithmic, double *issued);

/* The shader clock the GPU sustains INSIDE its dominant kernel (measurement only; nothing in the
 * reference corresponds).  While stx_clock_marks is on, one workgroup of every 2-D Winograd
 * convolution launch of this engine (the kernel that does 80 % of a tile evaluation's work) reads
 * the core-cycle counter and the constant 100 MHz counter before and after its chunk loop and
 * stores the two differences: the clock under exactly that load, with whatever the GPU's other
 * streams are running.  (Round 3 read the clock with a one-wave kernel BETWEEN the heavy kernels and
 * saw 2.43 GHz: the part clocks up the moment the matrix pipes go quiet; inside the convolutions it
 * sustains 2.0-2.3 GHz.)  stx_clock_marks_read synchronises the stream, returns the marks recorded
 * since the last read in MHz, launch order (*n_values <= max_values; at most 16384 are kept; 0
 * for a launch whose loop was shorter than a microsecond) and clears them.  bench.py switches them
 * on for its second, longer measurement only: the fp32 MFMA peak scales with this clock, which
 * depends on the kernels AND on their operands' bits.  With stx_profile_enable on as well,
 * stx_profile_read carries the mark of each convolution group as a fourth column. */
int stx_clock_marks(stx_engine *e, int on);
int stx_clock_marks_read(stx_engine *e, double *mhz, int max_values, int *n_values);

#ifdef __cplusplus
}
#endif
#endif /* STX_H_ */
