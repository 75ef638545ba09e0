// libstx host side: the tile engine behind include/stx.h.
//
// One engine = one GPU + one HIP stream + one copy of the network (graph, weights packed for the
// MFMA kernels) + the current targets (content feature maps, style Grams).  It plays the role of
// the reference's TileWorker process (style_transfer.py:169-259) with CaffeModel.eval_features_tile
// / eval_sc_grad_tile (style_transfer.py:421-427,556-612) inside, but is driven by plain function
// calls that enqueue kernels asynchronously instead of pickled messages over multiprocessing
// queues and POSIX shared memory.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace stx {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

// Growable device buffer.  Growth frees and reallocates (hipFree synchronises the device, so
// kernels still reading the old allocation have finished); it happens only when a larger tile
// than ever before arrives.
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return STX_OK;
        if (ptr) STX_HIP(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
        const size_t want = (need + 255) & ~(size_t)255;
        hipError_t err = hipMalloc(&ptr, want);
        if (err != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(err));
            ptr = nullptr;
            return STX_ERR_NOMEM;
        }
        bytes = want;
        return STX_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    float *f() const { return static_cast<float *>(ptr); }
};

struct Layer {
    std::string name, bottom, top;
    int type = 0, num_output = 0, ksize = 0, pad = 0, stride = 1, pool_mode = 0;
    int bottom_blob = -1, top_blob = -1;
};

struct Blob {
    std::string name;
    int channels = 0;
    int producer = -1;   // layer index that writes it (conv / pool / input)
    bool relu = false;   // an in-place ReLU layer follows the producer
    int scale = 1;       // 224 // height at a 224 input (CaffeModel.layer_info, style_transfer.py:415-419)
    int h = 0, w = 0;    // current tile
    DevBuf data, diff;
    DevBuf codes;             // pooled blobs: one window code per element (pool.hip), written by the
    bool codes_valid = false; // forward pass that produced `data` if its kernel can
    DevBuf relu_codes;        // rectified blobs: sign nibbles per 2x2 window (ConvProblem::out_codes / in_codes),
    bool relu_codes_valid = false;   // written by the convolution that produces the blob, or by the one that reads it
    bool relu_codes_wanted = false;  // ... or would have been, had its kernel taken them (ConvProblem::wants_codes)
    // max |data| / max |diff| (or an upper bound of it) on the device, for the fp16-split convolution
    // that reads the blob (conv_h2.hip): the slot group (a blob index) of the engine's table that
    // holds it -- the blob's own when its producer tracked it, the blob's below / above when a
    // pooling layer passed the bound on -- or -1 when nobody has left one in this pass
    int amax_data = -1, amax_diff = -1;
    size_t count() const { return (size_t)channels * h * w; }
};

struct ConvParams {
    int cin = 0, cout = 0, ks = 0;
    DevBuf w, b;                              // Caffe layout on the device
    bool set = false;
    std::map<int, std::unique_ptr<DevBuf>> packed;  // key = dir * 1024 + config id
};

struct ContentTarget {
    int index, blob, C, h, w;
    std::unique_ptr<DevBuf> feat;
};

struct StyleTarget {
    int index, blob, C;
    std::unique_ptr<DevBuf> gram;
};

struct LossTerm {
    size_t scalar_index;   // float in the host mirror of the scalar buffer
    double coef;
};

struct PendingLoss {
    double *out;
    std::vector<LossTerm> terms;       // sum coef * scalar
    std::vector<LossTerm> dterms;      // sum coef * double scalar (image ops)
};

// What the engines of one GPU have in common: the network's weights, the banks packed for the
// kernels and the current targets.  A farm runs several engines (HIP streams + activation
// buffers) per GPU; each holding its own copy cost 4 x (80 MB of weights + ~200 MB of packed
// banks + the per-scale content maps: 537 MB at 4096^2) per GPU and as many uploads over xGMI.
struct SharedState {
    std::map<int, ConvParams> conv;    // layer index -> params
    std::vector<ContentTarget> contents;
    std::vector<StyleTarget> styles;
    int n_contents = 0, n_styles = 0;
    std::vector<stx_engine *> members;
    std::mutex mutex;                  // packs and target swaps (members may be driven by different threads)
    size_t target_uploads = 0;         // stx_set_contents_and_styles calls that copied data
    double target_bytes = 0;           // bytes those calls copied (cumulative)
};

}  // namespace stx

using namespace stx;

struct stx_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool clock_marks = false;              // stx_clock_marks: one mark per 2-D Winograd launch
    DevBuf marks_buf;
    int marks_used = 0, last_mark = -1;    // (last_mark: the slot of the launch just queued, or -1)
    std::vector<std::unique_ptr<DevBuf>> sgrad_tap;   // S = sym(D) F of every style tap
    // start / stop of the last few tile calls (a ring: stx_last_tile_ms reports the newest call
    // that has finished, so a host that runs ahead does not wait for the call it just queued)
    static constexpr int kTimed = 4;
    hipEvent_t ev_start[kTimed] = {}, ev_stop[kTimed] = {};
    int ev_cur = 0;
    int ev_recorded = 0;               // ring slots that hold a recorded pair (at most kTimed)
    hipEvent_t ev_tune0 = nullptr, ev_tune1 = nullptr;
    bool timed = false;
    double flop_algorithmic = 0, flop_issued = 0;   // matrix work of the current / last tile call
    std::vector<Layer> layers;
    std::vector<Blob> blobs;
    std::map<std::string, int> blob_index, layer_index;
    std::shared_ptr<SharedState> sh;   // weights, packed banks, targets (shared per GPU)

    DevBuf splitk;                     // split-K partial sums of small-plane convolutions
    DevBuf amax;                       // [data | diff][blob][kAmaxSlots] words of float bits (Blob::amax_data)
    unsigned *amax_slots(int blob, bool diff) const {
        return static_cast<unsigned *>(amax.ptr) + ((size_t)(diff ? blobs.size() : 0) + blob) * kAmaxSlots;
    }
    // the first layer leaves the Gram partials of its own output when that blob is a style tap of
    // the call (conv_first.hip): which blob, whether this call's forward pass wrote them, how many
    DevBuf first_gram;
    int first_gram_blob = -1, first_gram_parts = 0;
    bool first_gram_valid = false;
    DevBuf gram_partials, gram, dsym, dsym_pieces, symm_partials, upload;
    DevBuf term_scratch;               // per style term of a tile call: block sums / maxima + SYMM partials (sum jobs)
    // Loss scalars of the calls queued so far: device floats (tile terms) and doubles (image-op
    // reductions), each with a pinned host mirror, and the losses that will be published from
    // them.  TWO arenas: stx_fence closes the current one behind an event and opens the other, so
    // that a step loop can queue iteration i + 1 before it waits (stx_fence_wait) for the
    // scalars of iteration i -- the host runs one iteration ahead of the GPU instead of letting
    // it idle while the statistics of a step travel home.
    struct ScalarArena {
        DevBuf scalars;                    // device floats
        float *host = nullptr;             // pinned mirror
        size_t used = 0;
        DevBuf dscalars;                   // device doubles (image-op reductions)
        double *dhost = nullptr;
        size_t dused = 0;
        std::vector<PendingLoss> pending;
        hipEvent_t fence = nullptr;
        unsigned long long ticket = 0;     // 0: open; else closed by stx_fence and not yet published
    };
    ScalarArena arena[2];
    int cur = 0;
    unsigned long long next_ticket = 1;
    ScalarArena &A() { return arena[cur]; }
    size_t scalars_cap = 0;
    size_t n_tile_evals = 0;               // stx_sc_grad_tile calls (STX_Q_TILE_EVALS)
    std::vector<hipEvent_t> fence_events;     // stx_engine_wait: ring of events recorded on this stream
    size_t fence_next = 0;
    size_t dscalars_cap = 64;
    DevBuf red_scratch;                // float partials for image-op reductions

    bool winograd = true;   // 1-D Winograd F(2,3) for the 3x3 layers (STX_WINOGRAD=0: direct only)
    bool autotune = true;   // tile-config autotuning (process-wide cache, see choose_conv_config)
    bool pool_codes = true; // forward pooling leaves window codes for the backward pass (STX_POOL_CODES=0: off)

    // optional per-kernel-group timing (stx_profile_enable): event pairs around launch groups
    bool profiling = false;
    struct ProfEntry {
        std::string label;
        double flops;
        hipEvent_t start, stop;
        int mark = -1;      // clock mark of the group's convolution launch (stx_clock_marks), or -1
    };
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;

    int set_device() {
        STX_HIP(hipSetDevice(device));
        return STX_OK;
    }
    int find_blob(const char *name) const {
        if (!name) return -1;
        auto it = blob_index.find(name);
        return it == blob_index.end() ? -1 : it->second;
    }
};

namespace {

constexpr size_t kScalarFloats = 1 << 18;   // per-call scalar arena (sums + small partials)

// RAII timing of one launch group when profiling is on (no-op otherwise).
struct ProfScope {
    stx_engine *e;
    int index = -1;
    hipStream_t stream;
    ProfScope(stx_engine *eng, const std::string &label, double flops, hipStream_t on = nullptr)
        : e(eng), stream(on ? on : eng->stream) {
        if (!e->profiling) return;
        auto take = [&]() {
            hipEvent_t ev = nullptr;
            if (!e->event_pool.empty()) {
                ev = e->event_pool.back();
                e->event_pool.pop_back();
            } else if (hipEventCreate(&ev) != hipSuccess) {
                ev = nullptr;
            }
            return ev;
        };
        stx_engine::ProfEntry pe{label, flops, take(), take()};
        if (!pe.start || !pe.stop) return;
        (void)hipEventRecord(pe.start, stream);
        e->prof.push_back(pe);
        index = (int)e->prof.size() - 1;
    }
    ~ProfScope() {
        if (index < 0) return;
        (void)hipEventRecord(e->prof[index].stop, stream);
        e->prof[index].mark = e->last_mark;
        e->last_mark = -1;
    }
};

double conv_flops(int K, int M, int H, int W, int ks) {
    return 2.0 * K * M * ks * ks * (double)H * W;
}


int alloc_scalars(stx_engine *e, size_t n, size_t *index) {
    stx_engine::ScalarArena &a = e->A();
    if (a.used + n > e->scalars_cap) {
        set_error("scalar arena exhausted (%zu + %zu > %zu)", a.used, n, e->scalars_cap);
        return STX_ERR_NOMEM;
    }
    *index = a.used;
    a.used += n;
    return STX_OK;
}

int alloc_dscalars(stx_engine *e, size_t n, size_t *index) {
    stx_engine::ScalarArena &a = e->A();
    if (a.dused + n > e->dscalars_cap - 4) {   // the last slots serve synchronous results
        // no wrap: results are consumed at every stx_sync / stx_fence_wait, which also resets the arena
        set_error("double-scalar arena exhausted; call stx_sync more often");
        return STX_ERR_NOMEM;
    }
    *index = a.dused;
    a.dused += n;
    return STX_OK;
}

// Copies caller memory (host or device) into a device destination on the engine stream.
int copy_in(stx_engine *e, void *dst, const void *src, int mem, size_t bytes) {
    if (mem == STX_HOST)
        STX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
    else   // the source may live on another GPU of the node (peer copy over xGMI)
        STX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, e->stream));
    return STX_OK;
}

int copy_out(stx_engine *e, void *dst, int mem, const void *src, size_t bytes) {
    if (mem == STX_HOST)
        STX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
    else
        STX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, e->stream));
    return STX_OK;
}

// Sets blob shapes for a th x tw tile and makes sure data buffers exist for `needed` blobs.
int shape_blobs(stx_engine *e, int th, int tw, const std::vector<char> &needed, bool with_diff) {
    Blob &in = e->blobs[e->layers[0].top_blob];
    in.h = th;
    in.w = tw;
    for (size_t li = 1; li < e->layers.size(); ++li) {
        const Layer &L = e->layers[li];
        if (L.type == STX_LAYER_RELU) continue;
        const Blob &b = e->blobs[L.bottom_blob];
        Blob &t = e->blobs[L.top_blob];
        if (L.type == STX_LAYER_CONV) {
            t.h = b.h;
            t.w = b.w;
        } else {
            t.h = pooled_len(b.h);
            t.w = pooled_len(b.w);
        }
    }
    for (size_t bi = 0; bi < e->blobs.size(); ++bi) {
        if (!needed[bi]) continue;
        Blob &b = e->blobs[bi];
        STX_TRY(b.data.ensure(b.count() * sizeof(float)));
        if (with_diff) STX_TRY(b.diff.ensure(b.count() * sizeof(float)));
    }
    return STX_OK;
}

// Marks `blob` and everything it depends on.
void mark_ancestors(const stx_engine *e, int blob, std::vector<char> &needed) {
    while (blob >= 0 && !needed[blob]) {
        needed[blob] = 1;
        const int p = e->blobs[blob].producer;
        if (p <= 0) break;
        blob = e->layers[p].bottom_blob;
    }
}

int get_packed(stx_engine *e, int layer, int dir, const ConvConfig &cfg, const float **out) {
    std::lock_guard<std::mutex> lock(e->sh->mutex);
    ConvParams &cp = e->sh->conv[layer];
    if (!cp.set) {
        set_error("weights of layer %s were never set", e->layers[layer].name.c_str());
        return STX_ERR_STATE;
    }
    // (the 2-D geometries share a bank, so do the two channel tilings of the fp16-split kernel)
    const int key = dir * 1024 + (cfg.id >= 300 ? 300 : cfg.id >= 200 ? 200 : cfg.id);
    auto it = cp.packed.find(key);
    if (it == cp.packed.end()) {
        const int M = dir ? cp.cin : cp.cout, K = dir ? cp.cout : cp.cin;
        std::unique_ptr<DevBuf> buf(new DevBuf);
        if (cfg.id >= 100) {   // Winograd-transformed bank
            STX_TRY(buf->ensure(wino_packed_floats(cfg, K, M) * sizeof(float)));
            STX_TRY(wino_pack_weights(e->stream, cp.w.f(), cp.cout, cp.cin, dir, cfg, buf->f()));
        } else {
            STX_TRY(buf->ensure(conv_packed_floats(cfg, K, M, cp.ks) * sizeof(float)));
            STX_TRY(conv_pack_weights(e->stream, cp.w.f(), cp.cout, cp.cin, cp.ks, dir, cfg,
                                      buf->f()));
        }
        // the other engines of this GPU will read the bank from their own streams
        if (e->sh->members.size() > 1) STX_HIP(hipStreamSynchronize(e->stream));
        it = cp.packed.emplace(key, std::move(buf)).first;
    }
    *out = it->second->f();
    return STX_OK;
}

// Picks the tile configuration of a packed-weight convolution.  All configurations accumulate k
// in the same order, so they produce bit-identical results; which one is fastest depends on how
// many workgroups the plane yields (co-resident workgroups hide each other's stage swaps and
// epilogues).  The first time a shape is seen every candidate is timed with HIP events on the
// engine stream (a few launches, once per shape and scale) and the winner is cached.
// The cache is shared by all engines of the process (several engines drive the same GPU as
// separate streams; they must agree, and later ones need not re-measure).  Key: device + shape.
static std::mutex g_tuned_mutex;
static std::map<std::vector<int>, int> g_tuned;

// 3x3 layers with more than 32 output channels that the fp16-split kernel does not take (h2_choice) run the
// fp32 2-D Winograd kernel F(2x2,3x3) (4/9 of the direct kernel's MFMAs); everything else the direct kernel.
// The choice depends on the shape only, never on timing: the rounding differs between the kernels, and a
// given shape must always take the same path.  STX_CONV_ALGO=direct|wino2|wino2a|wino2b|wino2c overrides it
// for tests and measurements (a / b / c: one patch geometry only).
static bool wino_choice(stx_engine *e, int ksize, int K, int M, int H, int W, ConvConfig *out,
                        bool inject = false) {
    (void)inject;
    if (ksize != 3 || K < 8 || M <= 4) return false;
    const char *algo = sw_env("STX_CONV_ALGO");
    if (algo && *algo) {
        if (!strcmp(algo, "direct")) return false;
        if (!strcmp(algo, "wino2")) { *out = wino2_config(wino2_pick_geometry(H, W)); return true; }
        if (!strcmp(algo, "wino2a")) { *out = wino2_config(0); return true; }   // one geometry only
        if (!strcmp(algo, "wino2b")) { *out = wino2_config(1); return true; }
        if (!strcmp(algo, "wino2c")) { *out = wino2_config(2); return true; }
    }
    if (!e->winograd || M <= 32) return false;
    *out = wino2_config(wino2_pick_geometry(H, W));
    return true;
}

int choose_conv_config(stx_engine *e, int li, int dir, ConvProblem p, ConvConfig *out,
                       bool inject = false) {
    if (wino_choice(e, p.ksize, p.K, p.M, p.H, p.W, out, inject)) return STX_OK;
    const ConvConfig fallback = conv_pick_config(p.ksize, p.K, p.M, p.H, p.W);
    *out = fallback;
    if (!e->autotune || p.ksize != 3 || p.K <= 4 || p.M <= 32) return STX_OK;
    // planes too small to fill the chip run the small-tile config with a K split that depends on
    // the shape only (split results differ in rounding from unsplit ones, so no timing here)
    if (conv_splitk_factor(fallback, p, true) > 1 ||
        conv_num_workgroups(conv_config_by_id(5), p.M, p.H, p.W) < 256)
        return STX_OK;
    const std::vector<int> key = {e->device, p.ksize, p.K, p.M, p.H, p.W, p.epilogue};
    {
        std::lock_guard<std::mutex> lock(g_tuned_mutex);
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            *out = conv_config_by_id(it->second);
            return STX_OK;
        }
    }
    const int candidates[] = {0, 1, 2, 5};
    float best_ms = 1e30f;
    int best = fallback.id;
    for (int id : candidates) {
        const ConvConfig cfg = conv_config_by_id(id);
        if (cfg.bm > 64 && p.M <= 64) continue;            // half-empty channel tiles
        const float *packed = nullptr;
        STX_TRY(get_packed(e, li, dir, cfg, &packed));
        p.w = packed;
        STX_TRY(conv_launch(e->stream, cfg, p, true));     // warm (also builds nothing lazily)
        STX_HIP(hipEventRecord(e->ev_tune0, e->stream));
        for (int r = 0; r < 2; ++r) STX_TRY(conv_launch(e->stream, cfg, p, true));
        STX_HIP(hipEventRecord(e->ev_tune1, e->stream));
        STX_HIP(hipEventSynchronize(e->ev_tune1));
        float ms = 0.f;
        STX_HIP(hipEventElapsedTime(&ms, e->ev_tune0, e->ev_tune1));
        if (ms < best_ms) {
            best_ms = ms;
            best = id;
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_tuned_mutex);
        g_tuned[key] = best;
    }
    *out = conv_config_by_id(best);
    return STX_OK;
}

// Gives the problem a split-K scratch buffer when conv_launch will slice the reduction.
int attach_splitk(stx_engine *e, const ConvConfig &cfg, ConvProblem &p) {
    const size_t need = conv_splitk_floats(cfg, p, true);
    if (!need) return STX_OK;
    STX_TRY(e->splitk.ensure(need * sizeof(float)));
    p.splitk_ws = e->splitk.f();
    p.splitk_ws_floats = e->splitk.bytes / sizeof(float);
    return STX_OK;
}

constexpr int kMaxClockMarks = 16384;

int launch_conv(stx_engine *e, const ConvConfig &cfg, const ConvProblem &problem) {
    ConvProblem p = problem;
    // stx_clock_marks: the eight-wave Winograd kernel times the chunk loop of one of its workgroups
    e->last_mark = -1;
    if (e->clock_marks && ((cfg.id >= 200 && cfg.id < 210) || cfg.id >= 300) && e->marks_used < kMaxClockMarks) {
        e->last_mark = e->marks_used++;
        p.clock_out = static_cast<long long *>(e->marks_buf.ptr) + 2 * (size_t)e->last_mark;
    }
    // bookkeeping for stx_last_tile_flops: algorithmic = direct convolution, issued = what the
    // chosen kernel puts on the matrix cores (tile padding not counted)
    const double direct = 2.0 * p.M * p.K * p.ksize * p.ksize * (double)p.H * p.W;
    e->flop_algorithmic += direct;
    // (ids 300+: 6 of 9 multiplies, each as three fp16 products of 1/16 of an fp32 MFMA's time per k)
    e->flop_issued += cfg.id >= 300   ? direct * (6.0 / 9.0) * (3.0 / 16.0)
                      : cfg.id >= 200 ? direct * 4.0 / 9.0
                      : cfg.id >= 100 ? direct * 2.0 / 3.0
                                      : direct;
    if (cfg.id >= 100) return wino_launch(e->stream, cfg, p, conv_splitk_factor(cfg, p, true));
    return conv_launch(e->stream, cfg, p, true);
}

// Which fused-pooling kernels also leave the window codes the backward pooling runs from: the
// eight-wave 2-D Winograd kernel does, the four-wave one (ids 210+) writes the pooled values only.
static bool conv_writes_pool_codes(const ConvConfig &cfg) { return (cfg.id >= 200 && cfg.id < 210) || cfg.id >= 300; }

// True if a launch of `p` under `cfg` writes p.pool_out itself (2-D Winograd, no K split).
static bool conv_fuses_pool(const ConvConfig &cfg, const ConvProblem &p) {
    // (STX_POOL_FWD_FUSE=0: the stand-alone pooling kernel everywhere, for A/B measurements and tests)
    const char *env = sw_env("STX_POOL_FWD_FUSE");
    if (env && atoi(env) == 0) return false;
    if (cfg.id >= 300) return conv_splitk_factor(cfg, p, true) == 1 && h2_fuses_pool(p);
    return cfg.id >= 200 && conv_splitk_factor(cfg, p, true) == 1 && wino2_fuses_pool(p);
}

// The fp16-split kernel (conv_h2.hip) for this problem?  By shape and epilogue only -- never by timing:
// it rounds differently from the fp32 kernels, and a given shape must always take the same path.
//   forward: layers with at least STX_CONV_H2 input channels (default 64; 0: never);
//   backward: at least STX_CONV_H2_BWD channels of incoming gradient (default 64).
// A forward blob that differs in its last bits flips ReLU / max-pooling near-ties, and the two 64-channel
// layers hold most of a tile's decisions.  Every bound of tests/ holds with them on this kernel; the one
// chaotic fixture -- the reference's L-BFGS run of BASELINE config 4 in miniature, tiles of 30 x 33 pixels
// -- follows another of the REFERENCE'S OWN branches (tests/golden/cfg4_sensitivity.py: the reference with
// its convolutions rounded at this level takes that branch in half of its runs).
// The backward pass decides nothing: its rounding moves the gradient by 1e-7 and no further.
// STX_CONV_ALGO=h2|h2a|h2b|h2c forces the kernel (any / the 64- / the 128-channel / the two-patch tiling)
// wherever it applies.
static bool h2_enabled() {
    const char *algo = sw_env("STX_CONV_ALGO");
    if (algo && *algo) return !strncmp(algo, "h2", 2);
    // (the thresholds of h2_choice: with STX_CONV_H2=0 and no STX_CONV_H2_BWD neither direction takes the kernel)
    const char *env = sw_env("STX_CONV_H2"), *envb = sw_env("STX_CONV_H2_BWD");
    const int fwd_min = env ? atoi(env) : 64;
    const int bwd_min = envb ? atoi(envb) : (env && atoi(env) <= 0 ? 0 : 64);
    return fwd_min > 0 || bwd_min > 0;
}

static bool h2_choice(const ConvProblem &p, ConvConfig *out) {
    const char *algo = sw_env("STX_CONV_ALGO");
    int force = 0;
    if (algo && *algo) {
        if (!strcmp(algo, "h2")) force = 4;
        else if (!strcmp(algo, "h2a")) force = 1;
        else if (!strcmp(algo, "h2b")) force = 2;
        else if (!strcmp(algo, "h2c")) force = 3;
        else return false;             // some other kernel family was asked for
    }
    const char *env = sw_env("STX_CONV_H2"), *envb = sw_env("STX_CONV_H2_BWD");
    const int min_k = p.epilogue == kEpiForward ? (env ? atoi(env) : 64)
                                                : (envb ? atoi(envb) : env && atoi(env) <= 0 ? 0 : 64);
    if (!force && (min_k <= 0 || p.K < min_k || p.M < 64)) return false;
    if (!h2_usable(p)) return false;
    *out = force == 1 ? h2_config(1) : force == 2 ? h2_config(2) : force == 3 ? h2_config(1, 2) : h2_pick_config(p);
    return true;
}

// The slots with max |x| of a blob's data / diff for a kernel that is about to read it: what its
// producer left (Blob::amax_data / amax_diff), else a pass over the array now.
static int amax_for(stx_engine *e, int blob, bool diff, const unsigned **out) {
    Blob &b = e->blobs[blob];
    int &src = diff ? b.amax_diff : b.amax_data;
    if (src < 0) {
        ProfScope scope(e, std::string("absmax ") + b.name, 0.0);
        STX_TRY(absmax_launch(e->stream, diff ? b.diff.f() : b.data.f(), b.count(), e->amax_slots(blob, diff)));
        src = blob;
    }
    *out = e->amax_slots(src, diff);
    return STX_OK;
}

// `pool` (or null): the 2x2/2 pooling layer that consumes this convolution's blob; *pooled tells
// the caller whether the convolution wrote its output too.
int run_conv_forward(stx_engine *e, int li, bool force_relu, const Layer *pool = nullptr,
                     bool *pooled = nullptr, bool relu_codes = false, bool top_unobserved = false,
                     bool out_codes_wanted = false) {
    const Layer &L = e->layers[li];
    Blob &b = e->blobs[L.bottom_blob];
    Blob &t = e->blobs[L.top_blob];
    const ConvParams &cp = e->sh->conv[li];
    ConvProblem p{};
    p.x = b.data.f();
    p.y = t.data.f();
    p.bias = cp.b.f();
    p.K = cp.cin;
    p.M = cp.cout;
    p.H = b.h;
    p.W = b.w;
    p.ksize = cp.ks;
    p.relu = (t.relu || force_relu) ? 1 : 0;
    p.epilogue = kEpiForward;
    if (conv_first_usable(cp.cin, cp.cout, cp.ks) && !pool) {
        // the first layer: its own kernel, straight from the Caffe-layout bank; with the Gram
        // partials of the blob when it is a style tap of this call
        if (pooled) *pooled = false;
        b.relu_codes_valid = false;
        b.relu_codes_wanted = false;
        t.relu_codes_valid = false;
        unsigned *y_amax = nullptr;
        t.amax_data = -1;
        if (h2_enabled()) {
            y_amax = e->amax_slots(L.top_blob, false);
            t.amax_data = L.top_blob;
        }
        float *gram = nullptr;
        if (L.top_blob == e->first_gram_blob) {
            const int parts = conv_first_workgroups(b.h, b.w);
            // (+ room for gram_finish's per-block sums of squares behind the partial tiles)
            STX_TRY(e->first_gram.ensure(((size_t)parts * 64 * 64 + 64 * 64 / 64 + 64) * sizeof(float)));
            gram = e->first_gram.f();
            e->first_gram_parts = parts;
            e->first_gram_valid = true;
        }
        ProfScope scope(e, "fwd " + L.name, conv_flops(cp.cin, cp.cout, b.h, b.w, cp.ks));
        const double direct = conv_flops(cp.cin, cp.cout, b.h, b.w, cp.ks);
        e->flop_algorithmic += direct;
        e->flop_issued += direct;
        return conv_first_launch(e->stream, p.x, cp.w.f(), cp.b.f(), p.y, cp.cin, b.h, b.w, p.relu, gram, y_amax);
    }
    ConvConfig cfg;
    // the fp16-split kernel where it applies (it neither writes nor reads ReLU nibbles)
    const bool h2 = h2_choice(p, &cfg);
    // (a blob whose producer already left its nibbles needs none from its consumer)
    b.relu_codes_wanted = !h2 && !b.relu_codes_valid && relu_codes && b.relu && b.channels <= 128;      // (see below)
    p.wants_codes = b.relu_codes_wanted;
    if (!h2) STX_TRY(choose_conv_config(e, li, 0, p, &cfg));
    t.amax_data = -1;
    if (h2) STX_TRY(amax_for(e, L.bottom_blob, false, &p.x_amax));
    // the eight-wave fp32 kernel leaves its output's maximum too (conv3_1 feeds conv3_2)
    if (h2 || (cfg.id >= 200 && cfg.id < 210 && h2_enabled())) {
        p.y_amax = e->amax_slots(L.top_blob, false);
        t.amax_data = L.top_blob;          // (a K-sliced launch leaves it through its reduce pass)
    }
    const float *packed = nullptr;
    STX_TRY(get_packed(e, li, 0, cfg, &packed));
    p.w = packed;
    STX_TRY(attach_splitk(e, cfg, p));
    // a backward pass will follow: let this layer leave the sign nibbles of its (rectified) input
    // (up to 128 input channels -- conv1_2 and conv2_2 of a VGG: their backward pass is co-limited
    // by HBM and gains 40 / 24 us from the byte masks on a 1024^2 tile, while emitting them costs
    // the forward pass 10 / 16 us; from 256 channels on the backward pass is matrix-bound, gains
    // 0-7 us and the forward pass pays 5-10: measured, profiles/r03_relu_codes_ab.txt)
    if (b.relu_codes_wanted) {
        const size_t bytes = (size_t)b.channels * ((b.h + 1) / 2) * ((b.w + 1) / 2);
        STX_TRY(b.relu_codes.ensure(bytes));
        p.in_codes = static_cast<unsigned char *>(b.relu_codes.ptr);
        b.relu_codes_valid = conv_uses_relu_codes(cfg, p, conv_splitk_factor(cfg, p, true));
        if (!b.relu_codes_valid) p.in_codes = nullptr;
    }
    if (pooled) *pooled = false;
    if (pool) {
        Blob &pt = e->blobs[pool->top_blob];
        p.pool_out = pt.data.f();
        p.pool_mode = pool->pool_mode;
        pt.codes_valid = false;
        pt.amax_data = -1;
        if (conv_fuses_pool(cfg, p)) {
            *pooled = true;
            pt.amax_data = t.amax_data;    // max (or mean) of 2x2 windows: the same bound
            if (conv_writes_pool_codes(cfg) && e->pool_codes) {
                STX_TRY(pt.codes.ensure(pt.count() + 4));    // (+ 4: conv_h2.hip fetches three codes as one dword)
                p.pool_codes = static_cast<unsigned char *>(pt.codes.ptr);
                pt.codes_valid = true;
                // the full-resolution blob is then dead weight unless somebody looks at it: the
                // next layer reads the pooled blob, the backward pooling the codes (conv1_2 of a
                // 1024^2 tile: 268 MB that were written and never read)
                p.skip_y = top_unobserved;
            }
        } else {
            p.pool_out = nullptr;
        }
    }
    // ... and the nibbles of its own (rectified) output, when a convolution reads it and its backward
    // pass will mask with it: the epilogue holds one 2x2 window per lane and channel, so the byte
    // costs a handful of compares -- and the consumer's backward epilogue reads 1 byte instead of
    // 16 per lane and channel (the epilogues of one round all run at the same moment: their reads
    // and stores are a bandwidth-bound burst)
    t.relu_codes_valid = false;
    // (only beside the fp16-split kernels: STX_CONV_H2=0 keeps round 4's schedule to the letter)
    if (relu_codes && t.relu && out_codes_wanted && h2_enabled()) {
        const size_t bytes = (size_t)t.channels * ((t.h + 1) / 2) * ((t.w + 1) / 2);
        STX_TRY(t.relu_codes.ensure(bytes));
        p.out_codes = static_cast<unsigned char *>(t.relu_codes.ptr);
        t.relu_codes_valid = conv_writes_out_codes(cfg, p, conv_splitk_factor(cfg, p, true));
        if (!t.relu_codes_valid) p.out_codes = nullptr;
    }
    ProfScope scope(e, "fwd " + L.name, conv_flops(cp.cin, cp.cout, b.h, b.w, cp.ks));
    return launch_conv(e, cfg, p);
}

// The backward problem of convolution layer li, as far as the choice of kernel depends on it.
static ConvProblem conv_backward_shape(stx_engine *e, int li) {
    const Layer &L = e->layers[li];
    const Blob &b = e->blobs[L.bottom_blob];
    const ConvParams &cp = e->sh->conv[li];
    ConvProblem p{};
    p.K = cp.cout;
    p.M = cp.cin;
    p.H = b.h;
    p.W = b.w;
    p.ksize = cp.ks;
    p.epilogue = kEpiDgrad;
    return p;
}

// Can the backward pass of convolution li take the gradient of the 2x2/2 pooling layer behind it as it
// stands -- pooled, with the window codes -- and route it inside its own patch staging (conv_h2.hip, PIN)?
// Then the pooling layer's backward kernel does not run, and the gradient of the convolution's output
// blob (four times the pooled one) is neither written nor read.  STX_POOL_BWD_FUSE=0 keeps the kernel.
static bool conv_backward_takes_pooled(stx_engine *e, int li) {
    const char *env = sw_env("STX_POOL_BWD_FUSE");
    if (env && atoi(env) == 0) return false;
    const ConvProblem p = conv_backward_shape(e, li);
    ConvConfig cfg;
    return p.ksize == 3 && p.M > 4 && h2_choice(p, &cfg) && h2_takes_pooled_input(cfg, p);
}

// `pooled` (or null): the pooling layer behind this convolution whose backward pass the caller skipped
// (conv_backward_takes_pooled): the incoming gradient is that of the pooled blob.
int run_conv_backward(stx_engine *e, int li, const ConvInject *inj, bool *fused, const Layer *pooled = nullptr) {
    const Layer &L = e->layers[li];
    Blob &b = e->blobs[L.bottom_blob];
    const Blob &t = e->blobs[L.top_blob];
    const ConvParams &cp = e->sh->conv[li];
    ConvProblem p{};
    p.x = t.diff.f();
    p.y = b.diff.f();
    p.mask = b.relu ? b.data.f() : nullptr;
    // (kernels that cannot read the nibbles use the fp32 blob: conv_uses_relu_codes)
    p.mask_codes = b.relu && b.relu_codes_valid ? static_cast<const unsigned char *>(b.relu_codes.ptr) : nullptr;
    p.wants_codes = b.relu && b.relu_codes_wanted;
    p.K = cp.cout;
    p.M = cp.cin;
    p.H = b.h;
    p.W = b.w;
    p.ksize = cp.ks;
    p.epilogue = kEpiDgrad;
    if (cp.ks == 3 && cp.cin <= 4) {
        // backward into a <= 4-channel blob (the image): dedicated 4x4x1-MFMA kernel
        if (fused) *fused = false;
        std::lock_guard<std::mutex> lock(e->sh->mutex);
        ConvParams &cpm = e->sh->conv[li];
        auto it = cpm.packed.find(1 * 1024 + 999);
        if (it == cpm.packed.end()) {
            std::unique_ptr<DevBuf> buf(new DevBuf);
            STX_TRY(buf->ensure(conv_small_packed_floats(cp.cout) * sizeof(float)));
            STX_TRY(conv_small_pack(e->stream, cp.w.f(), cp.cout, cp.cin, 1, buf->f()));
            if (e->sh->members.size() > 1) STX_HIP(hipStreamSynchronize(e->stream));
            it = cpm.packed.emplace(1 * 1024 + 999, std::move(buf)).first;
        }
        ProfScope scope(e, "bwd " + L.name, conv_flops(cp.cout, cp.cin, b.h, b.w, cp.ks));
        const double direct = 2.0 * p.M * p.K * 9 * (double)p.H * p.W;
        e->flop_algorithmic += direct;
        e->flop_issued += direct * 4.0 / p.M;    // the 4x4x1 MFMA computes four output channels
        return conv_small_launch(e->stream, p.x, it->second->f(), p.y, p.mask, p.K, p.M, p.H, p.W);
    }
    ConvConfig cfg;
    const bool h2 = h2_choice(p, &cfg);      // (it reads the ReLU nibbles as the eight-wave fp32 kernel does)
    if (!h2) STX_TRY(choose_conv_config(e, li, 1, p, &cfg, inj != nullptr));   // tuned without the injection terms
    const bool can_fuse = cfg.id != 3 && cfg.id != 4 && cfg.id != 8;   // Winograd ids fuse too  // those two have no injecting epilogue
    if (fused) *fused = inj && can_fuse;
    if (inj && can_fuse) p.inject = *inj;
    b.amax_diff = -1;
    if (pooled) {
        const Blob &pt = e->blobs[pooled->top_blob];
        if (!h2 || !h2_takes_pooled_input(cfg, p) || !pt.codes_valid) {
            set_error("run_conv_backward: %s cannot take the gradient of %s pooled", L.name.c_str(), pt.name.c_str());
            return STX_ERR_UNSUPPORTED;
        }
        p.x = pt.diff.f();
        p.pin_codes = static_cast<const unsigned char *>(pt.codes.ptr);
        p.pin_mode = pooled->pool_mode;
        p.pin_mask = t.relu;
        STX_TRY(amax_for(e, pooled->top_blob, true, &p.x_amax));    // (routing / averaging never raises the maximum)
    } else if (h2) {
        STX_TRY(amax_for(e, L.top_blob, true, &p.x_amax));
    }
    if (h2 || (cfg.id >= 200 && cfg.id < 210 && h2_enabled())) {
        p.y_amax = e->amax_slots(L.bottom_blob, true);
        b.amax_diff = L.bottom_blob;
    }
    const float *packed = nullptr;
    STX_TRY(get_packed(e, li, 1, cfg, &packed));
    p.w = packed;
    STX_TRY(attach_splitk(e, cfg, p));
    ProfScope scope(e, "bwd " + L.name, conv_flops(cp.cout, cp.cin, b.h, b.w, cp.ks));
    return launch_conv(e, cfg, p);
}

// Runs the layers needed for `needed` blobs, in graph order.  `relu_blob` (or -1) is rectified
// even when no ReLU layer follows it (np.maximum(0, .) at style_transfer.py:426,567).
// `after_blob` (optional) is called as soon as a blob is complete, before the next layer is queued.
// `observed` (optional): blobs whose data somebody reads after the pass (taps, requested maps);
// a convolution whose only consumer is a pooling layer fused into it need not store the others.
int forward(stx_engine *e, const std::vector<char> &needed, int relu_blob,
            const std::function<int(int)> *after_blob = nullptr, bool relu_codes = false,
            const std::vector<char> *observed = nullptr) {
    int pooled_layer = -1;      // pooling layer whose output the producing convolution wrote
    // the maxima the fp16-split convolutions leave for each other (Blob::amax_data): none yet
    STX_TRY(e->amax.ensure((2 * e->blobs.size() + 2) * kAmaxSlots * sizeof(unsigned)));
    // (the data slots and, behind them, the diff slots of a backward walk that may follow: one fill)
    STX_HIP(hipMemsetAsync(e->amax_slots(0, false), 0, 2 * e->blobs.size() * kAmaxSlots * sizeof(unsigned), e->stream));
    for (Blob &b : e->blobs) {
        b.amax_data = -1;
        b.relu_codes_valid = false;
    }
    for (size_t li = 1; li < e->layers.size(); ++li) {
        const Layer &L = e->layers[li];
        if (L.type == STX_LAYER_RELU || !needed[L.top_blob]) continue;
        const Blob &b = e->blobs[L.bottom_blob];
        Blob &t = e->blobs[L.top_blob];
        if (L.type == STX_LAYER_CONV) {
            // a 2x2/2 pooling layer fed by this blob (and nothing rectifying the pooled blob,
            // which would have to come after the pooling) can ride on the convolution's epilogue
            const Layer *pool = nullptr;
            int pool_li = -1;
            for (size_t lj = li + 1; lj < e->layers.size(); ++lj) {
                const Layer &P = e->layers[lj];
                if (P.type == STX_LAYER_POOL && P.bottom_blob == L.top_blob && needed[P.top_blob] &&
                    P.ksize == 2 && P.stride == 2 && P.pad == 0 && !e->blobs[P.top_blob].relu &&
                    P.top_blob != relu_blob) {
                    pool = &P;
                    pool_li = (int)lj;
                    break;
                }
            }
            bool pooled = false;
            bool unobserved = false;
            if (pool && observed && !(*observed)[L.top_blob] && L.top_blob != relu_blob) {
                int readers = 0;
                for (size_t lj = li + 1; lj < e->layers.size(); ++lj)
                    readers += e->layers[lj].type != STX_LAYER_RELU && e->layers[lj].bottom_blob == L.top_blob;
                unobserved = readers == 1;
            }
            // a convolution on the path reads this blob: its backward pass masks with the blob's signs
            bool conv_reader = false;
            for (size_t lj = li + 1; lj < e->layers.size(); ++lj)
                conv_reader |= e->layers[lj].type == STX_LAYER_CONV && e->layers[lj].bottom_blob == L.top_blob &&
                               needed[e->layers[lj].top_blob];
            STX_TRY(run_conv_forward(e, (int)li, L.top_blob == relu_blob, pool, &pooled, relu_codes,
                                     unobserved, conv_reader));
            if (pooled) pooled_layer = pool_li;
            if (after_blob) {
                STX_TRY((*after_blob)(L.top_blob));
                if (pooled) STX_TRY((*after_blob)(pool->top_blob));
            }
        } else if ((int)li == pooled_layer) {
            continue;
        } else {
            {
                ProfScope scope(e, "fwd " + L.name, 0.0);
                unsigned char *codes = nullptr;
                t.codes_valid = false;
                if (e->pool_codes) {
                    STX_TRY(t.codes.ensure(t.count() + 4));      // (see run_conv_forward)
                    codes = static_cast<unsigned char *>(t.codes.ptr);
                    t.codes_valid = true;
                }
                STX_TRY(pool_forward_launch(e->stream, b.data.f(), b.channels, b.h, b.w, L.pool_mode,
                                            t.data.f(), codes));
                t.amax_data = b.amax_data;      // (a ReLU behind it only lowers the maximum)
                if (t.relu || L.top_blob == relu_blob)
                    STX_TRY(relu_inplace_launch(e->stream, t.data.f(), t.count()));
            }   // (the loss terms of a tapped pooled blob are timed under their own labels)
            if (after_blob) STX_TRY((*after_blob)(L.top_blob));
        }
    }
    return STX_OK;
}

// S = sym(tril(G - Gs)) F into `sgrad`, sum |S| into *abs_sum (style_transfer.py:587-593).  The
// three-piece bf16 kernel where it applies (STX_SYMM=fp32 keeps the fp32-MFMA 1x1 path).
// Style terms of one tapped blob, the launches of style_transfer.py:584-593 in order: Gram of
// `feat` -> D = G - target (fp32 + bf16 pieces) -> S = sym(D) feat into `sgrad`;
// sc[0] = sum of squares of tril(D), sc[1] = sum |S| (one small launch for both).
// f_amax (or null): the kAmaxSlots words bounding |feat| that its producer left -- the fp16 two-piece
// Gram and SYMM kernels (f16x2.h) scale by them; without them a pass over `feat` comes first.
// term_scratch + defer (or null: sc[0], sc[1] are final when this returns): 2 x gram_finish's blocks +
// the SYMM kernel's workgroups floats that outlive the call (style_term_scratch_floats), and the list
// that receives the two final sums for ONE launch behind the forward pass (sum_jobs_launch).
size_t style_term_scratch_floats(int C, int HW) {
    return 2 * (size_t)ceil_div(C * C, 64) + (size_t)symm_num_workgroups(C, HW) + 64;
}

int launch_style_terms(stx_engine *e, hipStream_t stream, const float *feat, int C, int h, int w,
                       const float *target, float *sgrad, float *sc, const std::string &name,
                       const unsigned *f_amax = nullptr, float *term_scratch = nullptr,
                       std::vector<SumJob> *defer = nullptr) {
    const int HW = h * w;
    // the first layer's kernel may have left this blob's Gram partials already (conv_first.hip)
    const bool fused = e->first_gram_valid && e->first_gram_blob >= 0 &&
                       feat == e->blobs[e->first_gram_blob].data.f() && C == 64;
    GramPlan plan = gram_plan(C, HW);
    if (fused) {
        plan.splits = e->first_gram_parts;
        plan.tiles = 1;
        plan.parts = 1;
        plan.partial_floats = (size_t)plan.splits * 64 * 64;
    }
    const int fin_blocks = gram_finish_blocks(plan);
    float *const partials = fused ? e->first_gram.f() : nullptr;
    // (behind the partial tiles: gram_finish's per-block sums of squares and maxima)
    if (!fused) STX_TRY(e->gram_partials.ensure((plan.partial_floats + 2 * fin_blocks) * sizeof(float)));
    STX_TRY(e->dsym.ensure((size_t)C * C * sizeof(float)));
    const bool gram_h2 = !fused && gram_h2_usable(feat, C, HW);
    const bool symm_h2 = symm_h2_usable(feat, sgrad, C, HW);
    const bool bf3 = !symm_h2 && symm_bf3_usable(feat, sgrad, C, HW);
    if ((gram_h2 || symm_h2) && !f_amax) {
        STX_TRY(e->amax.ensure((2 * e->blobs.size() + 2) * kAmaxSlots * sizeof(unsigned)));
        unsigned *scratch = e->amax_slots((int)e->blobs.size(), true);
        ProfScope scope(e, "absmax " + name, 0.0, stream);
        STX_TRY(absmax_launch(stream, feat, (size_t)C * HW, scratch));
        f_amax = scratch;
    }
    if (bf3) STX_TRY(e->dsym_pieces.ensure(symm_pieces_elems(C) * sizeof(unsigned short)));
    unsigned short *pieces = bf3 && C % 64 == 0 ? static_cast<unsigned short *>(e->dsym_pieces.ptr) : nullptr;
    {
        ProfScope scope(e, "gram " + name, 2.0 * C * C * (double)HW, stream);
        if (!fused) STX_TRY(gram_partials_launch(stream, feat, plan, e->gram_partials.f(), gram_h2 ? f_amax : nullptr));
        STX_TRY(gram_finish_launch(stream, fused ? partials : e->gram_partials.f(), plan, nullptr, target,
                                   e->dsym.f(), nullptr, pieces, gram_h2 ? f_amax : nullptr,
                                   defer ? term_scratch : nullptr));
    }
    ProfScope scope(e, "symm " + name, 2.0 * C * C * (double)HW, stream);
    const float *block_sumsq = defer ? term_scratch : (fused ? partials : e->gram_partials.f()) + plan.partial_floats;
    // the two final sums: now, or as two jobs of the caller's one launch
    auto finish = [&](float *symm_partials, int n_wg) -> int {
        if (!defer) return sum_partials2_launch(stream, block_sumsq, fin_blocks, sc, symm_partials, n_wg, sc + 1);
        defer->push_back(SumJob{block_sumsq, fin_blocks, sc});
        defer->push_back(SumJob{symm_partials, n_wg, sc + 1});
        return STX_OK;
    };
    if (symm_h2 || bf3) {
        const int n_wg = symm_num_workgroups(C, HW);
        float *symm_partials = defer ? term_scratch + 2 * fin_blocks : nullptr;
        if (!defer) {
            STX_TRY(e->symm_partials.ensure((size_t)n_wg * sizeof(float)));
            symm_partials = e->symm_partials.f();
        }
        if (symm_h2)
            STX_TRY(symm_h2_launch(stream, feat, e->dsym.f(), reinterpret_cast<const unsigned *>(block_sumsq + fin_blocks),
                                   fin_blocks, f_amax, sgrad, symm_partials, C, HW));
        else
            STX_TRY(symm_bf3_launch(stream, feat, e->dsym.f(), static_cast<unsigned short *>(e->dsym_pieces.ptr),
                                    pieces != nullptr, sgrad, symm_partials, C, HW));
        return finish(symm_partials, n_wg);
    }
    const ConvConfig cfg = conv_pick_config(1, C, C, h, w);
    const int n_wg = conv_num_workgroups(cfg, C, h, w);
    STX_TRY(e->symm_partials.ensure((size_t)n_wg * sizeof(float)));
    ConvProblem p{};
    p.x = feat;
    p.w = e->dsym.f();
    p.y = sgrad;
    p.partials = e->symm_partials.f();
    p.K = C;
    p.M = C;
    p.H = h;
    p.W = w;
    p.ksize = 1;
    p.epilogue = kEpiSymm;
    STX_TRY(conv_launch(stream, cfg, p, false));
    // (this path keeps its SYMM partials in the engine's shared buffer: its two sums are launched here)
    return sum_partials2_launch(stream, block_sumsq, fin_blocks, sc, e->symm_partials.f(), n_wg, sc + 1);
}

int begin_timing(stx_engine *e) {
    e->ev_cur = (e->ev_cur + 1) % stx_engine::kTimed;
    STX_HIP(hipEventRecord(e->ev_start[e->ev_cur], e->stream));
    e->flop_algorithmic = e->flop_issued = 0;
    return STX_OK;
}

int end_timing(stx_engine *e) {
    STX_HIP(hipEventRecord(e->ev_stop[e->ev_cur], e->stream));
    if (e->ev_recorded < stx_engine::kTimed) ++e->ev_recorded;
    e->timed = true;
    return STX_OK;
}

// Publishes the losses of one arena from its host mirrors (the copies have landed) and empties it.
void publish_arena(stx_engine::ScalarArena &a) {
    for (const PendingLoss &pl : a.pending) {
        double v = 0.0;
        for (const LossTerm &t : pl.terms) v += t.coef * (double)a.host[t.scalar_index];
        for (const LossTerm &t : pl.dterms) v += t.coef * a.dhost[t.scalar_index];
        if (pl.out) *pl.out = v;
    }
    a.pending.clear();
    a.used = 0;
    a.dused = 0;
    a.ticket = 0;
}

int do_sync(stx_engine *e) {
    STX_HIP(hipStreamSynchronize(e->stream));
    // the closed arena (if any) is the older one
    publish_arena(e->arena[e->cur ^ 1]);
    publish_arena(e->arena[e->cur]);
    return STX_OK;
}

// Waits for the streams of every engine that shares e's state (weights or targets are about to be
// replaced under them).  Their pending results stay pending.
int quiesce_members(stx_engine *e) {
    for (stx_engine *m : e->sh->members) STX_HIP(hipStreamSynchronize(m->stream));
    return STX_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
// ---- the switches' snapshot (common.h: sw_env).  Old snapshots are never freed: a thread may still hold a
// pointer into one, and a snapshot is a few hundred bytes.
namespace stx {
namespace {
typedef std::map<std::string, std::string> SwitchMap;
std::atomic<const SwitchMap *> g_switches{nullptr};
std::mutex g_switches_mutex;
extern "C" char **environ;

const SwitchMap *switches_snapshot() {
    auto *m = new SwitchMap;
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "STX_", 4) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (eq) (*m)[std::string(*e, eq - *e)] = eq + 1;
    }
    return m;
}
}  // namespace

void sw_reread() {
    std::lock_guard<std::mutex> lock(g_switches_mutex);
    g_switches.store(switches_snapshot(), std::memory_order_release);
}

const char *sw_env(const char *name) {
    const SwitchMap *m = g_switches.load(std::memory_order_acquire);
    if (!m) {
        std::lock_guard<std::mutex> lock(g_switches_mutex);
        m = g_switches.load(std::memory_order_acquire);
        if (!m) {
            m = switches_snapshot();
            g_switches.store(m, std::memory_order_release);
        }
    }
    auto it = m->find(name);
    return it == m->end() ? nullptr : it->second.c_str();
}
}  // namespace stx

extern "C" {

const char *stx_version(void) { return "libstx 0.1 (gfx950)"; }

int stx_reread_env(void) {
    stx::sw_reread();
    return STX_OK;
}

const char *stx_last_error(void) { return g_error.c_str(); }

int stx_device_count(int *count) {
    if (!count) return STX_ERR_ARG;
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return STX_OK;
}

int stx_device_name(int device, char *buf, size_t buf_len) {
    if (!buf || !buf_len) return STX_ERR_ARG;
    hipDeviceProp_t prop;
    STX_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buf_len, "%s", prop.gcnArchName);
    return STX_OK;
}

}  // extern "C"

// Peer access between the GPUs of the node (tile and target copies of a multi-GPU farm are peer
// reads / writes over xGMI).  Tried once per device; a pair that cannot be enabled is remembered
// and reported once on stderr -- copies between those two GPUs still work (the runtime stages
// them through host memory), they are only slower.  STX_Q_PEERS_WITHOUT_ACCESS counts them.
static std::mutex g_peer_mutex;
static std::map<int, std::vector<int>> g_peers_missing;   // device -> peers without direct access

void enable_peer_access(int device) {
    std::lock_guard<std::mutex> lock(g_peer_mutex);
    if (g_peers_missing.count(device)) return;
    std::vector<int> &missing = g_peers_missing[device];
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (int peer = 0; peer < n_dev; ++peer) {
        if (peer == device) continue;
        int can = 0;
        hipError_t err = hipDeviceCanAccessPeer(&can, device, peer);
        if (err == hipSuccess && can) {
            err = hipDeviceEnablePeerAccess(peer, 0);
            if (err == hipErrorPeerAccessAlreadyEnabled) err = hipSuccess;
        } else if (err == hipSuccess) {
            err = hipErrorPeerAccessUnsupported;
        }
        (void)hipGetLastError();
        if (err != hipSuccess) {
            missing.push_back(peer);
            fprintf(stderr, "libstx: no peer access GPU %d -> GPU %d (%s); copies between them are "
                            "staged by the runtime\n", device, peer, hipGetErrorString(err));
        }
    }
}

int peers_without_access(int device) {
    std::lock_guard<std::mutex> lock(g_peer_mutex);
    auto it = g_peers_missing.find(device);
    return it == g_peers_missing.end() ? 0 : (int)it->second.size();
}

// `share`: the state of an engine on the same GPU to join (weights, packed banks, targets), or null.
static int build_engine(int device, const stx_layer_desc *layers, int n_layers,
                        std::shared_ptr<SharedState> share, stx_engine **out) {
    if (!layers || n_layers < 2 || !out) {
        set_error("stx_engine_create: bad arguments");
        return STX_ERR_ARG;
    }
    if (layers[0].type != STX_LAYER_INPUT || !layers[0].top) {
        set_error("stx_engine_create: layer 0 must be the input layer");
        return STX_ERR_ARG;
    }
    std::unique_ptr<stx_engine> e(new stx_engine);
    e->device = device;
    const bool joined = share != nullptr;
    e->sh = joined ? share : std::make_shared<SharedState>();
    hipError_t err = hipSetDevice(device);
    if (err != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device, hipGetErrorString(err));
        return STX_ERR_HIP;
    }
    auto add_blob = [&](const std::string &name, int channels, int producer) {
        Blob b;
        b.name = name;
        b.channels = channels;
        b.producer = producer;
        e->blob_index[name] = (int)e->blobs.size();
        e->blobs.push_back(std::move(b));
        return (int)e->blobs.size() - 1;
    };
    for (int i = 0; i < n_layers; ++i) {
        const stx_layer_desc &d = layers[i];
        Layer L;
        L.name = d.name ? d.name : "";
        L.bottom = d.bottom ? d.bottom : "";
        L.top = d.top ? d.top : "";
        L.type = d.type;
        L.num_output = d.num_output;
        L.ksize = d.kernel_size;
        L.pad = d.pad;
        L.stride = d.stride;
        L.pool_mode = d.pool_mode;
        if (L.top.empty()) {
            set_error("layer %d (%s) has no top blob", i, L.name.c_str());
            return STX_ERR_ARG;
        }
        if (i == 0) {
            L.top_blob = add_blob(L.top, d.num_output > 0 ? d.num_output : 3, 0);
        } else {
            auto it = e->blob_index.find(L.bottom);
            if (it == e->blob_index.end()) {
                set_error("layer %s: unknown bottom blob '%s'", L.name.c_str(), L.bottom.c_str());
                return STX_ERR_ARG;
            }
            L.bottom_blob = it->second;
            if (L.type == STX_LAYER_RELU) {
                if (L.top != L.bottom) {
                    set_error("layer %s: only in-place ReLU is supported", L.name.c_str());
                    return STX_ERR_UNSUPPORTED;
                }
                L.top_blob = L.bottom_blob;
                e->blobs[L.top_blob].relu = true;
            } else if (L.type == STX_LAYER_CONV) {
                if (!((L.ksize == 3 && L.pad == 1) || (L.ksize == 1 && L.pad == 0))) {
                    set_error("layer %s: only 3x3/pad 1 and 1x1/pad 0 convolutions are supported",
                              L.name.c_str());
                    return STX_ERR_UNSUPPORTED;
                }
                if (e->blob_index.count(L.top)) {
                    set_error("layer %s: top blob '%s' already exists", L.name.c_str(), L.top.c_str());
                    return STX_ERR_UNSUPPORTED;
                }
                L.top_blob = add_blob(L.top, L.num_output, i);
                if (!joined) {
                    ConvParams &cp = e->sh->conv[i];
                    cp.cin = e->blobs[L.bottom_blob].channels;
                    cp.cout = L.num_output;
                    cp.ks = L.ksize;
                }
            } else if (L.type == STX_LAYER_POOL) {
                if (L.ksize != 2 || L.stride != 2 ||
                    (L.pool_mode != STX_POOL_MAX && L.pool_mode != STX_POOL_AVE)) {
                    set_error("layer %s: only 2x2 stride-2 MAX/AVE pooling is supported",
                              L.name.c_str());
                    return STX_ERR_UNSUPPORTED;
                }
                if (e->blob_index.count(L.top)) {
                    set_error("layer %s: top blob '%s' already exists", L.name.c_str(), L.top.c_str());
                    return STX_ERR_UNSUPPORTED;
                }
                L.top_blob = add_blob(L.top, e->blobs[L.bottom_blob].channels, i);
            } else {
                set_error("layer %s: unsupported type %d", L.name.c_str(), L.type);
                return STX_ERR_UNSUPPORTED;
            }
        }
        e->layer_index[L.name] = i;
        e->layers.push_back(std::move(L));
    }
    // scale of every blob: 224 // (blob height for a 224 x 224 input)
    {
        std::vector<int> h224(e->blobs.size(), 224);
        for (size_t li = 1; li < e->layers.size(); ++li) {
            const Layer &L = e->layers[li];
            if (L.type == STX_LAYER_CONV) h224[L.top_blob] = h224[L.bottom_blob];
            if (L.type == STX_LAYER_POOL) h224[L.top_blob] = pooled_len(h224[L.bottom_blob]);
        }
        for (size_t bi = 0; bi < e->blobs.size(); ++bi) e->blobs[bi].scale = 224 / h224[bi];
    }
    enable_peer_access(device);
    STX_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (int i = 0; i < stx_engine::kTimed; ++i) {
        STX_HIP(hipEventCreate(&e->ev_start[i]));
        STX_HIP(hipEventCreate(&e->ev_stop[i]));
    }
    STX_HIP(hipEventCreate(&e->ev_tune0));
    STX_HIP(hipEventCreate(&e->ev_tune1));
    if (const char *env = sw_env("STX_AUTOTUNE")) e->autotune = atoi(env) != 0;
    if (const char *env = sw_env("STX_POOL_CODES")) e->pool_codes = atoi(env) != 0;
    if (const char *env = sw_env("STX_WINOGRAD")) e->winograd = atoi(env) != 0;
    e->scalars_cap = kScalarFloats;
    for (stx_engine::ScalarArena &a : e->arena) {
        STX_TRY(a.scalars.ensure(e->scalars_cap * sizeof(float)));
        STX_HIP(hipHostMalloc(reinterpret_cast<void **>(&a.host), e->scalars_cap * sizeof(float),
                              hipHostMallocDefault));
        STX_TRY(a.dscalars.ensure(e->dscalars_cap * sizeof(double)));
        STX_HIP(hipHostMalloc(reinterpret_cast<void **>(&a.dhost), e->dscalars_cap * sizeof(double),
                              hipHostMallocDefault));
        STX_HIP(hipEventCreateWithFlags(&a.fence, hipEventDisableTiming));
    }
    STX_TRY(e->red_scratch.ensure(4 * 1024 * sizeof(float)));
    {
        std::lock_guard<std::mutex> lock(e->sh->mutex);
        e->sh->members.push_back(e.get());
    }
    *out = e.release();
    return STX_OK;
}

extern "C" {

int stx_engine_create(int device, const stx_layer_desc *layers, int n_layers, stx_engine **out) {
    return build_engine(device, layers, n_layers, nullptr, out);
}

int stx_engine_create_shared(stx_engine *primary, stx_engine **out) {
    if (!primary || !out) {
        set_error("stx_engine_create_shared: bad arguments");
        return STX_ERR_ARG;
    }
    std::vector<stx_layer_desc> descs(primary->layers.size());
    for (size_t i = 0; i < descs.size(); ++i) {
        const Layer &L = primary->layers[i];
        stx_layer_desc &d = descs[i];
        d.name = L.name.c_str();
        d.type = L.type;
        d.bottom = L.bottom.empty() ? nullptr : L.bottom.c_str();
        d.top = L.top.c_str();
        d.num_output = i == 0 ? primary->blobs[L.top_blob].channels : L.num_output;
        d.kernel_size = L.ksize;
        d.pad = L.pad;
        d.stride = L.stride;
        d.pool_mode = L.pool_mode;
    }
    // every filter bank the group has packed so far (and its weights and targets) is complete
    // before the new member's stream can touch it
    STX_TRY(primary->set_device());
    for (stx_engine *m : primary->sh->members) STX_HIP(hipStreamSynchronize(m->stream));
    return build_engine(primary->device, descs.data(), (int)descs.size(), primary->sh, out);
}

void stx_engine_destroy(stx_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &b : e->sgrad_tap) b->release();
    e->marks_buf.release();
    e->amax.release();
    for (Blob &b : e->blobs) {
        b.data.release();
        b.diff.release();
        b.codes.release();
        b.relu_codes.release();
    }
    for (hipEvent_t ev : e->fence_events) (void)hipEventDestroy(ev);
    bool last;
    {
        std::lock_guard<std::mutex> lock(e->sh->mutex);
        auto &m = e->sh->members;
        m.erase(std::remove(m.begin(), m.end(), e), m.end());
        last = m.empty();
    }
    if (last) {     // the shared state goes with its last engine
        for (auto &kv : e->sh->conv) {
            kv.second.w.release();
            kv.second.b.release();
            for (auto &p : kv.second.packed) p.second->release();
        }
        for (auto &c : e->sh->contents) c.feat->release();
        for (auto &s : e->sh->styles) s.gram->release();
    }
    DevBuf *bufs[] = {&e->splitk, &e->gram_partials, &e->gram, &e->dsym, &e->dsym_pieces, &e->symm_partials,
                      &e->upload, &e->red_scratch, &e->first_gram};
    for (DevBuf *b : bufs) b->release();
    for (stx_engine::ScalarArena &a : e->arena) {
        a.scalars.release();
        a.dscalars.release();
        if (a.host) (void)hipHostFree(a.host);
        if (a.dhost) (void)hipHostFree(a.dhost);
        if (a.fence) (void)hipEventDestroy(a.fence);
    }
    for (auto &pe : e->prof) {
        (void)hipEventDestroy(pe.start);
        (void)hipEventDestroy(pe.stop);
    }
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    for (int i = 0; i < stx_engine::kTimed; ++i) {
        if (e->ev_start[i]) (void)hipEventDestroy(e->ev_start[i]);
        if (e->ev_stop[i]) (void)hipEventDestroy(e->ev_stop[i]);
    }
    if (e->ev_tune0) (void)hipEventDestroy(e->ev_tune0);
    if (e->ev_tune1) (void)hipEventDestroy(e->ev_tune1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int stx_set_conv_weights(stx_engine *e, const char *conv_layer, const float *weights,
                         const float *bias, int mem) {
    if (!e || !conv_layer || !weights) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    auto it = e->layer_index.find(conv_layer);
    if (it == e->layer_index.end() || e->layers[it->second].type != STX_LAYER_CONV) {
        set_error("stx_set_conv_weights: '%s' is not a convolution layer", conv_layer);
        return STX_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(e->sh->mutex);
    const bool shared = e->sh->members.size() > 1;
    if (shared) STX_TRY(quiesce_members(e));
    ConvParams &cp = e->sh->conv[it->second];
    const size_t nw = (size_t)cp.cout * cp.cin * cp.ks * cp.ks;
    STX_TRY(cp.w.ensure(nw * sizeof(float)));
    STX_TRY(cp.b.ensure((size_t)cp.cout * sizeof(float)));
    STX_TRY(copy_in(e, cp.w.ptr, weights, mem, nw * sizeof(float)));
    if (bias)
        STX_TRY(copy_in(e, cp.b.ptr, bias, mem, (size_t)cp.cout * sizeof(float)));
    else
        STX_HIP(hipMemsetAsync(cp.b.ptr, 0, (size_t)cp.cout * sizeof(float), e->stream));
    // host buffers may be reused by the caller right away; engines sharing the bank read it from
    // their own streams
    if (mem == STX_HOST || shared) STX_HIP(hipStreamSynchronize(e->stream));
    for (auto &p : cp.packed) p.second->release();
    cp.packed.clear();
    cp.set = true;
    return STX_OK;
}

int stx_sync(stx_engine *e) {
    if (!e) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return do_sync(e);
}

int stx_fence(stx_engine *e, unsigned long long *ticket) {
    if (!e || !ticket) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    stx_engine::ScalarArena &a = e->A();
    STX_HIP(hipEventRecord(a.fence, e->stream));
    a.ticket = e->next_ticket++;
    *ticket = a.ticket;
    e->cur ^= 1;
    stx_engine::ScalarArena &b = e->A();
    if (b.ticket) {      // nobody waited for the arena that is about to be reused: publish it now
        STX_HIP(hipEventSynchronize(b.fence));
        publish_arena(b);
    }
    return STX_OK;
}

int stx_fence_wait(stx_engine *e, unsigned long long ticket) {
    if (!e) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    for (stx_engine::ScalarArena &a : e->arena) {
        if (a.ticket && a.ticket == ticket) {
            STX_HIP(hipEventSynchronize(a.fence));
            publish_arena(a);
        }
    }
    return STX_OK;      // (an older ticket: published long ago)
}

int stx_engine_device(stx_engine *e, int *device) {
    if (!e || !device) return STX_ERR_ARG;
    *device = e->device;
    return STX_OK;
}

int stx_engine_stream(stx_engine *e, void **hip_stream) {
    if (!e || !hip_stream) return STX_ERR_ARG;
    *hip_stream = e->stream;
    return STX_OK;
}

int stx_engine_wait(stx_engine *e, stx_engine *other) {
    if (!e || !other) return STX_ERR_ARG;
    if (e == other) return STX_OK;
    constexpr size_t kRing = 16;    // a wait reads the event's state when it is queued: re-recording later is safe
    STX_TRY(other->set_device());
    if (other->fence_events.size() < kRing) {
        hipEvent_t ev;
        STX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        other->fence_events.push_back(ev);
        other->fence_next = other->fence_events.size() - 1;
    }
    hipEvent_t ev = other->fence_events[other->fence_next];
    other->fence_next = (other->fence_next + 1) % kRing;
    STX_HIP(hipEventRecord(ev, other->stream));
    STX_TRY(e->set_device());
    STX_HIP(hipStreamWaitEvent(e->stream, ev, 0));
    return STX_OK;
}

int stx_engine_query(stx_engine *e, int what, double *value) {
    if (!e || !value) return STX_ERR_ARG;
    std::lock_guard<std::mutex> lock(e->sh->mutex);
    switch (what) {
        case STX_Q_SHARED_ENGINES: *value = (double)e->sh->members.size(); break;
        case STX_Q_TARGET_UPLOADS: *value = (double)e->sh->target_uploads; break;
        case STX_Q_TARGET_BYTES: *value = e->sh->target_bytes; break;
        case STX_Q_WEIGHT_BYTES: {
            double b = 0;
            for (auto &kv : e->sh->conv) {
                b += (double)kv.second.w.bytes + (double)kv.second.b.bytes;
                for (auto &p : kv.second.packed) b += (double)p.second->bytes;
            }
            *value = b;
            break;
        }
        case STX_Q_TILE_EVALS: *value = (double)e->n_tile_evals; break;
        case STX_Q_PEERS_WITHOUT_ACCESS: *value = (double)peers_without_access(e->device); break;
        default:
            set_error("stx_engine_query: unknown item %d", what);
            return STX_ERR_ARG;
    }
    return STX_OK;
}

int stx_malloc(stx_engine *e, size_t bytes, void **dev_ptr) {
    if (!e || !dev_ptr) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    hipError_t err = hipMalloc(dev_ptr, bytes ? bytes : 4);
    if (err != hipSuccess) {
        set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(err));
        return STX_ERR_NOMEM;
    }
    return STX_OK;
}

int stx_free(stx_engine *e, void *dev_ptr) {
    if (!e) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    if (dev_ptr) STX_HIP(hipFree(dev_ptr));
    return STX_OK;
}

int stx_memset_async(stx_engine *e, void *dev_ptr, int value, size_t bytes) {
    if (!e || !dev_ptr) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    STX_HIP(hipMemsetAsync(dev_ptr, value, bytes, e->stream));
    return STX_OK;
}

int stx_memcpy_async(stx_engine *e, void *dst, int dst_mem, const void *src, int src_mem,
                     size_t bytes) {
    if (!e || !dst || !src) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    hipMemcpyKind kind = hipMemcpyDefault;
    if (dst_mem == STX_HOST && src_mem == STX_HOST) kind = hipMemcpyHostToHost;
    if (dst_mem == STX_HOST && src_mem == STX_DEVICE) kind = hipMemcpyDeviceToHost;
    if (dst_mem == STX_DEVICE && src_mem == STX_HOST) kind = hipMemcpyHostToDevice;
    STX_HIP(hipMemcpyAsync(dst, src, bytes, kind, e->stream));
    return STX_OK;
}

int stx_set_contents_and_styles(stx_engine *e, const stx_content_target *contents, int n_contents,
                                const stx_style_target *styles, int n_styles) {
    if (!e || (n_contents && !contents) || (n_styles && !styles)) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    std::lock_guard<std::mutex> lock(e->sh->mutex);
    const bool shared = e->sh->members.size() > 1;
    // the previous targets may still be in use by queued kernels (of any engine that shares them)
    STX_TRY(quiesce_members(e));
    double copied = 0;
    for (auto &c : e->sh->contents) c.feat->release();
    for (auto &s : e->sh->styles) s.gram->release();
    e->sh->contents.clear();
    e->sh->styles.clear();
    e->sh->n_contents = e->sh->n_styles = 0;
    bool host_src = false;
    for (int i = 0; i < n_contents; ++i) {
        const stx_content_target &c = contents[i];
        const int blob = e->find_blob(c.layer);
        if (blob < 0 || !c.features || c.channels != e->blobs[blob].channels || c.height <= 0 ||
            c.width <= 0 || c.content_index < 0) {
            set_error("content target %d: bad layer '%s' or shape", i, c.layer ? c.layer : "(null)");
            return STX_ERR_ARG;
        }
        ContentTarget t;
        t.index = c.content_index;
        t.blob = blob;
        t.C = c.channels;
        t.h = c.height;
        t.w = c.width;
        t.feat.reset(new DevBuf);
        const size_t bytes = (size_t)t.C * t.h * t.w * sizeof(float);
        STX_TRY(t.feat->ensure(bytes));
        STX_TRY(copy_in(e, t.feat->ptr, c.features, c.mem, bytes));
        copied += (double)bytes;
        host_src |= c.mem == STX_HOST;
        e->sh->n_contents = std::max(e->sh->n_contents, t.index + 1);
        e->sh->contents.push_back(std::move(t));
    }
    for (int i = 0; i < n_styles; ++i) {
        const stx_style_target &s = styles[i];
        const int blob = e->find_blob(s.layer);
        if (blob < 0 || !s.gram || s.channels != e->blobs[blob].channels || s.style_index < 0) {
            set_error("style target %d: bad layer '%s' or shape", i, s.layer ? s.layer : "(null)");
            return STX_ERR_ARG;
        }
        StyleTarget t;
        t.index = s.style_index;
        t.blob = blob;
        t.C = s.channels;
        t.gram.reset(new DevBuf);
        const size_t bytes = (size_t)t.C * t.C * sizeof(float);
        STX_TRY(t.gram->ensure(bytes));
        STX_TRY(copy_in(e, t.gram->ptr, s.gram, s.mem, bytes));
        copied += (double)bytes;
        host_src |= s.mem == STX_HOST;
        e->sh->n_styles = std::max(e->sh->n_styles, t.index + 1);
        e->sh->styles.push_back(std::move(t));
    }
    // (the sharing engines use the new targets from their own streams)
    if (host_src || shared) STX_HIP(hipStreamSynchronize(e->stream));
    e->sh->target_uploads += 1;
    e->sh->target_bytes += copied;
    return STX_OK;
}

int stx_features_tile(stx_engine *e, const float *img, int img_mem, int th, int tw,
                      const char *const *layers, int n_layers, float *const *out, int out_mem) {
    if (!e || !img || th <= 0 || tw <= 0 || n_layers <= 0 || !layers || !out) {
        set_error("stx_features_tile: bad arguments");
        return STX_ERR_ARG;
    }
    STX_TRY(e->set_device());
    std::vector<char> needed(e->blobs.size(), 0);
    std::vector<int> want(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        want[i] = e->find_blob(layers[i]);
        if (want[i] < 0 || !out[i]) {
            set_error("stx_features_tile: unknown layer '%s'", layers[i] ? layers[i] : "(null)");
            return STX_ERR_ARG;
        }
        mark_ancestors(e, want[i], needed);
    }
    STX_TRY(shape_blobs(e, th, tw, needed, false));
    Blob &in = e->blobs[e->layers[0].top_blob];
    STX_TRY(copy_in(e, in.data.ptr, img, img_mem, in.count() * sizeof(float)));
    STX_TRY(begin_timing(e));
    e->first_gram_blob = -1;       // (no loss terms here: the first layer computes no Gram partials)
    e->first_gram_valid = false;
    // the reference rectifies the net's last blob (style_transfer.py:426)
    const int last_blob = (int)e->blobs.size() - 1;
    std::vector<char> observed(e->blobs.size(), 0);
    for (int i = 0; i < n_layers; ++i) observed[want[i]] = 1;
    STX_TRY(forward(e, needed, needed[last_blob] ? last_blob : -1, nullptr, false, &observed));
    STX_TRY(end_timing(e));
    for (int i = 0; i < n_layers; ++i) {
        const Blob &b = e->blobs[want[i]];
        STX_TRY(copy_out(e, out[i], out_mem, b.data.ptr, b.count() * sizeof(float)));
    }
    return STX_OK;
}

}  // extern "C"

namespace {

struct Tap {
    int blob;
    const stx_tap *t;
};

// One stx_sc_grad_tile call.
struct TileCall {
    const float *img;
    int img_mem, th, tw, rx, ry, start[2];
    const stx_tap *taps;
    int n_taps;
    float *grad_out;
    int grad_mem;
};

struct TilePlan {
    std::vector<Tap> order;         // taps, deepest first
    std::vector<char> needed;       // blobs on the path
    std::vector<int> tap_of;        // blob -> index into order, or -1
};

// Validates the taps against the graph and the targets, orders them and shapes the blobs.
int sc_grad_prepare(stx_engine *e, const TileCall &c, TilePlan &plan) {
    // ---- taps in deep -> shallow order (style_transfer.py:231-233)
    std::vector<Tap> &order = plan.order;
    const stx_tap *taps = c.taps;
    const int n_taps = c.n_taps;
    for (int i = 0; i < n_taps; ++i) {
        const int blob = e->find_blob(taps[i].layer);
        if (blob <= 0) {
            set_error("stx_sc_grad_tile: unknown tap layer '%s'",
                      taps[i].layer ? taps[i].layer : "(null)");
            return STX_ERR_ARG;
        }
        for (const Tap &o : order)
            if (o.blob == blob) {
                set_error("stx_sc_grad_tile: layer '%s' is tapped twice", taps[i].layer);
                return STX_ERR_ARG;
            }
        if (!taps[i].is_content && !taps[i].is_style && !taps[i].is_dd) continue;
        order.push_back(Tap{blob, &taps[i]});
    }
    if (order.empty()) {
        set_error("stx_sc_grad_tile: no content, style or Deep-Dream layer");
        return STX_ERR_ARG;
    }
    std::sort(order.begin(), order.end(), [](const Tap &a, const Tap &b) { return a.blob > b.blob; });
    std::vector<char> &needed = plan.needed;
    needed.assign(e->blobs.size(), 0);
    mark_ancestors(e, order[0].blob, needed);
    std::vector<int> &tap_of = plan.tap_of;
    tap_of.assign(e->blobs.size(), -1);
    for (size_t i = 0; i < order.size(); ++i) {
        if (!needed[order[i].blob]) {
            set_error("stx_sc_grad_tile: tapped layers must lie on one path through the network "
                      "('%s' does not feed '%s')", e->blobs[order[i].blob].name.c_str(),
                      e->blobs[order[0].blob].name.c_str());
            return STX_ERR_UNSUPPORTED;
        }
        tap_of[order[i].blob] = (int)i;
    }
    for (const Tap &tp : order) {
        if (tp.t->is_content && e->sh->n_contents == 0) {
            set_error("stx_sc_grad_tile: no content targets set");
            return STX_ERR_STATE;
        }
        if (tp.t->is_style && e->sh->n_styles == 0) {
            set_error("stx_sc_grad_tile: no style targets set");
            return STX_ERR_STATE;
        }
    }

    return shape_blobs(e, c.th, c.tw, needed, true);
}

// Enqueues the evaluation proper: forward pass with the loss terms of the tapped blobs, backward
// walk, the mirror copy of the loss scalars.  The tile is already in the input blob; the gradient
// is left in its diff.
int sc_grad_run(stx_engine *e, const TileCall &c, const TilePlan &plan, PendingLoss &pl) {
    const std::vector<Tap> &order = plan.order;
    const std::vector<char> &needed = plan.needed;
    const std::vector<int> &tap_of = plan.tap_of;
    const int data_blob = e->layers[0].top_blob;

    // ---- loss terms of the tapped blobs
    struct Term {
        bool style;
        const float *src;        // style: S = sym(tril(G - Gs)) F;  content: the content map
        const float *sums;       // style: &sum|S|;  content: {sum c^2, sum |c|}
        float coef;
        ContentWindow win;
    };
    std::vector<std::vector<Term>> terms(order.size());
    while (e->sgrad_tap.size() < order.size()) e->sgrad_tap.emplace_back(new DevBuf);
    // The final sums of the loss terms (two per style term, two per content term) are collected and run
    // as ONE launch behind the forward pass (STX_SUMS_LATE=0: each where it arises, as rounds 1-4 did);
    // what they add up must outlive the term's own launches: one scratch region per style term.
    const bool sums_late = !(sw_env("STX_SUMS_LATE") && !atoi(sw_env("STX_SUMS_LATE")));
    std::vector<SumJob> sum_jobs;
    size_t scratch_used = 0;
    if (sums_late) {
        size_t need = 0;
        for (const Tap &tp : order) {
            if (!tp.t->is_style) continue;
            const Blob &b = e->blobs[tp.blob];
            for (const StyleTarget &st : e->sh->styles)
                if (st.blob == tp.blob) need += style_term_scratch_floats(b.channels, b.h * b.w);
        }
        STX_TRY(e->term_scratch.ensure(need * sizeof(float)));
    }
    // Loss terms of tap k (Gram -> G - Gs -> SYMM, content residual sums).  They are queued the
    // moment the tapped blob is complete, in the middle of the forward pass, while the blob is
    // still in the L2 / Infinity Cache the convolution just wrote it through (the shallow blobs
    // were re-fetched from HBM when all taps ran after the forward pass: 1.1 GB per tile by PMC).
    auto launch_terms = [&](size_t k) -> int {
        const Tap &tp = order[k];
        Blob &b = e->blobs[tp.blob];
        const double lw = tp.t->layer_weight;
        if (tp.t->is_content) {
            bool any = false;
            for (const ContentTarget &ct : e->sh->contents) {
                if (ct.blob != tp.blob) continue;
                any = true;
                ContentWindow win;
                win.C = b.channels;
                win.fh = b.h;
                win.fw = b.w;
                win.ch = ct.h;
                win.cw = ct.w;
                // start_ = start // scale (style_transfer.py:572); roll // scale per layer (:647-655)
                win.oy = (int)std::floor((double)c.start[0] / b.scale);
                win.ox = (int)std::floor((double)c.start[1] / b.scale);
                win.sx = (int)std::floor((double)c.rx / b.scale);
                win.sy = (int)std::floor((double)c.ry / b.scale);
                if (win.oy + win.fh > win.ch || win.ox + win.fw > win.cw) {
                    set_error("content window [%d+%d, %d+%d] exceeds the %dx%d map of layer %s",
                              win.oy, win.fh, win.ox, win.fw, win.ch, win.cw, b.name.c_str());
                    return STX_ERR_ARG;
                }
                size_t si;
                STX_TRY(alloc_scalars(e, 2 + 2 * 1024, &si));
                float *sums = e->A().scalars.f() + si;
                {
                    ProfScope scope(e, "content " + b.name, 0.0, e->stream);
                    STX_TRY(content_sums_launch(e->stream, b.data.f(), ct.feat->f(), win, sums,
                                                sums_late ? &sum_jobs : nullptr));
                }
                pl.terms.push_back(LossTerm{si, lw * tp.t->content_weight * 0.5});
                terms[k].push_back(Term{false, ct.feat->f(), sums,
                                        (float)(lw * tp.t->content_weight), win});
            }
            if (!any) {
                set_error("no content target for layer %s", b.name.c_str());
                return STX_ERR_STATE;
            }
        }
        if (tp.t->is_style) {
            int n_here = 0;
            for (const StyleTarget &st : e->sh->styles) n_here += st.blob == tp.blob;
            if (!n_here) {
                set_error("no style target for layer %s", b.name.c_str());
                return STX_ERR_STATE;
            }
            STX_TRY(e->sgrad_tap[k]->ensure((size_t)n_here * b.count() * sizeof(float)));
            int slot = 0;
            for (const StyleTarget &st : e->sh->styles) {
                if (st.blob != tp.blob) continue;
                const int C = b.channels, HW = b.h * b.w;
                if (C % 4 != 0) {
                    set_error("style layer %s: channel count %d is not a multiple of 4", b.name.c_str(),
                              C);
                    return STX_ERR_UNSUPPORTED;
                }
                float *sgrad = e->sgrad_tap[k]->f() + (size_t)slot++ * b.count();
                size_t si;
                STX_TRY(alloc_scalars(e, 2, &si));
                float *sc = e->A().scalars.f() + si;   // [0] = sum tril(D)^2, [1] = sum |S|
                // (the maximum the blob's producer left, if it left one: the fp16-split kernels' scale)
                const unsigned *f_amax = b.amax_data >= 0 ? e->amax_slots(b.amax_data, false) : nullptr;
                float *scratch = nullptr;
                if (sums_late) {
                    scratch = e->term_scratch.f() + scratch_used;
                    scratch_used += style_term_scratch_floats(C, HW);
                }
                STX_TRY(launch_style_terms(e, e->stream, b.data.f(), C, b.h, b.w, st.gram->f(), sgrad, sc,
                                           b.name, f_amax, scratch, sums_late ? &sum_jobs : nullptr));
                (void)HW;
                pl.terms.push_back(LossTerm{si, lw * tp.t->style_weight * 0.5 / e->sh->n_styles});
                terms[k].push_back(Term{true, sgrad, sc + 1,
                                        (float)(lw * tp.t->style_weight / e->sh->n_styles), ContentWindow{}});
            }
        }
        if (tp.t->is_dd) {
            // Deep-Dream term (style_transfer.py:602-604): the content term against a zero map with
            // a negative weight -- loss -= lw*dd*1/2|F|^2, diff -= lw*dd*normalize(F)
            ContentWindow win{};
            win.C = b.channels;
            win.fh = win.ch = b.h;
            win.fw = win.cw = b.w;
            size_t si;
            STX_TRY(alloc_scalars(e, 2 + 2 * 1024, &si));
            float *sums = e->A().scalars.f() + si;
            {
                ProfScope scope(e, "dream " + b.name, 0.0, e->stream);
                STX_TRY(content_sums_launch(e->stream, b.data.f(), nullptr, win, sums,
                                            sums_late ? &sum_jobs : nullptr));
            }
            pl.terms.push_back(LossTerm{si, -lw * tp.t->dd_weight * 0.5});
            terms[k].push_back(Term{false, nullptr, sums, (float)(-lw * tp.t->dd_weight), win});
        }
        return STX_OK;
    };
    // (STX_TERMS_LATE=1: all loss terms after the forward pass, for A/B measurements)
    const bool interleave = !(sw_env("STX_TERMS_LATE") && atoi(sw_env("STX_TERMS_LATE")));
    const std::function<int(int)> hook = [&](int blob) -> int {
        const int k = tap_of[blob];
        return k >= 0 ? launch_terms((size_t)k) : STX_OK;
    };
    STX_TRY(begin_timing(e));
    std::vector<char> observed(e->blobs.size(), 0);
    for (const Tap &tp : order) observed[tp.blob] = 1;
    // a style tap on the first layer's blob: that layer's kernel leaves its Gram partials
    e->first_gram_blob = -1;
    e->first_gram_valid = false;
    for (const Tap &tp : order) {
        const int pl = e->blobs[tp.blob].producer;
        if (tp.t->is_style && pl > 0 && e->layers[pl].type == STX_LAYER_CONV &&
            e->layers[pl].bottom_blob == data_blob && e->blobs[tp.blob].channels == 64)
            e->first_gram_blob = tp.blob;
    }
    STX_TRY(forward(e, needed, order[0].blob, interleave ? &hook : nullptr, true, &observed));
    if (!interleave) {
        // (shallowest tap first, the order the interleaved schedule queues them in: the host adds
        // the loss terms up in queueing order, in double precision, and must get the same bits)
        for (size_t k = order.size(); k-- > 0;) STX_TRY(launch_terms(k));
    }
    if (!sum_jobs.empty()) {
        ProfScope scope(e, "sums", 0.0);
        STX_TRY(sum_jobs_launch(e->stream, sum_jobs.data(), (int)sum_jobs.size()));
    }

    // Adds the terms of tap k to its blob's diff with stand-alone kernels (used for the deepest
    // tap, for blobs produced by a pooling backward, and when a tap has more than one content or
    // style term; otherwise the terms ride in the epilogue of the convolution backward above).
    auto inject = [&](size_t k, bool &diff_written) -> int {
        Blob &b = e->blobs[order[k].blob];
        ProfScope scope(e, "inject " + b.name, 0.0);
        b.amax_diff = -1;
        for (size_t ti = 0; ti < terms[k].size(); ++ti) {       // content terms come first, like the reference
            const Term &t = terms[k][ti];
            // the last term's kernel writes the blob's final gradient: it leaves its maximum for the
            // fp16-split convolution that reads it next (the slots were zeroed when the walk began)
            unsigned *amax = nullptr;
            if (ti + 1 == terms[k].size() && h2_enabled()) {
                amax = e->amax_slots(order[k].blob, true);
                b.amax_diff = order[k].blob;
            }
            if (t.style)
                STX_TRY(inject_style_launch(e->stream, b.diff.f(), t.src, b.count(), t.sums, t.coef,
                                            diff_written, amax));
            else
                STX_TRY(inject_content_launch(e->stream, b.diff.f(), b.data.f(), t.src, t.win,
                                              t.sums, t.coef, diff_written, amax));
            diff_written = true;
        }
        return STX_OK;
    };
    auto fusable = [&](size_t k) {
        int ns = 0, nc = 0;
        for (const Term &t : terms[k]) {
            if (!t.style && !t.src) return false;      // Deep-Dream terms take the stand-alone path
            (t.style ? ns : nc)++;
        }
        return ns <= 1 && nc <= 1;
    };

    // ---- backward walk from the deepest tap to the image (style_transfer.py:569-610)
    int cur = order[0].blob;
    // (the diff slots were zeroed with the data slots when the forward pass began)
    for (Blob &b : e->blobs) b.amax_diff = -1;
    {
        bool written = false;
        STX_TRY(inject(0, written));
        if (!written)
            STX_HIP(hipMemsetAsync(e->blobs[cur].diff.ptr, 0, e->blobs[cur].count() * sizeof(float),
                                   e->stream));
    }
    const Layer *pooled = nullptr;      // a pooling layer whose backward pass rides in the next convolution's
    while (cur != data_blob) {
        const int li = e->blobs[cur].producer;
        const Layer &L = e->layers[li];
        Blob &bot = e->blobs[L.bottom_blob];
        const Blob &top = e->blobs[cur];
        const int k = tap_of[L.bottom_blob];
        bool fused = false;
        if (L.type == STX_LAYER_POOL && top.codes_valid && k < 0 && L.ksize == 2 && L.stride == 2 && L.pad == 0 &&
            e->layers[bot.producer].type == STX_LAYER_CONV && conv_backward_takes_pooled(e, bot.producer)) {
            // the convolution under the pooling layer un-pools inside its patch staging: nothing to launch,
            // the gradient of `bot` never exists (nobody else wants it: no loss term taps that blob)
            pooled = &L;
            cur = L.bottom_blob;
            continue;
        }
        if (L.type == STX_LAYER_CONV) {
            ConvInject inj{};
            if (k >= 0 && fusable((size_t)k)) {
                for (const Term &t : terms[k]) {
                    if (t.style) {
                        inj.sgrad = t.src;
                        inj.s_abs_sum = t.sums;
                        inj.s_coef = t.coef;
                    } else {
                        inj.content = t.src;
                        inj.c_sums = t.sums;
                        inj.c_coef = t.coef;
                        inj.win = t.win;
                        inj.feat = bot.data.f();
                    }
                }
                fused = true;
            }
            STX_TRY(run_conv_backward(e, li, fused ? &inj : nullptr, &fused, pooled));
            pooled = nullptr;
        } else {
            ProfScope scope(e, "bwd " + L.name, 0.0);
            if (top.codes_valid)
                STX_TRY(pool_backward_codes_launch(
                    e->stream, top.diff.f(), static_cast<const unsigned char *>(top.codes.ptr),
                    bot.channels, bot.h, bot.w, L.pool_mode, bot.relu, bot.diff.f()));
            else
                STX_TRY(pool_backward_launch(e->stream, top.diff.f(), bot.data.f(), bot.channels,
                                             bot.h, bot.w, L.pool_mode, bot.relu, bot.diff.f()));
            bot.amax_diff = top.amax_diff;     // routing / averaging never raises the maximum
        }
        cur = L.bottom_blob;
        if (k >= 0 && !fused) {
            bool written = true;   // the upstream gradient is already in diff
            // (the terms are added behind the kernel that left a maximum; the slots hold that one, and
            // max is monotone: zero them so that the last term's kernel leaves the new one)
            STX_HIP(hipMemsetAsync(e->amax_slots(L.bottom_blob, true), 0, kAmaxSlots * sizeof(unsigned), e->stream));
            STX_TRY(inject((size_t)k, written));
        }
    }
    STX_TRY(end_timing(e));
    // mirror the scalars used so far (small) for the loss
    STX_HIP(hipMemcpyAsync(e->A().host, e->A().scalars.ptr, e->A().used * sizeof(float),
                           hipMemcpyDeviceToHost, e->stream));
    return STX_OK;
}

int sc_grad_eager(stx_engine *e, const TileCall &c, double *loss_out) {
    // the scalar arena holds the reductions of every call queued since the last stx_sync; drain
    // it (publishing the pending losses) before it could overflow
    {
        const size_t per_call = (size_t)c.n_taps * 2100 *
                                (size_t)std::max(1, e->sh->n_contents + e->sh->n_styles);
        if (e->A().used + per_call > e->scalars_cap) STX_TRY(do_sync(e));
        if (per_call > e->scalars_cap) {
            set_error("stx_sc_grad_tile: %d taps need more scalar space than the arena holds", c.n_taps);
            return STX_ERR_NOMEM;
        }
    }
    TilePlan plan;
    STX_TRY(sc_grad_prepare(e, c, plan));
    Blob &in = e->blobs[e->layers[0].top_blob];
    // (a tile handed over in the engine's own buffers, stx_tile_buffers, needs no copies)
    if (c.img != in.data.ptr) STX_TRY(copy_in(e, in.data.ptr, c.img, c.img_mem, in.count() * sizeof(float)));
    PendingLoss pl;
    pl.out = loss_out;
    STX_TRY(sc_grad_run(e, c, plan, pl));
    if (c.grad_out != in.diff.ptr)
        STX_TRY(copy_out(e, c.grad_out, c.grad_mem, in.diff.ptr, in.count() * sizeof(float)));
    e->A().pending.push_back(std::move(pl));
    ++e->n_tile_evals;
    return STX_OK;
}

}  // namespace

extern "C" {

int stx_sc_grad_tile(stx_engine *e, const float *img, int img_mem, int th, int tw,
                     const int roll_xy[2], const int start_yx[2], const stx_tap *taps, int n_taps,
                     double *loss_out, float *grad_out, int grad_mem, int sync_now) {
    if (!e || !img || th <= 0 || tw <= 0 || !taps || n_taps <= 0 || !grad_out || !start_yx) {
        set_error("stx_sc_grad_tile: bad arguments");
        return STX_ERR_ARG;
    }
    STX_TRY(e->set_device());
    const TileCall c{img, img_mem, th, tw, roll_xy ? roll_xy[0] : 0, roll_xy ? roll_xy[1] : 0,
                     {start_yx[0], start_yx[1]}, taps, n_taps, grad_out, grad_mem};
    STX_TRY(sc_grad_eager(e, c, loss_out));
    if (sync_now) return do_sync(e);
    return STX_OK;
}

int stx_tile_buffers(stx_engine *e, int th, int tw, float **tile_in, float **grad_out) {
    if (!e || th <= 0 || tw <= 0 || !tile_in || !grad_out) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    Blob &in = e->blobs[e->layers[0].top_blob];
    const size_t bytes = (size_t)in.channels * th * tw * sizeof(float);
    STX_TRY(in.data.ensure(bytes));
    STX_TRY(in.diff.ensure(bytes));
    *tile_in = in.data.f();
    *grad_out = in.diff.f();
    return STX_OK;
}

int stx_gram_matrix(stx_engine *e, const float *feat, int feat_mem, int channels, int hw,
                    float *gram_out, int gram_mem) {
    if (!e || !feat || !gram_out || channels <= 0 || hw <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const float *src = feat;
    if (feat_mem == STX_HOST) {
        STX_TRY(e->upload.ensure((size_t)channels * hw * sizeof(float)));
        STX_TRY(copy_in(e, e->upload.ptr, feat, STX_HOST, (size_t)channels * hw * sizeof(float)));
        src = e->upload.f();
    }
    const GramPlan plan = gram_plan(channels, hw);
    STX_TRY(e->gram_partials.ensure(plan.partial_floats * sizeof(float)));
    STX_TRY(e->gram.ensure((size_t)channels * channels * sizeof(float)));
    const unsigned *f_amax = nullptr;
    if (gram_h2_usable(src, channels, hw)) {       // the fp16 two-piece kernel: scaled by the array's maximum
        STX_TRY(e->amax.ensure((2 * e->blobs.size() + 2) * kAmaxSlots * sizeof(unsigned)));
        unsigned *scratch = e->amax_slots((int)e->blobs.size(), true);
        STX_TRY(absmax_launch(e->stream, src, (size_t)channels * hw, scratch));
        f_amax = scratch;
    }
    STX_TRY(gram_partials_launch(e->stream, src, plan, e->gram_partials.f(), f_amax));
    STX_TRY(gram_finish_launch(e->stream, e->gram_partials.f(), plan, e->gram.f(), nullptr, nullptr,
                               nullptr, nullptr, f_amax));
    STX_TRY(copy_out(e, gram_out, gram_mem, e->gram.ptr, (size_t)channels * channels * sizeof(float)));
    if (feat_mem == STX_HOST || gram_mem == STX_HOST) STX_HIP(hipStreamSynchronize(e->stream));
    return STX_OK;
}

// ------------------------------------------------------------------------------- image ops
int stx_image_cut_tile(stx_engine *e, const float *img, int H, int W, const int roll_xy[2], int y0,
                       int x0, int th, int tw, float *tile) {
    if (!e || !img || !tile || H <= 0 || W <= 0 || th <= 0 || tw <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return cut_tile_launch(e->stream, img, H, W, roll_xy ? roll_xy[0] : 0, roll_xy ? roll_xy[1] : 0,
                           y0, x0, th, tw, tile);
}

int stx_image_put_tile(stx_engine *e, float *grad, int H, int W, const int roll_xy[2], int y0,
                       int x0, int th, int tw, const float *tile_grad) {
    if (!e || !grad || !tile_grad || H <= 0 || W <= 0 || th <= 0 || tw <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return put_tile_launch(e->stream, grad, H, W, roll_xy ? roll_xy[0] : 0,
                           roll_xy ? roll_xy[1] : 0, y0, x0, th, tw, tile_grad);
}

int stx_map_place(stx_engine *e, float *dst, int channels, int dst_h, int dst_w, int y0, int x0,
                  const float *src, int h, int w) {
    if (!e || !dst || !src || channels <= 0 || h <= 0 || w <= 0 || y0 < 0 || x0 < 0 ||
        y0 + h > dst_h || x0 + w > dst_w) {
        set_error("stx_map_place: window [%d+%d, %d+%d] does not fit a %dx%d map", y0, h, x0, w,
                  dst_h, dst_w);
        return STX_ERR_ARG;
    }
    STX_TRY(e->set_device());
    return place_window_launch(e->stream, dst, dst_h, dst_w, y0, x0, src, channels, h, w);
}

int stx_map_roll_add(stx_engine *e, float *acc, const float *src, int channels, int h, int w,
                     const int roll_xy[2], double alpha, double init_divisor) {
    if (!e || !acc || !src || channels <= 0 || h <= 0 || w <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const bool init = init_divisor != 0.0;
    return roll_add_launch(e->stream, acc, src, channels, h, w, roll_xy ? roll_xy[0] : 0,
                           roll_xy ? roll_xy[1] : 0, (float)(init ? init_divisor : alpha), init);
}

int stx_image_resample(stx_engine *e, const float *src, int channels, int H, int W, float *dst,
                       int out_h, int out_w, const int *bounds_x, const double *weights_x,
                       int ksize_x, const int *bounds_y, const double *weights_y, int ksize_y,
                       int clamp_min_zero) {
    if (!e || !src || !dst || !bounds_x || !weights_x || !bounds_y || !weights_y || channels <= 0 ||
        H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || ksize_x <= 0 || ksize_y <= 0)
        return STX_ERR_ARG;
    STX_TRY(e->set_device());
    // device scratch: [bounds_x | bounds_y] ints, [kx | ky] doubles, horizontal-pass image
    const size_t nbx = 2 * (size_t)out_w, nby = 2 * (size_t)out_h;
    const size_t nkx = (size_t)out_w * ksize_x, nky = (size_t)out_h * ksize_y;
    const size_t tmp_floats = (size_t)channels * H * out_w;
    const size_t k_off = ((nbx + nby) * sizeof(int) + 7) & ~(size_t)7;
    const size_t t_off = (k_off + (nkx + nky) * sizeof(double) + 255) & ~(size_t)255;
    STX_TRY(e->upload.ensure(t_off + tmp_floats * sizeof(float)));
    char *base = static_cast<char *>(e->upload.ptr);
    int *d_bx = reinterpret_cast<int *>(base), *d_by = d_bx + nbx;
    double *d_kx = reinterpret_cast<double *>(base + k_off), *d_ky = d_kx + nkx;
    float *tmp = reinterpret_cast<float *>(base + t_off);
    STX_HIP(hipMemcpyAsync(d_bx, bounds_x, nbx * sizeof(int), hipMemcpyHostToDevice, e->stream));
    STX_HIP(hipMemcpyAsync(d_by, bounds_y, nby * sizeof(int), hipMemcpyHostToDevice, e->stream));
    STX_HIP(hipMemcpyAsync(d_kx, weights_x, nkx * sizeof(double), hipMemcpyHostToDevice, e->stream));
    STX_HIP(hipMemcpyAsync(d_ky, weights_y, nky * sizeof(double), hipMemcpyHostToDevice, e->stream));
    // Pillow runs the horizontal pass first, then the vertical pass on its float32 result
    STX_TRY(resample_launch(e->stream, 0, src, channels, H, W, tmp, H, out_w, d_bx, d_kx, ksize_x, 0));
    STX_TRY(resample_launch(e->stream, 1, tmp, channels, H, out_w, dst, out_h, out_w, d_by, d_ky,
                            ksize_y, clamp_min_zero));
    // the coefficient tables are host memory of the caller: finish the copies before returning
    STX_HIP(hipStreamSynchronize(e->stream));
    return STX_OK;
}

int stx_image_regularizers(stx_engine *e, const float *img, float *grad, int H, int W,
                           const float mean_bgr[3], double tv_scale, double tv_power, double p_scale,
                           double p_power, const float *aux, double aux_scale,
                           const int aux_roll_xy[2], double *loss_out) {
    if (!e || !img || !grad || !mean_bgr || H <= 0 || W <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    size_t di;
    STX_TRY(alloc_dscalars(e, 3, &di));
    double *terms = static_cast<double *>(e->A().dscalars.ptr) + di;
    STX_TRY(regularizers_launch(e->stream, img, grad, H, W, mean_bgr, (float)tv_scale,
                                (float)tv_power, (float)p_scale, (float)p_power, aux,
                                (float)aux_scale, aux_roll_xy ? aux_roll_xy[0] : 0,
                                aux_roll_xy ? aux_roll_xy[1] : 0, terms, e->red_scratch.f(),
                                e->red_scratch.bytes / sizeof(float)));
    STX_HIP(hipMemcpyAsync(e->A().dhost + di, terms, 3 * sizeof(double), hipMemcpyDeviceToHost,
                           e->stream));
    PendingLoss pl;
    pl.out = loss_out;
    pl.dterms.push_back(LossTerm{di + 0, tv_scale});
    pl.dterms.push_back(LossTerm{di + 1, p_scale});
    pl.dterms.push_back(LossTerm{di + 2, aux ? aux_scale * 0.5 : 0.0});
    e->A().pending.push_back(std::move(pl));
    return STX_OK;
}

int stx_image_swt_haar(stx_engine *e, const float *img, float *grad, int H, int W,
                       const int roll_xy[2], double scale, double power, double *loss_out) {
    if (!e || !img || !grad || H <= 0 || W <= 0 || power <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    size_t di;
    STX_TRY(alloc_dscalars(e, 1, &di));
    double *term = static_cast<double *>(e->A().dscalars.ptr) + di;
    STX_TRY(swt_haar_launch(e->stream, img, grad, H, W, roll_xy ? roll_xy[0] : 0,
                            roll_xy ? roll_xy[1] : 0, (float)scale, (float)power, term,
                            e->red_scratch.f(), e->red_scratch.bytes / sizeof(float)));
    STX_HIP(hipMemcpyAsync(e->A().dhost + di, term, sizeof(double), hipMemcpyDeviceToHost,
                           e->stream));
    PendingLoss pl;
    pl.out = loss_out;
    pl.dterms.push_back(LossTerm{di, scale});
    e->A().pending.push_back(std::move(pl));
    return STX_OK;
}

int stx_adam_step(stx_engine *e, float *params, const float *grad, float *g1, float *g2, float *p1,
                  float *avg_out, size_t n, double lr, double b1, double b2, double bp1, double corr1,
                  double corr2, double corrp) {
    if (!e || !params || !grad || !g1 || !g2 || !p1 || !avg_out) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return adam_launch(e->stream, params, grad, g1, g2, p1, avg_out, n, lr, b1, b2, bp1, corr1,
                       corr2, corrp);
}

static int sync_scalar(stx_engine *e, size_t di, int n, double *out) {
    STX_HIP(hipMemcpyAsync(e->A().dhost + di, static_cast<double *>(e->A().dscalars.ptr) + di,
                           n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    STX_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n; ++i) out[i] = e->A().dhost[di + i];
    return STX_OK;
}

int stx_vec_dot(stx_engine *e, const float *x, const float *y, size_t n, double *out) {
    if (!e || !x || !y || !out) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const size_t di = e->dscalars_cap - 2;   // reserved slot for synchronous scalar results
    STX_TRY(dot_launch(e->stream, x, y, n, static_cast<double *>(e->A().dscalars.ptr) + di,
                       e->red_scratch.f(), e->red_scratch.bytes / sizeof(float)));
    return sync_scalar(e, di, 1, out);
}

int stx_vec_mean_abs(stx_engine *e, const float *x, size_t n, double *out) {
    if (!e || !x || !out || !n) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const size_t di = e->dscalars_cap - 2;
    STX_TRY(abs_sum_launch(e->stream, x, n, static_cast<double *>(e->A().dscalars.ptr) + di,
                           e->red_scratch.f(), e->red_scratch.bytes / sizeof(float)));
    STX_TRY(sync_scalar(e, di, 1, out));
    *out /= (double)n;
    return STX_OK;
}

int stx_vec_dot_async(stx_engine *e, const float *x, const float *y, size_t n, double *out_dev) {
    if (!e || !x || !y || !out_dev) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return dot_launch(e->stream, x, y, n, out_dev, e->red_scratch.f(),
                      e->red_scratch.bytes / sizeof(float));
}

int stx_vec_abs_sum_async(stx_engine *e, const float *x, size_t n, double *out_dev) {
    if (!e || !x || !out_dev) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return abs_sum_launch(e->stream, x, n, out_dev, e->red_scratch.f(),
                          e->red_scratch.bytes / sizeof(float));
}

int stx_vec_axpy_dev(stx_engine *e, double c1, const double *a_dev, double da, double c2,
                     const double *b_dev, double db, const float *x, float *y, size_t n) {
    if (!e || !a_dev || !x || !y || da == 0.0 || (b_dev && db == 0.0)) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return axpy_dev_launch(e->stream, c1, a_dev, da, c2, b_dev, db, x, y, n);
}

int stx_vec_scale_dev(stx_engine *e, double c, const double *den_dev, double den_div, float *x,
                      size_t n) {
    if (!e || !den_dev || !x || den_div == 0.0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return scale_dev_launch(e->stream, c, den_dev, den_div, x, n);
}

int stx_vec_axpy_dot_dev(stx_engine *e, double c1, const double *a_dev, double da, double c2,
                         const double *b_dev, double db, double scale_c, const double *scale_den_dev,
                         double scale_div, const float *x, const float *src, float *y, const float *z,
                         size_t n, double *out_dev) {
    if (!e || !a_dev || !x || !src || !y || !z || !out_dev || da == 0.0 || (b_dev && db == 0.0) ||
        (scale_den_dev && scale_div == 0.0))
        return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return axpy_dot_dev_launch(e->stream, c1, a_dev, da, c2, b_dev, db, scale_c, scale_den_dev, scale_div, x,
                               src, y, z, n, out_dev, e->red_scratch.f(), e->red_scratch.bytes / sizeof(float));
}

int stx_vec_lbfgs_pair(stx_engine *e, const float *g_new, float *g_old, const float *s, float *y, size_t n,
                       double *out_dev2, double *sy_host_sync) {
    if (!e || !g_new || !g_old || !s || !y || !out_dev2 || !sy_host_sync) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    STX_TRY(lbfgs_pair_launch(e->stream, g_new, g_old, s, y, n, out_dev2, e->red_scratch.f(),
                              e->red_scratch.bytes / sizeof(float)));
    const size_t di = e->dscalars_cap - 2;   // the pinned mirror's slot for synchronous scalar results
    STX_HIP(hipMemcpyAsync(e->A().dhost + di, out_dev2, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    STX_HIP(hipStreamSynchronize(e->stream));
    *sy_host_sync = e->A().dhost[di];
    return STX_OK;
}

int stx_vec_scale2_axpy(stx_engine *e, double c1, double c2, float *s, float *params, size_t n) {
    if (!e || !s || !params) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return scale2_axpy_launch(e->stream, (float)c1, (float)c2, s, params, n);
}

int stx_vec_axpy(stx_engine *e, double a, const float *x, float *y, size_t n) {
    if (!e || !x || !y) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return axpy_launch(e->stream, (float)a, x, y, n);
}

int stx_vec_scale(stx_engine *e, double a, float *x, size_t n) {
    if (!e || !x) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return scale_launch(e->stream, (float)a, x, n);
}

int stx_image_step_stats(stx_engine *e, const float *avg, float *old, int H, int W, double stats[2]) {
    if (!e || !avg || !old || !stats || H <= 0 || W <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const size_t di = e->dscalars_cap - 2;
    STX_TRY(step_stats_launch(e->stream, avg, old, H, W, static_cast<double *>(e->A().dscalars.ptr) + di,
                              e->red_scratch.f(), e->red_scratch.bytes / sizeof(float)));
    double raw[2];
    STX_TRY(sync_scalar(e, di, 2, raw));
    const double n = 3.0 * H * W;
    stats[0] = raw[0] / n;
    stats[1] = std::sqrt(raw[1] / n);
    return STX_OK;
}

int stx_image_step_stats_async(stx_engine *e, const float *avg, float *old, int H, int W,
                               double raw_sums[2]) {
    if (!e || !avg || !old || !raw_sums || H <= 0 || W <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    size_t di;
    STX_TRY(alloc_dscalars(e, 2, &di));
    double *dev = static_cast<double *>(e->A().dscalars.ptr) + di;
    STX_TRY(step_stats_launch(e->stream, avg, old, H, W, dev, e->red_scratch.f(),
                              e->red_scratch.bytes / sizeof(float)));
    STX_HIP(hipMemcpyAsync(e->A().dhost + di, dev, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    for (int i = 0; i < 2; ++i) {
        PendingLoss pl;
        pl.out = raw_sums + i;
        pl.dterms.push_back(LossTerm{di + (size_t)i, 1.0});
        e->A().pending.push_back(std::move(pl));
    }
    return STX_OK;
}

int stx_image_to_u8(stx_engine *e, const float *img, int H, int W, const float mean_bgr[3],
                    uint8_t *out_rgb_u8) {
    if (!e || !img || !mean_bgr || !out_rgb_u8 || H <= 0 || W <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return to_u8_launch(e->stream, img, H, W, mean_bgr, out_rgb_u8);
}

// --------------------------------------------------------------------------- test hooks
static int scratch_pack(stx_engine *e, const float *w, int Mo, int Ko, int ks, int dir,
                        const ConvConfig &cfg, const float **packed) {
    const int M = dir ? Ko : Mo, K = dir ? Mo : Ko;
    if (cfg.id >= 100) {
        STX_TRY(e->upload.ensure(wino_packed_floats(cfg, K, M) * sizeof(float)));
        STX_TRY(wino_pack_weights(e->stream, w, Mo, Ko, dir, cfg, e->upload.f()));
    } else {
        STX_TRY(e->upload.ensure(conv_packed_floats(cfg, K, M, ks) * sizeof(float)));
        STX_TRY(conv_pack_weights(e->stream, w, Mo, Ko, ks, dir, cfg, e->upload.f()));
    }
    *packed = e->upload.f();
    return STX_OK;
}

// Same shape-only selection as the tile path (Winograd where it applies), without the tuner.
static ConvConfig hook_config(stx_engine *e, int ksize, int K, int M, int H, int W) {
    ConvConfig cfg;
    if (wino_choice(e, ksize, K, M, H, W, &cfg)) return cfg;
    return conv_pick_config(ksize, K, M, H, W);
}

// The fp16-split kernel for a stand-alone operator call, by the tile path's rule; the input's maximum
// comes from a pass over it, the output's goes to a scratch group of the table.
static int hook_h2(stx_engine *e, ConvProblem &p, ConvConfig *cfg) {
    ConvConfig h;
    if (!h2_choice(p, &h)) return STX_OK;
    *cfg = h;
    STX_TRY(e->amax.ensure((2 * e->blobs.size() + 2) * kAmaxSlots * sizeof(unsigned)));
    unsigned *scratch = e->amax_slots((int)e->blobs.size(), true);
    STX_TRY(absmax_launch(e->stream, p.x, (size_t)p.K * p.H * p.W, scratch));
    STX_HIP(hipMemsetAsync(scratch + kAmaxSlots, 0, kAmaxSlots * sizeof(unsigned), e->stream));
    p.x_amax = scratch;
    p.y_amax = scratch + kAmaxSlots;
    return STX_OK;
}

int stx_op_conv_forward(stx_engine *e, const float *x, int Cin, int H, int W, const float *w,
                        const float *b, int Cout, int ksize, int relu, float *y) {
    if (!e || !x || !w || !y) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    if (conv_first_usable(Cin, Cout, ksize))      // the tile path's first-layer kernel
        return conv_first_launch(e->stream, x, w, b, y, Cin, H, W, relu, nullptr);
    ConvConfig cfg = hook_config(e, ksize, Cin, Cout, H, W);
    ConvProblem p{};
    p.x = x;
    p.y = y;
    p.bias = b;
    p.K = Cin;
    p.M = Cout;
    p.H = H;
    p.W = W;
    p.ksize = ksize;
    p.relu = relu;
    p.epilogue = kEpiForward;
    STX_TRY(hook_h2(e, p, &cfg));
    const float *packed = nullptr;
    STX_TRY(scratch_pack(e, w, Cout, Cin, ksize, 0, cfg, &packed));
    p.w = packed;
    STX_TRY(attach_splitk(e, cfg, p));
    return launch_conv(e, cfg, p);
}

int stx_op_conv_backward_data(stx_engine *e, const float *dy, int Cout, int H, int W, const float *w,
                              int Cin, int ksize, const float *relu_mask_data, float *dx) {
    if (!e || !dy || !w || !dx) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    if (ksize == 3 && Cin <= 4) {
        STX_TRY(e->upload.ensure(conv_small_packed_floats(Cout) * sizeof(float)));
        STX_TRY(conv_small_pack(e->stream, w, Cout, Cin, 1, e->upload.f()));
        return conv_small_launch(e->stream, dy, e->upload.f(), dx, relu_mask_data, Cout, Cin, H, W);
    }
    ConvConfig cfg = hook_config(e, ksize, Cout, Cin, H, W);
    ConvProblem p{};
    p.x = dy;
    p.y = dx;
    p.mask = relu_mask_data;
    p.K = Cout;
    p.M = Cin;
    p.H = H;
    p.W = W;
    p.ksize = ksize;
    p.epilogue = kEpiDgrad;
    STX_TRY(hook_h2(e, p, &cfg));
    const float *packed = nullptr;
    STX_TRY(scratch_pack(e, w, Cout, Cin, ksize, 1, cfg, &packed));
    p.w = packed;
    STX_TRY(attach_splitk(e, cfg, p));
    return launch_conv(e, cfg, p);
}

int stx_op_pool_forward(stx_engine *e, const float *x, int C, int H, int W, int mode, float *y) {
    if (!e || !x || !y) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    return pool_forward_launch(e->stream, x, C, H, W, mode, y);
}

int stx_op_pool_backward(stx_engine *e, const float *dy, const float *x, int C, int H, int W,
                         int mode, const float *relu_mask_data, float *dx) {
    if (!e || !dy || !x || !dx) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    // the mask source is the pool input itself (post-ReLU data of the blob below)
    return pool_backward_launch(e->stream, dy, x, C, H, W, mode, relu_mask_data != nullptr, dx);
}

int stx_op_style_terms(stx_engine *e, const float *feat, int C, int h, int w,
                       const float *gram_target, float *s_out, float *normalized_out,
                       double *half_sumsq, double *abs_sum) {
    if (!e || !feat || !gram_target || C <= 0 || C % 4 || h <= 0 || w <= 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    const int HW = h * w;
    const size_t count = (size_t)C * HW;
    // the launches of the style branch of stx_sc_grad_tile, in the same order
    STX_TRY(e->upload.ensure(count * sizeof(float)));
    float *sgrad = s_out ? s_out : e->upload.f();
    STX_TRY(do_sync(e));
    size_t si;
    STX_TRY(alloc_scalars(e, 2, &si));
    float *sc = e->A().scalars.f() + si;
    STX_TRY(launch_style_terms(e, e->stream, feat, C, h, w, gram_target, sgrad, sc, "op"));
    if (normalized_out)
        STX_TRY(inject_style_launch(e->stream, normalized_out, sgrad, count, sc + 1, 1.0f, false));
    STX_HIP(hipMemcpyAsync(e->A().host, e->A().scalars.ptr, e->A().used * sizeof(float),
                           hipMemcpyDeviceToHost, e->stream));
    STX_HIP(hipStreamSynchronize(e->stream));
    if (half_sumsq) *half_sumsq = 0.5 * (double)e->A().host[si];
    if (abs_sum) *abs_sum = (double)e->A().host[si + 1];
    e->A().used = 0;
    return STX_OK;
}

int stx_op_content_terms(stx_engine *e, const float *feat, int C, int h, int w,
                         const float *content, int content_h, int content_w, int oy, int ox,
                         const int roll_xy[2], float *normalized_out, double sums[2]) {
    if (!e || !feat || !content || C <= 0 || h <= 0 || w <= 0) return STX_ERR_ARG;
    if (oy < 0 || ox < 0 || oy + h > content_h || ox + w > content_w) {
        set_error("stx_op_content_terms: window exceeds the content map");
        return STX_ERR_ARG;
    }
    STX_TRY(e->set_device());
    ContentWindow win;
    win.C = C;
    win.fh = h;
    win.fw = w;
    win.ch = content_h;
    win.cw = content_w;
    win.oy = oy;
    win.ox = ox;
    win.sx = roll_xy ? roll_xy[0] : 0;
    win.sy = roll_xy ? roll_xy[1] : 0;
    STX_TRY(do_sync(e));
    size_t si;
    STX_TRY(alloc_scalars(e, 2 + 2 * 1024, &si));
    float *s = e->A().scalars.f() + si;
    STX_TRY(content_sums_launch(e->stream, feat, content, win, s));
    if (normalized_out)
        STX_TRY(inject_content_launch(e->stream, normalized_out, feat, content, win, s, 1.0f, false));
    STX_HIP(hipMemcpyAsync(e->A().host, e->A().scalars.ptr, (si + 2) * sizeof(float),
                           hipMemcpyDeviceToHost, e->stream));
    STX_HIP(hipStreamSynchronize(e->stream));
    if (sums) {
        sums[0] = (double)e->A().host[si];
        sums[1] = (double)e->A().host[si + 1];
    }
    e->A().used = 0;
    return STX_OK;
}

int stx_profile_enable(stx_engine *e, int on) {
    if (!e) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    STX_HIP(hipStreamSynchronize(e->stream));
    for (auto &pe : e->prof) {
        e->event_pool.push_back(pe.start);
        e->event_pool.push_back(pe.stop);
    }
    e->prof.clear();
    e->profiling = on != 0;
    return STX_OK;
}

int stx_profile_read(stx_engine *e, char *buf, size_t buf_len, size_t *needed) {
    if (!e || (!buf && buf_len)) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    STX_HIP(hipStreamSynchronize(e->stream));
    std::string out;
    std::vector<long long> marks((size_t)e->marks_used * 2);
    if (!marks.empty())
        STX_HIP(hipMemcpy(marks.data(), e->marks_buf.ptr, marks.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (auto &pe : e->prof) {
        float ms = 0.f;
        STX_HIP(hipEventElapsedTime(&ms, pe.start, pe.stop));
        double mhz = 0.0;      // the shader clock inside the group's convolution kernel (stx_clock_marks)
        if (pe.mark >= 0 && (size_t)pe.mark * 2 + 1 < marks.size() && marks[2 * pe.mark + 1] > 0)
            mhz = (double)marks[2 * pe.mark] / (double)marks[2 * pe.mark + 1] * 100.0;
        char line[256];
        snprintf(line, sizeof line, "%s\t%.6f\t%.6e\t%.1f\n", pe.label.c_str(), ms, pe.flops, mhz);
        out += line;
        e->event_pool.push_back(pe.start);
        e->event_pool.push_back(pe.stop);
    }
    e->prof.clear();
    if (needed) *needed = out.size() + 1;
    if (buf_len) {
        const size_t n = std::min(buf_len - 1, out.size());
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return STX_OK;
}

int stx_last_tile_ms(stx_engine *e, float *ms) {
    if (!e || !ms) return STX_ERR_ARG;
    if (!e->timed) {
        set_error("stx_last_tile_ms: no tile has been evaluated");
        return STX_ERR_STATE;
    }
    STX_TRY(e->set_device());
    // the newest call that has finished; if none of the last few has, wait for the newest
    // (only slots that were recorded: hipEventQuery calls a never-recorded event complete, and
    // hipEventElapsedTime then fails on it)
    int pick = e->ev_cur;
    for (int k = 0; k < e->ev_recorded; ++k) {
        const int i = (e->ev_cur - k + stx_engine::kTimed) % stx_engine::kTimed;
        const hipError_t q = hipEventQuery(e->ev_stop[i]);
        if (q == hipSuccess) {
            pick = i;
            break;
        }
        (void)hipGetLastError();      // hipErrorNotReady
    }
    STX_HIP(hipEventSynchronize(e->ev_stop[pick]));
    STX_HIP(hipEventElapsedTime(ms, e->ev_start[pick], e->ev_stop[pick]));
    return STX_OK;
}

int stx_clock_marks(stx_engine *e, int on) {
    if (!e) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    if (on) {
        STX_TRY(e->marks_buf.ensure((size_t)kMaxClockMarks * 2 * sizeof(long long)));
        STX_HIP(hipMemsetAsync(e->marks_buf.ptr, 0, (size_t)kMaxClockMarks * 2 * sizeof(long long), e->stream));
    }
    e->clock_marks = on != 0;
    return STX_OK;
}

int stx_clock_marks_read(stx_engine *e, double *mhz, int max_values, int *n_values) {
    if (!e || !mhz || !n_values || max_values < 0) return STX_ERR_ARG;
    STX_TRY(e->set_device());
    STX_HIP(hipStreamSynchronize(e->stream));
    const int n = std::min(max_values, e->marks_used);
    std::vector<long long> h((size_t)n * 2);
    if (n) STX_HIP(hipMemcpy(h.data(), e->marks_buf.ptr, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) mhz[i] = h[2 * i + 1] > 0 ? (double)h[2 * i] / (double)h[2 * i + 1] * 100.0 : 0.0;
    *n_values = n;
    e->marks_used = 0;
    return STX_OK;
}

int stx_last_tile_flops(stx_engine *e, double *algorithmic, double *issued) {
    if (!e || !algorithmic || !issued) return STX_ERR_ARG;
    if (!e->timed) {
        set_error("stx_last_tile_flops: no tile has been evaluated");
        return STX_ERR_STATE;
    }
    *algorithmic = e->flop_algorithmic;
    *issued = e->flop_issued;
    return STX_OK;
}

}  // extern "C"
