// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 fp32 matrix cores.
//
// One kernel serves four jobs of the per-tile path (style_transfer.py:556-612 delegates them to
// Caffe's Convolution layer and to scipy's SSYMM):
//   * conv forward + bias + ReLU                       (net.forward,  style_transfer.py:425,566)
//   * conv backward-to-data + ReLU mask of the blob below   (net.backward, style_transfer.py:608-610)
//     -- the same kernel with the filter bank transposed and rotated by 180 degrees
//   * S = sym(tril(G - Gs)) . F, the style gradient          (ssymm, num_utils.py:60-66)
//     -- a 1x1 "convolution" whose weights are the symmetric matrix itself, plus sum|S| partials
//
// GEMM view: D[m][p] = sum_k A[m][k] * B[k][p], m = output channel, p = pixel of a PR x PC patch,
// k = (input channel, tap).  v_mfma_f32_32x32x2_f32 computes a 32(m) x 32(p) block per wave-
// instruction with K = 2: lanes 0-31 feed k, lanes 32-63 feed k+1 (one VGPR per operand).  With
// p = 32 consecutive x of one image row, the B operand of tap (ky,kx) is a 32-float run of the
// LDS-staged input patch shifted by (ky,kx) -- conflict-free ds_read_b32 -- and every D register
// stores a 128-byte row segment of the NCHW output.  fp32 MFMA issues once per 64 cycles per
// SIMD, so LDS (two or six ds_read_b32 per eight MFMAs) and the global->LDS stage are far from
// their limits; the structure is: register-prefetch the next K-chunk from global while the
// current chunk is multiplied out of LDS, two workgroups per CU to cover the stage swap.
//
// Numerics: exact fp32 FMA chains in k order (the MFMA f32 path does not round differently from
// v_fma_f32), accumulation in fp32 like Caffe's SGEMM.

#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvKernelArgs {
    const float *x;
    const float *w;
    float *y;
    const float *bias;
    const float *mask;
    float *partials;
    int K, M, H, W;
    int n_chunks;          // ceil(K / KC)
    int tiles_x, tiles_y;  // pixel tiles
    int m_tiles;           // output-channel tiles
    int ksplit;            // K slices (kEpiPartial); 1 otherwise
    int w_row_stride;      // floats between consecutive k rows of the weight source
    int w_tile_stride;     // floats between consecutive output-channel tiles (packed mode)
    int x_bytes, w_bytes;  // sizes of the x and w buffers (hardware bounds check of the loads)
    int relu;
    ConvInject inj;        // kEpiDgradInject
};

// KS: kernel size (3 -> pad 1, 1 -> pad 0).  KC: reduction channels per stage (even).
// Wave grid WM x WN, each wave owns TM x TN blocks of 32 channels x 32 pixels.
// Pixel patch PR rows x PC cols (PC multiple of 32); PR * PC / 32 == TN * WN.
// PACKED: weights come as pre-tiled [m_tile][k_row][BM] slabs (always in bounds);
// otherwise as rows of a dense [K][M] matrix (the symmetric style matrix) with bounds masks.
// DB: two LDS stages.  The next chunk is written into the idle stage in the middle of the current
// chunk's MFMAs (its global loads were issued a chunk earlier), so a chunk costs one barrier
// instead of two and no wave ever sits in a write phase with the matrix pipe idle.
// Block-uniform values that come out of an integer division are computed on the vector ALU (the
// scalar unit has no divider) and stay in VGPRs; a buffer load whose scalar offset derives from
// them is then wrapped in a waterfall loop per load.  Pinning them to SGPRs keeps the whole
// address arithmetic of a chunk on the scalar unit.
static __device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int KS, int KC, int TM, int TN, int WM, int WN, int PR, int PC, int EPI, bool PACKED,
          bool DB = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_kernel(ConvKernelArgs a) {
    constexpr int KK = KS * KS;
    constexpr int BM = 32 * TM * WM;
    constexpr int XR = PR + KS - 1;
    constexpr int XC = PC + KS - 1;
    constexpr int PAD = KS / 2;
    constexpr int NT = 64 * WM * WN;
    constexpr int SEGS = PC / 32;
    static_assert(PR * SEGS == TN * WN, "pixel blocks must match the wave grid");
    static_assert(KC % 2 == 0, "MFMA consumes k in pairs");
    constexpr int W_FLOATS = KC * KK * BM;
    constexpr int X_FLOATS = KC * XR * XC;
    constexpr int W_VEC4 = W_FLOATS / 4;
    constexpr int NW = (W_VEC4 + NT - 1) / NT;       // float4 weight loads per thread per stage
    constexpr int NX = (X_FLOATS + NT - 1) / NT;     // input loads per thread per stage

    constexpr int STAGE = W_FLOATS + X_FLOATS;     // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wl = lds;
    float *Xl = lds + W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order.  The dispatcher places workgroup b on XCD b % 8, each with a private
    // L2.  Logical work item L = (pixel tile, channel tile) with the channel tile fastest, and
    // every XCD gets a contiguous range of L: all channel tiles of one pixel patch run back to
    // back on ONE XCD, so the input patch is fetched from HBM once instead of once per channel
    // tile (placement only affects speed, never results).
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
    const int kslice = sgpr(EPI == kEpiPartial ? L % a.ksplit : 0);
    const int Lt = sgpr(EPI == kEpiPartial ? L / a.ksplit : L);
    const int ptile = sgpr(Lt / m_tiles);
    const int mtile = Lt - ptile * m_tiles;
    const int c_begin = sgpr(EPI == kEpiPartial ? kslice * a.n_chunks / a.ksplit : 0);
    const int c_end = sgpr(EPI == kEpiPartial ? (kslice + 1) * a.n_chunks / a.ksplit : a.n_chunks);
    const int y0 = sgpr((ptile / a.tiles_x) * PR);
    const int x0 = sgpr((ptile % a.tiles_x) * PC);
    const int m0 = mtile * BM;
    const int HW = a.H * a.W;

    // ---- global -> register staging through buffer loads.  Out-of-range elements (zero padding
    // at the tile border, channel / row padding) get an offset beyond the descriptor's range and
    // come back as 0 from the hardware bounds check, so no select touches the loaded data and
    // the loads stay in flight across the whole MFMA loop of the current chunk.  All per-lane
    // offsets are computed once; a chunk only changes the scalar offset.
    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);
    unsigned xvoff[NX], wvoff[NW];
#pragma unroll
    for (int n = 0; n < NX; ++n) {
        const int e = tid + n * NT;
        const int ci = e / (XR * XC);
        const int rem = e - ci * (XR * XC);
        const int r = rem / XC, c = rem - r * XC;
        const int yy = y0 - PAD + r, xx = x0 - PAD + c;
        const bool ok = (NX * NT == X_FLOATS || e < X_FLOATS) && (unsigned)yy < (unsigned)a.H &&
                        (unsigned)xx < (unsigned)a.W;
        xvoff[n] = ok ? (unsigned)(ci * HW + yy * a.W + xx) * 4u : kOob;
    }
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int f = tid + n * NT;
        const int row = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
        bool ok = NW * NT == W_VEC4 || f < W_VEC4;
        if (!PACKED) ok = ok && m0 + c4 < a.M;   // rows beyond K fall off the end of the matrix
        wvoff[n] = ok ? (unsigned)(row * a.w_row_stride + c4) * 4u : kOob;
    }
    const unsigned w_base = PACKED ? (unsigned)(mtile * a.w_tile_stride) * 4u : (unsigned)m0 * 4u;
    const unsigned w_chunk = (unsigned)(KC * KK * a.w_row_stride) * 4u;
    const unsigned x_chunk = (unsigned)(KC * HW) * 4u;

    u32x4 wreg[NW];
    unsigned xreg[NX];
    auto load_stage = [&](int chunk) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * w_chunk));
        const unsigned xs = (unsigned)sgpr((int)((unsigned)chunk * x_chunk));
#pragma unroll
        for (int n = 0; n < NW; ++n) wreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[n], ws, 0);
#pragma unroll
        for (int n = 0; n < NX; ++n) xreg[n] = __builtin_amdgcn_raw_buffer_load_b32(rx, xvoff[n], xs, 0);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int f = tid + n * NT;
            if (NW * NT == W_VEC4 || f < W_VEC4)
                reinterpret_cast<u32x4 *>(Wl + buf * STAGE)[f] = wreg[n];
        }
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const int e = tid + n * NT;
            if (NX * NT == X_FLOATS || e < X_FLOATS)
                reinterpret_cast<unsigned *>(Xl + buf * STAGE)[e] = xreg[n];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-wave operand bases (lanes 32..63 read the odd k of each pair)
    const float *wl = Wl + half * (KK * BM) + wm * (TM * 32) + l31;
    int xoff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pb = wn * TN + j;
        xoff[j] = half * (XR * XC) + (pb / SEGS) * XC + (pb % SEGS) * 32 + l31;
    }
    // operands of k-step s (a pair of input channels q, tap t)
    constexpr int NS = (KC / 2) * KK;
    static_assert(NS % 2 == 0, "k-steps are software-pipelined in pairs");
    const float *wlc = wl, *xlc = Xl;              // operand bases of the stage being multiplied
    auto load_a = [&](int s, float (&av)[TM]) {
        const int q = s / KK, t = s % KK;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = wlc[((2 * q) * KK + t) * BM + i * 32];
    };
    auto load_b = [&](int s, float (&bv)[TN]) {
        const int q = s / KK, t = s % KK;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bv[j] = xlc[(2 * q) * (XR * XC) + (t / KS) * XC + (t % KS) + xoff[j]];
    };
    auto multiply = [&](const float (&av)[TM], const float (&bv)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    };

    load_stage(c_begin);
    store_stage(0);
    __syncthreads();

    constexpr int S_STORE = ((NS * 5 / 8) / 2) * 2;   // k-step before which the idle stage is filled
    int cur = 0;
#ifndef STX_ABLATE
#define STX_ABLATE 0   // timing experiments only: 1 no staging in the loop, 2 also no barriers,
#endif                 // 3 also no LDS operand reads.  Results are wrong when non-zero.
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const bool more = chunk + 1 < c_end;
        if (more && STX_ABLATE == 0) load_stage(chunk + 1);
        if (DB) {
            wlc = wl + cur * STAGE;
            xlc = Xl + cur * STAGE;
        }
        // A ring of R operand register sets: the LDS reads of step s+R-1 are issued before the
        // MFMAs of step s, so their latency hides behind (R-1) x TM*TN x 64 cycles of matrix
        // work.  sched_barrier pins that order: without it the scheduler sinks each read to
        // just before its first use and the wave stalls on LDS latency every k-step.
        constexpr int R = 2;   // deeper rings measured no faster (small tiles) or slower (VGPRs)
        float av[R][TM], bv[R][TN];
#pragma unroll
        for (int d = 0; d < R - 1; ++d) {
            load_a(d, av[d]);
            load_b(d, bv[d]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (DB && s == S_STORE && more) {
                store_stage(cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s + R - 1 < NS && (STX_ABLATE < 3 || chunk == c_begin)) {
                load_a(s + R - 1, av[(s + R - 1) % R]);
                load_b(s + R - 1, bv[(s + R - 1) % R]);
            }
            __builtin_amdgcn_sched_barrier(0);
            multiply(av[s % R], bv[s % R]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STX_ABLATE < 2) __syncthreads();
        if (DB) {
            cur ^= 1;
        } else if (more && STX_ABLATE == 0) {
            store_stage(0);
            __syncthreads();
        }
    }

    // ---- epilogue: D register r of a block holds row (r&3) + 8*(r>>2) + 4*half, column l31.
    float abs_sum = 0.f;
    float s_scale = 0.f, c_scale = 0.f;
    if (EPI == kEpiDgradInject) {
        // normalize(): x * 1 / (sum|x| / size + EPS) (num_utils.py:85-87), times lw * weight
        const float n = (float)((size_t)a.M * HW);
        if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / n + kEps));
        if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / n + kEps));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pb = wn * TN + j;
        const int yy = y0 + pb / SEGS;
        const int xx = x0 + (pb % SEGS) * 32 + l31;
        const bool in_img = yy < a.H && xx < a.W;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (in_img && m < a.M) {
                    const long idx = (long)m * HW + yy * a.W + xx;
                    float v = acc[i][j][r];
                    if (EPI == kEpiPartial) {
                        a.y[(long)kslice * a.M * HW + idx] = v;
                        continue;
                    }
                    if (EPI == kEpiForward) {
                        if (a.bias) v += a.bias[m];
                        if (a.relu) v = fmaxf(v, 0.f);
                    } else if (EPI == kEpiDgrad) {
                        if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                    } else if (EPI == kEpiDgradInject) {
                        if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                        if (a.inj.content)
                            v += c_scale * (a.inj.feat[idx] -
                                            a.inj.content[content_index(a.inj.win, m, yy, xx)]);
                        if (a.inj.sgrad) v += s_scale * a.inj.sgrad[idx];
                    } else {
                        abs_sum += fabsf(v);
                    }
                    a.y[idx] = v;
                }
            }
        }
    }
    if (EPI == kEpiSymm) {
        // workgroup reduction of sum|S| -> one partial per workgroup (summed later in fixed order)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) abs_sum += __shfl_down(abs_sum, off, 64);
        __syncthreads();
        if (lane == 0) lds[wave] = abs_sum;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int i = 0; i < WM * WN; ++i) s += lds[i];
            a.partials[blockIdx.x] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Instantiation table
// ------------------------------------------------------------------------------------------------
struct ConvVariant {
    int ks, kc, tm, tn, wm, wn, pr, pc, db;
};

static const ConvVariant kVariants[] = {
    /*0*/ {3, 8, 2, 4, 2, 2, 8, 32},   // BM 128 x 256 px : wide middle layers
    /*1*/ {3, 8, 2, 4, 1, 4, 8, 64},   // BM  64 x 512 px : 64-channel layers at full resolution
    /*2*/ {3, 8, 2, 2, 1, 4, 8, 32},   // BM  64 x 256 px : deep layers with few pixels
    /*3*/ {3, 8, 1, 2, 1, 4, 8, 32},   // BM  32 x 256 px : backward into the 3-channel image
    /*4*/ {3, 4, 2, 4, 1, 4, 8, 64},   // BM  64 x 512 px, KC 4 : first layer (3 input channels)
    /*5*/ {3, 8, 2, 1, 1, 4, 4, 32},   // BM  64 x 128 px : smallest planes
    /*6*/ {1, 16, 2, 4, 2, 2, 8, 32},  // 1x1, BM 128 : style-gradient product, C >= 128
    /*7*/ {1, 16, 2, 4, 1, 4, 8, 64},  // 1x1, BM  64 : style-gradient product, C == 64
    /*8*/ {3, 4, 2, 1, 1, 4, 4, 32},   // BM  64 x 128 px, KC 4 : first layer, small tiles
    /*9*/ {3, 8, 2, 1, 1, 4, 4, 32, 1},   // BM 64 x 128 px, two LDS stages
    /*10*/ {3, 4, 2, 1, 1, 4, 4, 32, 1},  // BM 64 x 128 px, KC 4, two LDS stages
    /*11*/ {3, 8, 2, 2, 1, 4, 8, 32, 1},  // BM 64 x 256 px, two LDS stages
    /*12*/ {1, 32, 2, 1, 1, 4, 4, 32},    // 1x1, BM 64 x 128 px, KC 32 : many small workgroups
    /*13*/ {1, 64, 2, 1, 1, 4, 4, 32},    // 1x1, BM 64 x 128 px, KC 64
    /*14*/ {1, 32, 2, 2, 1, 4, 8, 32},    // 1x1, BM 64 x 256 px, KC 32
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

static ConvConfig make_config(int id) {
    const ConvVariant &v = kVariants[id];
    ConvConfig c;
    c.id = id;
    c.bm = 32 * v.tm * v.wm;
    c.kc = v.kc;
    c.pr = v.pr;
    c.pc = v.pc;
    c.threads = 64 * v.wm * v.wn;
    const int kk = v.ks * v.ks;
    c.lds_bytes = sizeof(float) * (1 + v.db) * ((size_t)v.kc * kk * c.bm +
                                   (size_t)v.kc * (v.pr + v.ks - 1) * (v.pc + v.ks - 1));
    return c;
}

ConvConfig conv_config_by_id(int id) { return make_config(id); }

int conv_num_workgroups(const ConvConfig &cfg, int M, int H, int W) {
    return ceil_div(M, cfg.bm) * ceil_div(H, cfg.pr) * ceil_div(W, cfg.pc);
}

ConvConfig conv_pick_config(int ksize, int K, int M, int H, int W) {
    if (ksize == 1) {
        const char *force = sw_env("STX_CONV_SYMM");                       // tuning aid: 6, 7, 12-14
        if (force && *force) return make_config(atoi(force));
        // small planes: many small workgroups (measured: 64x64 px x 512 ch 0.086 -> 0.033 ms,
        // 128x128 0.087 -> 0.076; larger planes are faster with the big tiles)
        if ((long)H * W <= 128 * 128) return make_config(12);
        return make_config(M >= 128 ? 6 : 7);
    }
    if (K <= 4) {
        const char *first = sw_env("STX_CONV_FIRST");
        return make_config(first ? atoi(first) : 8);
    }
    if (M <= 32) return make_config(3);
    if (const char *force = sw_env("STX_CONV_FORCE")) {   // tuning aid: force one tile config
        const int id = atoi(force);
        if (id >= 0 && id <= 11 && id != 3 && id != 4 && id != 6 && id != 7 && id != 8)
            return make_config(id);
    }
    // Static default (the engine autotunes per shape on top of this): measured on MI355X, many
    // small workgroups beat few large ones because co-resident workgroups run out of phase and
    // cover each other's stage swaps and epilogues -- 64 channels x 128 pixels (5 workgroups per
    // CU) unless the plane is so large that 256-pixel tiles still give >= 8 workgroups per CU.
    ConvConfig c2 = make_config(2);
    if (conv_num_workgroups(c2, M, H, W) >= 2048 && K >= 128) return c2;
    return make_config(5);
}

size_t conv_packed_floats(const ConvConfig &cfg, int K, int M, int ksize) {
    const size_t kpad = (size_t)ceil_div(K, cfg.kc) * cfg.kc;
    return (size_t)ceil_div(M, cfg.bm) * kpad * ksize * ksize * cfg.bm;
}

// packed[mt][(k*KK + t)][mm] = W(m = mt*BM + mm, k, t), zero outside the filter bank.
__global__ void pack_weights_kernel(const float *__restrict__ w, int Mo, int Ko, int ks,
                                    int transpose_flip, int M, int K, int bm, int kpad,
                                    float *__restrict__ packed, size_t total) {
    const int kk = ks * ks;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int mm = i % bm;
        const size_t row = i / bm;
        const int t = row % kk;
        const int k = (row / kk) % kpad;
        const int mt = row / ((size_t)kk * kpad);
        const int m = mt * bm + mm;
        float v = 0.f;
        if (m < M && k < K) {
            if (!transpose_flip)
                v = w[((size_t)m * Ko + k) * kk + t];
            else  // backward-data: out channel m is the filter's input channel, taps rotated 180
                v = w[((size_t)k * Ko + m) * kk + (kk - 1 - t)];
        }
        packed[i] = v;
    }
}

int conv_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int ksize,
                      int transpose_flip, const ConvConfig &cfg, float *packed) {
    const int M = transpose_flip ? Ko : Mo;
    const int K = transpose_flip ? Mo : Ko;
    const int kpad = ceil_div(K, cfg.kc) * cfg.kc;
    const size_t total = conv_packed_floats(cfg, K, M, ksize);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    pack_weights_kernel<<<blocks, 256, 0, s>>>(w_caffe, Mo, Ko, ksize, transpose_flip, M, K, cfg.bm,
                                               kpad, packed, total);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

#define STX_CONV_VARIANT(ID, KS, KC, TM, TN, WM, WN, PR, PC) \
    STX_CONV_VARIANT_DB(ID, KS, KC, TM, TN, WM, WN, PR, PC, false)
#define STX_CONV_VARIANT_DB(ID, KS, KC, TM, TN, WM, WN, PR, PC, DBUF)                             \
    template <int EPI, bool PACKED>                                                               \
    static int launch_##ID(hipStream_t s, const ConvConfig &cfg, const ConvKernelArgs &args,      \
                           int n_wg) {                                                            \
        auto kern = conv_mfma_kernel<KS, KC, TM, TN, WM, WN, PR, PC, EPI, PACKED, DBUF>;          \
        if (cfg.lds_bytes > 64 * 1024) {                                                          \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                               (int)cfg.lds_bytes);                               \
            if (e != hipSuccess) {                                                                \
                set_error("hipFuncSetAttribute(lds=%zu): %s", cfg.lds_bytes,                      \
                          hipGetErrorString(e));                                                  \
                return STX_ERR_HIP;                                                               \
            }                                                                                     \
        }                                                                                         \
        kern<<<n_wg, cfg.threads, cfg.lds_bytes, s>>>(args);                                      \
        STX_CHECK_LAUNCH();                                                                       \
        return STX_OK;                                                                            \
    }

STX_CONV_VARIANT(0, 3, 8, 2, 4, 2, 2, 8, 32)
STX_CONV_VARIANT(1, 3, 8, 2, 4, 1, 4, 8, 64)
STX_CONV_VARIANT(2, 3, 8, 2, 2, 1, 4, 8, 32)
STX_CONV_VARIANT(3, 3, 8, 1, 2, 1, 4, 8, 32)
STX_CONV_VARIANT(4, 3, 4, 2, 4, 1, 4, 8, 64)
STX_CONV_VARIANT(5, 3, 8, 2, 1, 1, 4, 4, 32)
STX_CONV_VARIANT(6, 1, 16, 2, 4, 2, 2, 8, 32)
STX_CONV_VARIANT(7, 1, 16, 2, 4, 1, 4, 8, 64)
STX_CONV_VARIANT(8, 3, 4, 2, 1, 1, 4, 4, 32)
STX_CONV_VARIANT(12, 1, 32, 2, 1, 1, 4, 4, 32)
STX_CONV_VARIANT(13, 1, 64, 2, 1, 1, 4, 4, 32)
STX_CONV_VARIANT(14, 1, 32, 2, 2, 1, 4, 8, 32)
STX_CONV_VARIANT_DB(9, 3, 8, 2, 1, 1, 4, 4, 32, true)
STX_CONV_VARIANT_DB(10, 3, 4, 2, 1, 1, 4, 4, 32, true)
STX_CONV_VARIANT_DB(11, 3, 8, 2, 2, 1, 4, 8, 32, true)

// Sums the K slices in a fixed order and applies the epilogue the unsplit kernel would have.
struct SplitReduceArgs {
    const float *part;
    float *y;
    const float *bias, *mask;
    ConvInject inj;
    int ksplit, M, HW, W, relu, epilogue;
    unsigned *y_amax;       // ConvProblem::y_amax (or null): max |y| of what the pass writes
};

__device__ __forceinline__ float splitk_reduce_element(const SplitReduceArgs &a, size_t n, size_t i, float s_scale,
                                                       float c_scale) {
    float v = a.part[i];
    for (int k = 1; k < a.ksplit; ++k) v += a.part[(size_t)k * n + i];
    // (plane sets stay under 2^32 elements: a 32-bit division instead of a 64-bit one per element)
    const int m = (int)((unsigned)i / (unsigned)a.HW);
    if (a.epilogue == kEpiForward) {
        if (a.bias) v += a.bias[m];
        if (a.relu) v = fmaxf(v, 0.f);
    } else {
        if (a.mask) v = a.mask[i] > 0.f ? v : 0.f;
        if (a.inj.content) {
            const unsigned pix = (unsigned)i - (unsigned)m * (unsigned)a.HW;
            v += c_scale * (a.inj.feat[i] -
                            a.inj.content[content_index(a.inj.win, m, (int)(pix / (unsigned)a.W), (int)(pix % (unsigned)a.W))]);
        }
        if (a.inj.sgrad) v += s_scale * a.inj.sgrad[i];
    }
    a.y[i] = v;
    return fabsf(v);
}

// max |y| of a pass into the slots the next fp16-split convolution reads (conv_h2.hip): one atomic per wave
__device__ __forceinline__ void splitk_reduce_amax(const SplitReduceArgs &a, float amax) {
    if (!a.y_amax) return;          // (uniform)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(a.y_amax + (blockIdx.x & (kAmaxSlots - 1)),
                  __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// Four consecutive elements per thread (plane sizes that are a multiple of 4: the four share their
// channel; 16-byte accesses, the slices of a vector requested four at a time).  Every element is the sum
// of its slices in slice order, then the epilogue of splitk_reduce_element: the same bits.
typedef float f32x4r __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void splitk_reduce_vec4_kernel(SplitReduceArgs a) {
    const unsigned n = (unsigned)a.M * (unsigned)a.HW, n4 = n >> 2;
    float s_scale = 0.f, c_scale = 0.f;
    if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / (float)n + kEps));
    if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / (float)n + kEps));
    float amax = 0.f;
    const f32x4r *part = reinterpret_cast<const f32x4r *>(a.part);
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < n4; q += gridDim.x * 256u) {
        f32x4r v = part[q];
        int k = 1;
        for (; k + 3 < a.ksplit; k += 4) {
            const f32x4r p0 = part[(size_t)k * n4 + q], p1 = part[(size_t)(k + 1) * n4 + q];
            const f32x4r p2 = part[(size_t)(k + 2) * n4 + q], p3 = part[(size_t)(k + 3) * n4 + q];
            v += p0;
            v += p1;
            v += p2;
            v += p3;
        }
        for (; k < a.ksplit; ++k) v += part[(size_t)k * n4 + q];
        const unsigned i = q << 2, m = i / (unsigned)a.HW;
        if (a.epilogue == kEpiForward) {
            if (a.bias) v += a.bias[m];
            if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        } else {
            if (a.mask) {
                const f32x4r mk = reinterpret_cast<const f32x4r *>(a.mask)[q];
                v.x = mk.x > 0.f ? v.x : 0.f, v.y = mk.y > 0.f ? v.y : 0.f;
                v.z = mk.z > 0.f ? v.z : 0.f, v.w = mk.w > 0.f ? v.w : 0.f;
            }
            if (a.inj.content) {
                const f32x4r ft = reinterpret_cast<const f32x4r *>(a.inj.feat)[q];
                const unsigned pix = i - m * (unsigned)a.HW;
                float cv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cv[e] = a.inj.content[content_index(a.inj.win, (int)m, (int)((pix + e) / (unsigned)a.W),
                                                        (int)((pix + e) % (unsigned)a.W))];
                v.x += c_scale * (ft.x - cv[0]), v.y += c_scale * (ft.y - cv[1]);
                v.z += c_scale * (ft.z - cv[2]), v.w += c_scale * (ft.w - cv[3]);
            }
            if (a.inj.sgrad) {
                const f32x4r sg = reinterpret_cast<const f32x4r *>(a.inj.sgrad)[q];
                v.x += s_scale * sg.x, v.y += s_scale * sg.y, v.z += s_scale * sg.z, v.w += s_scale * sg.w;
            }
        }
        reinterpret_cast<f32x4r *>(a.y)[q] = v;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    splitk_reduce_amax(a, amax);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitReduceArgs a) {
    const size_t n = (size_t)a.M * a.HW;
    float s_scale = 0.f, c_scale = 0.f;
    if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / (float)n + kEps));
    if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / (float)n + kEps));
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        amax = fmaxf(amax, splitk_reduce_element(a, n, i, s_scale, c_scale));
    splitk_reduce_amax(a, amax);
}

// The same over the 64-channel x (pr x pc)-pixel patches of a range of 2-D Winograd work items
// (conv_wino2's tail split): one thread per element (the slices of an element are read one after
// the other, so the pass lives on the number of threads in flight: with 16 elements per thread it
// took 60 us for 32 patches), threads along the patch's rows.
struct ItemRange {
    int item_base, m_tiles, tiles_x, tiles_y, pr, pc, H;
};
constexpr int kItemBlocks = 64;       // x 256 threads = 64 channels x 256 pixels

__global__ __launch_bounds__(256) void splitk_reduce_items_kernel(SplitReduceArgs a, ItemRange r) {
    const size_t n = (size_t)a.M * a.HW;
    float s_scale = 0.f, c_scale = 0.f;
    if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / (float)n + kEps));
    if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / (float)n + kEps));
    int pt, mt;
    wino2_item_tiles(r.item_base + (int)(blockIdx.x / kItemBlocks), r.m_tiles, r.tiles_x * r.tiles_y, pt, mt);
    const int y0 = (pt / r.tiles_x) * r.pr, x0 = (pt % r.tiles_x) * r.pc, m0 = mt * 64;
    const int patch = r.pr * r.pc, total = 64 * patch;
    float amax = 0.f;
    for (int e = (int)(blockIdx.x % kItemBlocks) * 256 + (int)threadIdx.x; e < total; e += kItemBlocks * 256) {
        const int mm = e / patch, rem = e - mm * patch;
        const int yy = y0 + rem / r.pc, xx = x0 + rem % r.pc, m = m0 + mm;
        if (m < a.M && yy < r.H && xx < a.W)
            amax = fmaxf(amax, splitk_reduce_element(a, n, (size_t)m * a.HW + (size_t)yy * a.W + xx, s_scale, c_scale));
    }
    splitk_reduce_amax(a, amax);
}

int conv_splitk_factor(const ConvConfig &cfg, const ConvProblem &p, bool packed) {
    if (!packed || p.ksize != 3 || (p.epilogue != kEpiForward && p.epilogue != kEpiDgrad)) return 1;
    if (cfg.id >= 300) return h2_splitk_factor(cfg, p);
    if (cfg.id >= 200) return wino2_splitk_factor(cfg, p);
    if (cfg.id == 3 || cfg.id == 4 || cfg.id == 8) return 1;   // (ids 100-102: 1-D Winograd, allowed)
    const int n_wg = conv_num_workgroups(cfg, p.M, p.H, p.W);
    const int n_chunks = ceil_div(p.K, cfg.kc);
    // All workgroups of a launch become resident at once while they fit (about five per CU for
    // the small-tile configs), so a count that is not a multiple of the 256 CUs leaves the CUs
    // unevenly loaded for the whole kernel: 552 workgroups run as 3 on some CUs and 2 on others,
    // 72 % efficient.  Slice K until the load is within 10 % of even or the launch oversubscribes
    // the chip (then the dispatcher balances dynamically).
    constexpr int kCUs = 256, kResident = 5 * kCUs;
    const int max_split = std::min(16, n_chunks / 4);      // at least four chunks per slice
    for (int f = 1; f <= max_split; ++f) {
        const int n = n_wg * f;
        if (n > kResident) return f;
        const double even = (double)n / kCUs, worst = (double)ceil_div(n, kCUs);
        if (n >= 2 * kCUs && even / worst >= 0.9) return f;
    }
    return std::max(max_split, 1);
}

size_t conv_splitk_floats(const ConvConfig &cfg, const ConvProblem &p, bool packed) {
    int f = conv_splitk_factor(cfg, p, packed);
    if (packed && p.ksize == 3 && cfg.id >= 200 && cfg.id < 210)     // (the tail split's slices are whole planes too)
        f = std::max(f, wino2_max_slices(cfg, p));
    return f > 1 ? (size_t)f * p.M * p.H * p.W : 0;
}

int conv_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, bool packed) {
    const ConvVariant &v = kVariants[cfg.id];
    if (v.ks != p.ksize) {
        set_error("conv_launch: config %d is %dx%d, problem is %dx%d", cfg.id, v.ks, v.ks, p.ksize,
                  p.ksize);
        return STX_ERR_ARG;
    }
    ConvKernelArgs a;
    a.x = p.x;
    a.w = p.w;
    a.y = p.y;
    a.bias = p.bias;
    a.mask = p.mask;
    a.partials = p.partials;
    a.K = p.K;
    a.M = p.M;
    a.H = p.H;
    a.W = p.W;
    a.n_chunks = ceil_div(p.K, cfg.kc);
    a.tiles_x = ceil_div(p.W, cfg.pc);
    a.tiles_y = ceil_div(p.H, cfg.pr);
    a.m_tiles = ceil_div(p.M, cfg.bm);
    a.w_row_stride = packed ? cfg.bm : p.M;
    a.w_tile_stride = packed ? a.n_chunks * cfg.kc * p.ksize * p.ksize * cfg.bm : 0;
    const double xb = 4.0 * p.K * (double)p.H * p.W;
    const double wb = packed ? 4.0 * (double)conv_packed_floats(cfg, p.K, p.M, p.ksize)
                             : 4.0 * (double)p.K * p.M;
    if (xb >= 2147483648.0 || wb >= 2147483648.0) {
        set_error("conv_launch: a %d x %d x %d plane set exceeds the 2 GiB buffer-addressing limit "
                  "of the kernel (use a smaller --tile-size)", p.K, p.H, p.W);
        return STX_ERR_UNSUPPORTED;
    }
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
    a.relu = p.relu;
    a.inj = p.inject;
    a.ksplit = 1;
    const bool inject = p.epilogue == kEpiDgrad && (p.inject.sgrad || p.inject.content);
    int n_wg = conv_num_workgroups(cfg, p.M, p.H, p.W);
    const int ksplit = conv_splitk_factor(cfg, p, packed);
    const bool split = ksplit > 1 && p.splitk_ws &&
                       p.splitk_ws_floats >= (size_t)ksplit * p.M * p.H * p.W;
    if (split) {
        a.ksplit = ksplit;
        a.y = p.splitk_ws;
        n_wg *= ksplit;
    }

#define STX_DISPATCH(ID)                                                                          \
    case ID:                                                                                      \
        if (split) {                                                                              \
            STX_TRY((launch_##ID<kEpiPartial, true>(s, cfg, a, n_wg)));                           \
            goto reduce;                                                                          \
        }                                                                                         \
        if (p.epilogue == kEpiForward && packed)                                                  \
            return launch_##ID<kEpiForward, true>(s, cfg, a, n_wg);                               \
        if (inject && packed) return launch_##ID<kEpiDgradInject, true>(s, cfg, a, n_wg);          \
        if (p.epilogue == kEpiDgrad && packed) return launch_##ID<kEpiDgrad, true>(s, cfg, a, n_wg); \
        break;
#define STX_DISPATCH_NOINJ(ID)                                                                    \
    case ID:                                                                                      \
        if (p.epilogue == kEpiForward && packed)                                                  \
            return launch_##ID<kEpiForward, true>(s, cfg, a, n_wg);                               \
        if (p.epilogue == kEpiDgrad && packed && !inject)                                         \
            return launch_##ID<kEpiDgrad, true>(s, cfg, a, n_wg);                                 \
        break;
#define STX_DISPATCH_SYMM(ID)                                                                     \
    case ID:                                                                                      \
        if (p.epilogue == kEpiSymm && !packed) return launch_##ID<kEpiSymm, false>(s, cfg, a, n_wg); \
        if (p.epilogue == kEpiForward && packed)                                                  \
            return launch_##ID<kEpiForward, true>(s, cfg, a, n_wg);                               \
        if (p.epilogue == kEpiDgrad && packed && !inject)                                         \
            return launch_##ID<kEpiDgrad, true>(s, cfg, a, n_wg);                                 \
        break;
    switch (cfg.id) {
        STX_DISPATCH(0)
        STX_DISPATCH(1)
        STX_DISPATCH(2)
        STX_DISPATCH_NOINJ(3)
        STX_DISPATCH_NOINJ(4)
        STX_DISPATCH_NOINJ(8)
        STX_DISPATCH(9)
        STX_DISPATCH(10)
        STX_DISPATCH(11)
        STX_DISPATCH(5)
        STX_DISPATCH_SYMM(6)
        STX_DISPATCH_SYMM(7)
        STX_DISPATCH_SYMM(12)
        STX_DISPATCH_SYMM(13)
        STX_DISPATCH_SYMM(14)
    }
#undef STX_DISPATCH
#undef STX_DISPATCH_NOINJ
#undef STX_DISPATCH_SYMM
    set_error("conv_launch: no kernel for config %d epilogue %d packed %d", cfg.id, p.epilogue,
              (int)packed);
    return STX_ERR_UNSUPPORTED;
reduce:
    return splitk_reduce_launch(s, p, ksplit);
}

static SplitReduceArgs splitk_reduce_args(const ConvProblem &p, int ksplit) {
    SplitReduceArgs r;
    r.part = p.splitk_ws;
    r.y = p.y;
    r.bias = p.bias;
    r.mask = p.mask;
    r.inj = p.inject;
    r.ksplit = ksplit;
    r.M = p.M;
    r.HW = p.H * p.W;
    r.W = p.W;
    r.relu = p.relu;
    r.epilogue = p.epilogue;
    r.y_amax = p.y_amax;
    return r;
}

int splitk_reduce_launch(hipStream_t s, const ConvProblem &p, int ksplit) {
    const SplitReduceArgs r = splitk_reduce_args(p, ksplit);
    const size_t n = (size_t)p.M * p.H * p.W;
    if (n >= ((size_t)1 << 32)) {      // (the kernels index one plane set with 32 bits; only small planes split)
        set_error("splitk_reduce: %zu elements", n);
        return STX_ERR_UNSUPPORTED;
    }
    // four elements per thread where a vector stays inside one channel and every array is 16-byte aligned
    // (STX_REDUCE_VEC=0: the element-wise kernel; bit-identical either way)
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(r.part) | reinterpret_cast<uintptr_t>(r.y) |
                           reinterpret_cast<uintptr_t>(r.mask) | reinterpret_cast<uintptr_t>(r.inj.feat) |
                           reinterpret_cast<uintptr_t>(r.inj.sgrad);
    const char *env = sw_env("STX_REDUCE_VEC");
    if ((p.H * p.W) % 4 == 0 && (ptrs & 15) == 0 && n < (1u << 31) && !(env && atoi(env) == 0)) {
        const size_t n4 = n / 4;
        splitk_reduce_vec4_kernel<<<(int)std::min<size_t>((n4 + 255) / 256, p.y_amax ? 1024 : 4096), 256, 0, s>>>(r);
        STX_CHECK_LAUNCH();
        return STX_OK;
    }
    // (with y_amax every block ends in an atomic on one of kAmaxSlots words: fewer, longer blocks)
    splitk_reduce_kernel<<<(int)std::min<size_t>((n + 255) / 256, p.y_amax ? 1024 : 4096), 256, 0, s>>>(r);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int splitk_reduce_items_launch(hipStream_t s, const ConvProblem &p, const ConvConfig &cfg, int slices,
                               int item_base, int items) {
    const SplitReduceArgs r = splitk_reduce_args(p, slices);
    if ((size_t)p.M * p.H * p.W >= ((size_t)1 << 32)) {
        set_error("splitk_reduce: %zu elements", (size_t)p.M * p.H * p.W);
        return STX_ERR_UNSUPPORTED;
    }
    ItemRange range;
    range.item_base = item_base;
    range.m_tiles = ceil_div(p.M, 64);
    range.tiles_x = ceil_div(p.W, cfg.pc);
    range.tiles_y = ceil_div(p.H, cfg.pr);
    range.pr = cfg.pr;
    range.pc = cfg.pc;
    range.H = p.H;
    splitk_reduce_items_kernel<<<items * kItemBlocks, 256, 0, s>>>(r, range);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx

// ================================================================================================
// 3x3 convolution with at most 4 output channels: the backward pass into the 3-channel image
// (dX = W^T (*) dY of conv1_1, style_transfer.py:608).  A 32x32 MFMA tile would waste 29 of its
// 32 rows; v_mfma_f32_4x4x1_16b_f32 computes sixteen independent 4x4 outer products per
// instruction, which maps to 4 channels x 64 consecutive pixels with K = 1: lane l feeds pixel l
// as the B operand and weight row (l & 3) as the A operand, and D register r of lane l is output
// channel r at pixel l.  Three of four rows are useful and the instruction issues at the full
// 64 FLOP/clk/SIMD rate, so the layer becomes bound by reading the 64-channel gradient once
// from HBM instead of by matrix work.
// ================================================================================================
namespace stx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallConvArgs {
    const float *x;      // [K][H][W]
    const float *w;      // packed [Kpad * 9][4]
    float *y;            // [M][H][W], M <= 4
    const float *mask;   // optional [M][H][W]
    int K, M, H, W, n_chunks, tiles_x, x_bytes, w_bytes;
    int n_wg, wg_per_xcd;   // patches of the plane; patches per XCD (work order, see the kernel)
};

// VEC (plane width a multiple of 4, 16-byte aligned planes): the 64 interior columns of a patch
// row are 16 aligned 16-byte loads and the two halo columns two dword loads -- 6 loads per thread
// and chunk instead of 21 dword loads (the kernel reads the whole 64-channel gradient once and
// was bound by the number of load instructions: 268 MB in 92 us).  LDS rows are laid out as
// [3 pad][left halo][64 interior][right halo][3 pad] so that the interior is 16-byte aligned.
template <int KC, int PR, bool VEC>
__global__ __launch_bounds__(256) void conv3x3_m4_kernel(SmallConvArgs a) {
    constexpr int PC = 64, XR = PR + 2, XC = VEC ? PC + 8 : PC + 2, X0 = VEC ? 3 : 0, NT = 256, RW = PR / 4;
    constexpr int X_FLOATS = KC * XR * XC, W_FLOATS = KC * 9 * 4;
    constexpr int NX = (KC * XR * (PC + 2) + NT - 1) / NT;          // dword loads per thread (!VEC)
    constexpr int NV = KC * XR * (PC / 4);                          // 16-byte loads per chunk (VEC)
    constexpr int NVT = (NV + NT - 1) / NT;
    constexpr int NH = KC * XR * 2;                                 // halo dwords per chunk (VEC)
    static_assert(W_FLOATS / 4 <= NT, "one float4 of weights per thread");
    constexpr int NHT = (NH + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float Xl[X_FLOATS];
    __shared__ __attribute__((aligned(16))) float Wl[W_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work order.  Workgroup b of a launch runs on XCD b mod 8 (tools/ubench/xcc_map.hip), and each
    // XCD has its own L2: with patches dealt out in launch order the two halo columns of a patch
    // row (one dword each, but a whole cache line of the neighbouring patch) and its two halo rows
    // were fetched by an XCD that never saw the neighbour -- 556 MB fetched for 268 MB of input by
    // PMC (profiles/r03: 2.07x).  Every XCD takes a contiguous band of patch rows instead, in
    // raster order, so that a halo line is in the L2 the neighbour just pulled it through.
    const int patch = sgpr((int)(blockIdx.x & 7) * a.wg_per_xcd + (int)(blockIdx.x >> 3));
    if (patch >= a.n_wg) return;
    const int y0 = sgpr((patch / a.tiles_x) * PR), x0 = sgpr((patch % a.tiles_x) * PC);
    const int HW = a.H * a.W;
    constexpr unsigned kOob = 0x80000000u;
    // x_bytes == 0: a plane set of 2 GiB or more -- the descriptor is moved to each chunk's KC
    // planes instead of reaching them through the 32-bit offset
    const bool big = a.x_bytes == 0;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);
    unsigned xvoff[VEC ? 1 : NX], vvoff[VEC ? NVT : 1], hvoff[NHT];
    int vdst[VEC ? NVT : 1], hdst[NHT];
    if (VEC) {
#pragma unroll
        for (int n = 0; n < NVT; ++n) {
            const int e = tid + n * NT;                       // (channel, row, 16-byte column group)
            const int ci = e / (XR * (PC / 4)), rem = e - ci * (XR * (PC / 4));
            const int r = rem / (PC / 4), c4 = rem - r * (PC / 4);
            const int yy = y0 - 1 + r, xx = x0 + 4 * c4;
            // (W is a multiple of 4 and x0 of 64: a group is inside the row or outside it entirely)
            const bool ok = e < NV && (unsigned)yy < (unsigned)a.H && xx < a.W;
            vvoff[n] = ok ? (unsigned)(ci * HW + yy * a.W + xx) * 4u : kOob;
            vdst[n] = e < NV ? (ci * XR + r) * XC + X0 + 1 + 4 * c4 : -1;
        }
#pragma unroll
        for (int n = 0; n < NHT; ++n) {
            const int e = tid + n * NT;
            const int ci = e / (XR * 2), rem = e - ci * (XR * 2);
            const int r = rem >> 1, side = rem & 1;
            const int yy = y0 - 1 + r, xx = side ? x0 + PC : x0 - 1;
            const bool ok = e < NH && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            hvoff[n] = ok ? (unsigned)(ci * HW + yy * a.W + xx) * 4u : kOob;
            hdst[n] = e < NH ? (ci * XR + r) * XC + X0 + (side ? PC + 1 : 0) : -1;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const int e = tid + n * NT;
            const int ci = e / (XR * XC), rem = e - ci * (XR * XC);
            const int r = rem / XC, c = rem - r * XC;
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            const bool ok = e < X_FLOATS && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            xvoff[n] = ok ? (unsigned)(ci * HW + yy * a.W + xx) * 4u : kOob;
        }
    }
    const unsigned wvoff = tid < W_FLOATS / 4 ? (unsigned)tid * 16u : kOob;
    unsigned xreg[VEC ? 1 : NX], hreg[NHT];
    u32x4 vreg[VEC ? NVT : 1];
    u32x4 wreg;
    auto load_stage = [&](int chunk) {
        wreg = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, (unsigned)chunk * (W_FLOATS * 4u), 0);
        unsigned xs = (unsigned)sgpr((int)((unsigned)chunk * (unsigned)(KC * HW) * 4u));
        if (big) {
            const int k0 = sgpr(chunk * KC);
            const int nk = a.K - k0 < KC ? a.K - k0 : KC;
            rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + (size_t)k0 * (size_t)HW, 0,
                                                   (nk > 0 ? nk : 0) * HW * 4, 0x00020000);
            xs = 0;
        }
        if (VEC) {
#pragma unroll
            for (int n = 0; n < NVT; ++n) vreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rx, vvoff[n], xs, 0);
#pragma unroll
            for (int n = 0; n < NHT; ++n) hreg[n] = __builtin_amdgcn_raw_buffer_load_b32(rx, hvoff[n], xs, 0);
        } else {
#pragma unroll
            for (int n = 0; n < NX; ++n) xreg[n] = __builtin_amdgcn_raw_buffer_load_b32(rx, xvoff[n], xs, 0);
        }
    };
    auto store_stage = [&]() {
        if (tid < W_FLOATS / 4) reinterpret_cast<u32x4 *>(Wl)[tid] = wreg;
        if (VEC) {
#pragma unroll
            for (int n = 0; n < NVT; ++n)
                if (vdst[n] >= 0) *reinterpret_cast<u32x4 *>(Xl + vdst[n]) = vreg[n];
#pragma unroll
            for (int n = 0; n < NHT; ++n)
                if (hdst[n] >= 0) reinterpret_cast<unsigned *>(Xl)[hdst[n]] = hreg[n];
        } else {
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const int e = tid + n * NT;
                if (e < X_FLOATS) reinterpret_cast<unsigned *>(Xl)[e] = xreg[n];
            }
        }
    };
    f32x4 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wl = Wl + (lane & 3);
    const float *xl = Xl + (wave * RW) * XC + X0 + lane;

    load_stage(0);
    store_stage();
    __syncthreads();
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        const bool more = chunk + 1 < a.n_chunks;
        if (more) load_stage(chunk + 1);
#pragma unroll
        for (int ci = 0; ci < KC; ++ci) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float av = wl[(ci * 9 + t) * 4];
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const float bv = xl[ci * (XR * XC) + (r + t / 3) * XC + t % 3];
                    acc[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[r], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_stage();
            __syncthreads();
        }
    }
    const int xx = x0 + lane;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int yy = y0 + wave * RW + r;
        if (yy < a.H && xx < a.W) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < a.M) {
                    const size_t idx = (size_t)m * HW + (size_t)yy * a.W + xx;
                    float v = acc[r][m];
                    if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                    a.y[idx] = v;
                }
            }
        }
    }
}

constexpr int kSmallKC = 4, kSmallPR = 16;    // (tools/bench_small.py: 79 us against 86 with 8-channel chunks, three more workgroups per CU)

size_t conv_small_packed_floats(int K) { return (size_t)ceil_div(K, kSmallKC) * kSmallKC * 9 * 4; }

// packed[(k*9 + t)][m] for the backward-data direction of a Caffe bank w[Mo][Ko][3][3]:
// output channel m = filter input channel, k = filter output channel, taps rotated by 180 degrees
// (transpose_flip = 1), or the forward direction (0).
__global__ void pack_small_kernel(const float *__restrict__ w, int Mo, int Ko, int transpose_flip,
                                  int M, int K, int kpad, float *__restrict__ packed) {
    const int total = kpad * 9 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i & 3, t = (i >> 2) % 9, k = i / 36;
        float v = 0.f;
        if (m < M && k < K)
            v = transpose_flip ? w[((size_t)k * Ko + m) * 9 + (8 - t)] : w[((size_t)m * Ko + k) * 9 + t];
        packed[i] = v;
    }
}

int conv_small_pack(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                    float *packed) {
    const int M = transpose_flip ? Ko : Mo, K = transpose_flip ? Mo : Ko;
    const int kpad = ceil_div(K, kSmallKC) * kSmallKC;
    pack_small_kernel<<<ceil_div(kpad * 36, 256), 256, 0, s>>>(w_caffe, Mo, Ko, transpose_flip, M, K,
                                                               kpad, packed);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

int conv_small_launch(hipStream_t s, const float *x, const float *packed, float *y,
                      const float *mask, int K, int M, int H, int W) {
    if (M > 4) {
        set_error("conv_small_launch: M = %d > 4", M);
        return STX_ERR_ARG;
    }
    const double xb = 4.0 * K * (double)H * W;
    const char *force_big = sw_env("STX_WINO_BIG");
    const bool big = xb >= 2147483648.0 || (force_big && atoi(force_big) == 1);
    if (4.0 * kSmallKC * (double)H * W >= 2147483648.0) {
        set_error("conv_small_launch: a %d x %d plane is beyond the buffer-addressing limit", H, W);
        return STX_ERR_UNSUPPORTED;
    }
    SmallConvArgs a;
    a.x = x;
    a.w = packed;
    a.y = y;
    a.mask = mask;
    a.K = K;
    a.M = M;
    a.H = H;
    a.W = W;
    a.n_chunks = ceil_div(K, kSmallKC);
    a.tiles_x = ceil_div(W, 64);
    a.x_bytes = big ? 0 : (int)xb;
    a.w_bytes = (int)(conv_small_packed_floats(K) * 4);
    // 16-byte loads need 16-byte aligned rows: plane width a multiple of 4, aligned base
    const char *novec = sw_env("STX_SMALL_NOVEC");
    const bool vec = W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !(novec && atoi(novec));
    // 16-row patches (10 % of halo rows, 0.75 LDS reads per MFMA) where the plane still yields a
    // few workgroups per CU; 8-row patches (25 %, 1.2) on small planes
    const bool tall = (long)a.tiles_x * ceil_div(H, kSmallPR) >= 1024;
    int pr = tall ? kSmallPR : kSmallPR / 2;
#ifdef STX_SMALL_SWEEP      // tuning aid (tools/bench_small.py): STX_SMALL_TUNE=<KC><PR code>, identical results
    if (const char *tune = sw_env("STX_SMALL_TUNE")) {
        const int kc = atoi(tune) / 100, prr = atoi(tune) % 100;
        a.n_chunks = ceil_div(K, kc);
        a.n_wg = a.tiles_x * ceil_div(H, prr);
        a.wg_per_xcd = ceil_div(a.n_wg, 8);
        const int g = a.wg_per_xcd * 8;
#define STX_SWEEP(KC_, PR_) if (kc == KC_ && prr == PR_) { conv3x3_m4_kernel<KC_, PR_, true><<<g, 256, 0, s>>>(a); STX_CHECK_LAUNCH(); return STX_OK; }
        STX_SWEEP(4, 8) STX_SWEEP(4, 16) STX_SWEEP(4, 32) STX_SWEEP(8, 8) STX_SWEEP(8, 16) STX_SWEEP(8, 32) STX_SWEEP(16, 8) STX_SWEEP(16, 16)
#undef STX_SWEEP
    }
#endif
    a.n_wg = a.tiles_x * ceil_div(H, pr);
    a.wg_per_xcd = ceil_div(a.n_wg, 8);
    const int grid = a.wg_per_xcd * 8;
    if (tall) {
        if (vec) conv3x3_m4_kernel<kSmallKC, kSmallPR, true><<<grid, 256, 0, s>>>(a);
        else conv3x3_m4_kernel<kSmallKC, kSmallPR, false><<<grid, 256, 0, s>>>(a);
    } else {
        if (vec) conv3x3_m4_kernel<kSmallKC, kSmallPR / 2, true><<<grid, 256, 0, s>>>(a);
        else conv3x3_m4_kernel<kSmallKC, kSmallPR / 2, false><<<grid, 256, 0, s>>>(a);
    }
    STX_CHECK_LAUNCH();
    return STX_OK;
}

}  // namespace stx
