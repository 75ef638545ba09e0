// 3x3 convolution through 1-D Winograd F(2,3) along x on the fp32 matrix cores.
//
// Same job and the same interface as conv_mfma_kernel (forward + bias + ReLU, backward-to-data +
// ReLU mask + loss-gradient terms, split-K partials), one third fewer MFMAs.  For a pair of
// neighbouring outputs (x = 2t, 2t+1) of one row and the three taps g0,g1,g2 of one kernel row:
//     d0..d3 = in[2t-1 .. 2t+2]
//     V = [d0-d2, d1+d2, d2-d1, d1-d3]          U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
//     m_i = U_i * V_i  (summed over input channels and the three kernel rows)
//     out[2t] = m0+m1+m2        out[2t+1] = m1-m2-m3
// i.e. 4 multiplies instead of 6 per kernel row.  It is exact algebra (Lavin & Gray, F(2,3)); the
// rounding differs from the direct FMA chain at the 1e-7 level because sums of inputs / weights
// are formed before the products.
//
// GEMM view: D_i[m][t] = sum_k A_i[m][k] * B_i[k][t], i = 0..3 transform components, m = output
// channel, t = x-tile (two pixels), k = (input channel, kernel row).  One v_mfma_f32_32x32x2_f32
// covers 32 channels x 32 tiles = 64 pixels of one image row; a wave keeps the four component
// accumulators of its blocks and combines them in registers in the epilogue, so each lane stores
// two adjacent pixels.  The input patch is transformed on its way from global memory to LDS
// (six loads -> eight V values per thread and unit), the filter bank is transformed when packed.

#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


// Wave grid WM x WN, each wave owns TM x TN blocks of 32 channels x 64 pixels (32 x-tiles).
// Pixel patch: PR rows x 64*SEGS columns; PR * SEGS == TN * WN.
// Block-uniform values that come out of an integer division are computed on the vector ALU (the
// scalar unit has no divider) and stay in VGPRs; a buffer load whose scalar offset derives from
// them is then wrapped in a waterfall loop per load.  Pinning them to SGPRs keeps the whole
// address arithmetic of a chunk on the scalar unit.
static __device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int KC, int TM, int TN, int WM, int WN, int PR, int SEGS, int EPI>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_wino_kernel(WinoArgs a) {
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = 64 * WM * WN;
    constexpr int PC = 64 * SEGS;          // output columns of the patch
    constexpr int PT = 32 * SEGS;          // x-tiles per patch row
    constexpr int XR = PR + 2;             // input rows of the patch
    constexpr int W_FLOATS = KC * 12 * BM;
    constexpr int V_FLOATS = KC * XR * 4 * PT;
    constexpr int W_VEC4 = W_FLOATS / 4;
    constexpr int NW = (W_VEC4 + NT - 1) / NT;
    constexpr int UNITS = KC * XR * (PT / 2);          // a unit = two x-tiles = 4 output columns
    constexpr int NU = (UNITS + NT - 1) / NT;
    static_assert(PR * SEGS == TN * WN, "pixel blocks must match the wave grid");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wl = lds;
    float *Vl = lds + W_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware work order, see conv_mfma.hip
    const int m_tiles = a.m_tiles;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nb >> 3, r8 = nb & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
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

    constexpr unsigned kOob = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, a.w_bytes, 0x00020000);

    // unit u of the patch: channel ci, patch row r, tile pair tp -> raw inputs at columns
    // x0 - 1 + 4*tp + {0..5} of image row y0 - 1 + r
    unsigned xvoff[NU][6];
    int vdst[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const int u = tid + n * NT;
        const int ci = u / (XR * (PT / 2));
        const int rem = u - ci * (XR * (PT / 2));
        const int r = rem / (PT / 2), tp = rem - r * (PT / 2);
        const int yy = y0 - 1 + r;
        const bool row_ok = (NU * NT == UNITS || u < UNITS) && (unsigned)yy < (unsigned)a.H;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int xx = x0 - 1 + 4 * tp + i;
            xvoff[n][i] = row_ok && (unsigned)xx < (unsigned)a.W
                              ? (unsigned)(ci * HW + yy * a.W + xx) * 4u : kOob;
        }
        vdst[n] = (ci * XR + r) * (4 * PT) + 2 * tp;      // + i * PT for component i
    }
    unsigned wvoff[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int f = tid + n * NT;
        wvoff[n] = (NW * NT == W_VEC4 || f < W_VEC4) ? (unsigned)f * 16u : kOob;
    }
    const unsigned w_base = (unsigned)(mtile * a.w_tile_stride) * 4u;
    constexpr unsigned w_chunk = (unsigned)W_FLOATS * 4u;
    const unsigned x_chunk = (unsigned)(KC * HW) * 4u;

    u32x4 wreg[NW];
    float xreg[NU][6];
    auto load_stage = [&](int chunk) {
        const unsigned ws = (unsigned)sgpr((int)(w_base + (unsigned)chunk * w_chunk));
        const unsigned xs = (unsigned)sgpr((int)((unsigned)chunk * x_chunk));
#pragma unroll
        for (int n = 0; n < NW; ++n) wreg[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[n], ws, 0);
#pragma unroll
        for (int n = 0; n < NU; ++n)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                xreg[n][i] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rx, xvoff[n][i], xs, 0));
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int f = tid + n * NT;
            if (NW * NT == W_VEC4 || f < W_VEC4) reinterpret_cast<u32x4 *>(Wl)[f] = wreg[n];
        }
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int u = tid + n * NT;
            if (NU * NT == UNITS || u < UNITS) {
                // The asm makes the loaded values opaque until this point: otherwise the compiler
                // forms the differences right behind the loads, inside the MFMA loop, which needs
                // `s_waitcnt vmcnt` there and stalls the matrix pipe on global-memory latency.
                float d[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    d[i] = xreg[n][i];
                    asm volatile("" : "+v"(d[i]));
                }
                float *dst = Vl + vdst[n];
                // two x-tiles: inputs d0..d3 and d2..d5; components are PT floats apart
                *reinterpret_cast<float2 *>(dst) = make_float2(d[0] - d[2], d[2] - d[4]);
                *reinterpret_cast<float2 *>(dst + PT) = make_float2(d[1] + d[2], d[3] + d[4]);
                *reinterpret_cast<float2 *>(dst + 2 * PT) = make_float2(d[2] - d[1], d[4] - d[3]);
                *reinterpret_cast<float2 *>(dst + 3 * PT) = make_float2(d[1] - d[3], d[3] - d[5]);
            }
        }
    };

    f32x16 acc[TM][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.f;

    const float *wl = Wl + half * (12 * BM) + wm * (TM * 32) + l31;
    int voff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pb = wn * TN + j;
        voff[j] = half * (XR * 4 * PT) + (pb / SEGS) * (4 * PT) + (pb % SEGS) * 32 + l31;
    }
    // k-step s: channel pair q, kernel row ky, component c
    constexpr int NS = (KC / 2) * 12;
    auto load_a = [&](int s, float (&av)[TM]) {
        const int q = s / 12, t = s % 12;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = wl[((2 * q) * 12 + t) * BM + i * 32];
    };
    auto load_b = [&](int s, float (&bv)[TN]) {
        const int q = s / 12, t = s % 12, ky = t / 4, c = t % 4;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bv[j] = Vl[(2 * q) * (XR * 4 * PT) + ky * (4 * PT) + c * PT + voff[j]];
    };

    load_stage(c_begin);
    store_stage();
    __syncthreads();
#ifndef STX_ABLATE
#define STX_ABLATE 0   // timing experiments only (see conv_mfma.hip); results are wrong when non-zero
#endif
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const bool more = chunk + 1 < c_end;
        if (more && STX_ABLATE == 0) load_stage(chunk + 1);
        float av[2][TM], bv[2][TN];
        load_a(0, av[0]);
        load_b(0, bv[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
                load_a(s + 1, av[(s + 1) & 1]);
                load_b(s + 1, bv[(s + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int c = s % 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i], bv[s & 1][j],
                                                                        acc[i][j][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STX_ABLATE < 2) __syncthreads();
        if (more && STX_ABLATE == 0) {
            store_stage();
            __syncthreads();
        }
    }

    // ---- epilogue: combine the four components, then the same epilogues as conv_mfma_kernel.
    float s_scale = 0.f, c_scale = 0.f;
    if (EPI == kEpiDgradInject) {
        const float n = (float)((size_t)a.M * HW);
        if (a.inj.sgrad) s_scale = a.inj.s_coef * (1.0f / (a.inj.s_abs_sum[0] / n + kEps));
        if (a.inj.content) c_scale = a.inj.c_coef * (1.0f / (a.inj.c_sums[1] / n + kEps));
    }
    // float2 epilogue when rows are even and every array involved is 8-byte aligned
    const bool vec2 = ((a.W & 1) | (((size_t)a.y | (size_t)a.mask | (size_t)a.inj.feat |
                                     (size_t)a.inj.sgrad) & 7)) == 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pb = wn * TN + j;
        const int yy = y0 + pb / SEGS;
        const int xx0 = x0 + (pb % SEGS) * 64 + 2 * l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float c0 = acc[i][j][0][r], c1 = acc[i][j][1][r], c2 = acc[i][j][2][r],
                            c3 = acc[i][j][3][r];
                const float out[2] = {c0 + c1 + c2, c1 - c2 - c3};
                if (vec2) {
                    // even row length: the two outputs of a lane are one aligned float2
                    if (yy < a.H && xx0 < a.W && m < a.M) {
                        const long idx = (long)m * HW + yy * a.W + xx0;
                        float2 v = make_float2(out[0], out[1]);
                        if (EPI == kEpiPartial) {
                            *reinterpret_cast<float2 *>(a.y + (long)kslice * a.M * HW + idx) = v;
                            continue;
                        }
                        if (EPI == kEpiForward) {
                            if (a.bias) v.x += a.bias[m], v.y += a.bias[m];
                            if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
                        } else {
                            if (a.mask) {
                                const float2 k = *reinterpret_cast<const float2 *>(a.mask + idx);
                                v.x = k.x > 0.f ? v.x : 0.f;
                                v.y = k.y > 0.f ? v.y : 0.f;
                            }
                            if (EPI == kEpiDgradInject) {
                                if (a.inj.content) {
                                    const float2 f = *reinterpret_cast<const float2 *>(a.inj.feat + idx);
                                    v.x += c_scale * (f.x - a.inj.content[content_index(a.inj.win, m, yy, xx0)]);
                                    v.y += c_scale * (f.y - a.inj.content[content_index(a.inj.win, m, yy, xx0 + 1)]);
                                }
                                if (a.inj.sgrad) {
                                    const float2 g = *reinterpret_cast<const float2 *>(a.inj.sgrad + idx);
                                    v.x += s_scale * g.x;
                                    v.y += s_scale * g.y;
                                }
                            }
                        }
                        *reinterpret_cast<float2 *>(a.y + idx) = v;
                    }
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int xx = xx0 + e;
                    if (yy < a.H && xx < a.W && m < a.M) {
                        const long idx = (long)m * HW + yy * a.W + xx;
                        float v = out[e];
                        if (EPI == kEpiPartial) {
                            a.y[(long)kslice * a.M * HW + idx] = v;
                            continue;
                        }
                        if (EPI == kEpiForward) {
                            if (a.bias) v += a.bias[m];
                            if (a.relu) v = fmaxf(v, 0.f);
                        } else {
                            if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                            if (EPI == kEpiDgradInject) {
                                if (a.inj.content)
                                    v += c_scale * (a.inj.feat[idx] -
                                                    a.inj.content[content_index(a.inj.win, m, yy, xx)]);
                                if (a.inj.sgrad) v += s_scale * a.inj.sgrad[idx];
                            }
                        }
                        a.y[idx] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct WinoVariant {
    int kc, tm, tn, wm, wn, pr, segs;
};
static const WinoVariant kWino[] = {
    /*0*/ {8, 2, 1, 1, 4, 4, 1},   // 64 channels x (4 rows x 64 px)
    /*1*/ {8, 1, 1, 1, 4, 4, 1},   // 32 channels x (4 rows x 64 px)
    /*2*/ {4, 2, 1, 1, 4, 4, 1},   // 64 channels x (4 rows x 64 px), KC 4
};

ConvConfig wino_config_by_id(int id) {
    if (id >= 200) return id == 202 ? h2_config(1, 2) : h2_config(id - 200 + 1);
    if (id >= 110) return wino4_config(id - 110);
    if (id >= 100) return wino2_config(id - 100);
    const WinoVariant &v = kWino[id];
    ConvConfig c;
    c.id = 100 + id;                  // ids >= 100 mark Winograd configurations
    c.bm = 32 * v.tm * v.wm;
    c.kc = v.kc;
    c.pr = v.pr;
    c.pc = 64 * v.segs;
    c.threads = 64 * v.wm * v.wn;
    c.lds_bytes = sizeof(float) * ((size_t)v.kc * 12 * c.bm +
                                   (size_t)v.kc * (v.pr + 2) * 4 * 32 * v.segs);
    return c;
}

size_t wino_packed_floats(const ConvConfig &cfg, int K, int M) {
    if (cfg.id >= 300) return h2_packed_floats(K, M);
    if (cfg.id >= 200) return wino2_packed_floats(K, M);
    const size_t kpad = (size_t)ceil_div(K, cfg.kc) * cfg.kc;
    return (size_t)ceil_div(M, cfg.bm) * kpad * 12 * cfg.bm;
}

// packed[mt][(k*3 + ky)*4 + i][mm] = U_i of filter row ky of W(m = mt*BM + mm, k)
__global__ void wino_pack_kernel(const float *__restrict__ w, int Mo, int Ko, int transpose_flip,
                                 int M, int K, int bm, int kpad, float *__restrict__ packed,
                                 size_t total) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int mm = idx % bm;
        const size_t row = idx / bm;
        const int i = row % 4;
        const int ky = (row / 4) % 3;
        const int k = (row / 12) % kpad;
        const int mt = row / ((size_t)12 * kpad);
        const int m = mt * bm + mm;
        float v = 0.f;
        if (m < M && k < K) {
            float g[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx;
                g[kx] = transpose_flip ? w[((size_t)k * Ko + m) * 9 + (8 - t)]
                                       : w[((size_t)m * Ko + k) * 9 + t];
            }
            v = i == 0 ? g[0]
              : i == 1 ? (g[0] + g[1] + g[2]) * 0.5f
              : i == 2 ? (g[0] - g[1] + g[2]) * 0.5f
                       : g[2];
        }
        packed[idx] = v;
    }
}

int wino_pack_weights(hipStream_t s, const float *w_caffe, int Mo, int Ko, int transpose_flip,
                      const ConvConfig &cfg, float *packed) {
    if (cfg.id >= 300) return h2_pack_weights(s, w_caffe, Mo, Ko, transpose_flip, packed);
    if (cfg.id >= 200) return wino2_pack_weights(s, w_caffe, Mo, Ko, transpose_flip, packed);
    const int M = transpose_flip ? Ko : Mo;
    const int K = transpose_flip ? Mo : Ko;
    const int kpad = ceil_div(K, cfg.kc) * cfg.kc;
    const size_t total = wino_packed_floats(cfg, K, M);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    wino_pack_kernel<<<blocks, 256, 0, s>>>(w_caffe, Mo, Ko, transpose_flip, M, K, cfg.bm, kpad,
                                            packed, total);
    STX_CHECK_LAUNCH();
    return STX_OK;
}

#define STX_WINO_VARIANT(ID, KC, TM, TN, WM, WN, PR, SEGS)                                        \
    template <int EPI>                                                                            \
    static int wino_launch_##ID(hipStream_t s, const ConvConfig &cfg, const WinoArgs &args,       \
                                int n_wg) {                                                       \
        auto kern = conv_wino_kernel<KC, TM, TN, WM, WN, PR, SEGS, EPI>;                          \
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

STX_WINO_VARIANT(0, 8, 2, 1, 1, 4, 4, 1)
STX_WINO_VARIANT(1, 8, 1, 1, 1, 4, 4, 1)
STX_WINO_VARIANT(2, 4, 2, 1, 1, 4, 4, 1)

// Launches a Winograd configuration (cfg.id >= 100); `w` must come from wino_pack_weights.
int wino_launch(hipStream_t s, const ConvConfig &cfg, const ConvProblem &p, int ksplit) {
    if (cfg.id >= 300) return h2_launch(s, cfg, p, ksplit);
    if (cfg.id >= 200) return wino2_launch(s, cfg, p, ksplit);
    WinoArgs a;
    a.x = p.x;
    a.w = p.w;
    a.y = p.y;
    a.bias = p.bias;
    a.mask = p.mask;
    a.K = p.K;
    a.M = p.M;
    a.H = p.H;
    a.W = p.W;
    a.n_chunks = ceil_div(p.K, cfg.kc);
    a.tiles_x = ceil_div(p.W, cfg.pc);
    a.tiles_y = ceil_div(p.H, cfg.pr);
    a.m_tiles = ceil_div(p.M, cfg.bm);
    a.ksplit = 1;
    a.w_tile_stride = a.n_chunks * cfg.kc * 12 * cfg.bm;
    a.relu = p.relu;
    a.inj = p.inject;
    a.pool_out = nullptr;
    a.pool_mode = 0;
    const double xb = 4.0 * p.K * (double)p.H * p.W;
    const double wb = 4.0 * (double)wino_packed_floats(cfg, p.K, p.M);
    if (xb >= 2147483648.0 || wb >= 2147483648.0) {
        set_error("wino_launch: plane set exceeds the 2 GiB buffer-addressing limit");
        return STX_ERR_UNSUPPORTED;
    }
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
    const bool inject = p.epilogue == kEpiDgrad && (p.inject.sgrad || p.inject.content);
    int n_wg = a.m_tiles * a.tiles_x * a.tiles_y;
    const bool split = ksplit > 1 && p.splitk_ws &&
                       p.splitk_ws_floats >= (size_t)ksplit * p.M * p.H * p.W;
    if (split) {
        a.ksplit = ksplit;
        a.y = p.splitk_ws;
        n_wg *= ksplit;
    }
#define STX_WINO_DISPATCH(ID)                                                                     \
    case 100 + ID:                                                                                \
        if (split) {                                                                              \
            STX_TRY(wino_launch_##ID<kEpiPartial>(s, cfg, a, n_wg));                              \
            return splitk_reduce_launch(s, p, ksplit);                                            \
        }                                                                                         \
        if (p.epilogue == kEpiForward) return wino_launch_##ID<kEpiForward>(s, cfg, a, n_wg);     \
        if (inject) return wino_launch_##ID<kEpiDgradInject>(s, cfg, a, n_wg);                    \
        if (p.epilogue == kEpiDgrad) return wino_launch_##ID<kEpiDgrad>(s, cfg, a, n_wg);         \
        break;
    switch (cfg.id) {
        STX_WINO_DISPATCH(0)
        STX_WINO_DISPATCH(1)
        STX_WINO_DISPATCH(2)
    }
#undef STX_WINO_DISPATCH
    set_error("wino_launch: no kernel for config %d epilogue %d", cfg.id, p.epilogue);
    return STX_ERR_UNSUPPORTED;
}

}  // namespace stx
