// Microbenchmark ("price with coresident.hip re-run on the fp16 mix first"): is it worth
// running TWO four-wave workgroups per CU instead of ONE eight-wave workgroup for conv_h2's instruction mix,
// so that the prologue / epilogue of one runs under the fp16 matrix work of the other?
//
// A synthetic work item with the mix of conv_h2_kernel<*, 2, 1>, per wave and chunk:
//   72 v_mfma_f32_32x32x16_f16 (8 accumulators x 3 kernel rows x 3 products), 24 ds_read_b128 (B fragments),
//   12 buffer_load_dwordx4 (A fragments, L2-resident), one LDS-only barrier,
//   staging: U quad units (4 buffer_load_dwordx4, 4 x (4 v_add + 8 v_fma_mix + 2 ds_write_b64)) and U single
//   units (1 load, 2 x (2 v_add + 4 v_fma_mix + 4 ds_write_b16)), U = 1 with eight waves, 2 with four;
//   prologue: the first chunk's loads from a cold region, its staging, a barrier;
//   epilogue: 2 passes of (16 loads of 8 bytes, barrier, 16 ds_write_b128, barrier, 32 ds_read_b128, ~100
//   vector instructions, 16 stores of 8 bytes).
// mode 8: 512 threads, 128 KB of LDS, one workgroup per CU.  mode 4: 256 threads, 80 KB, two per CU, twice the
// workgroups (the same matrix work in total; every workgroup stages its whole patch, as the real one would).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coresident_h2.hip -o tools/ubench/bin/coresident_h2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ long long g_cycles[4];     // prologue, loop, epilogue of one workgroup of the last round

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void item(float *out, const float *in, unsigned in_bytes, int chunks, float sv, int stag_shift, int stag_ticks) {
    constexpr int NT = WAVES * 64;
    constexpr int U = WAVES == 8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)in_bytes, 0x00020000);
    const unsigned base = (unsigned)(((size_t)blockIdx.x * 2654435761u) % (64u << 20)) & ~1023u;   // the patch: cold
    const unsigned wbase = (256u << 20) + (unsigned)(wave & 3) * 65536u;                            // the bank: hot
    constexpr unsigned kV = 40960;                       // one V buffer
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 xr[U][4], xe[U];
    f16x8 af[2][2][2], bq[2][2];
    // a staggered start: two identical workgroups dispatched together onto one CU stay in phase for ever --
    // half of the first 512 wait half an item (wall clock, 100 MHz) before their first load
    if (stag_ticks > 0 && blockIdx.x < 512 && ((blockIdx.x >> stag_shift) & 1)) {
        const long long w0 = wall_clock64();
        while (wall_clock64() - w0 < stag_ticks) __builtin_amdgcn_s_sleep(8);
    }
    const long long t0 = clock64();
    auto x_load = [&](int c) __attribute__((always_inline)) {
        const unsigned so = base + (unsigned)c * 16u * 65536u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xr[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u + 4u, so + (u * 4 + i) * 65536u, 0));
            xe[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u + 4u, so + (8 + u) * 65536u, 0));
        }
    };
    auto a_load = [&](int slot, int c, int ky) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                af[slot][b][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rs, lane * 16u, wbase + (unsigned)(((c * 3 + ky) * 2 + b) * 2 + q) * 1024u, 0));
    };
    auto piece = [&](int u, int c, char *vbuf) __attribute__((always_inline)) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = c == 0 ? xr[u][i].x - xr[u][i].z : c == 1 ? xr[u][i].y + xr[u][i].z : c == 2 ? xr[u][i].z - xr[u][i].y : xr[u][i].y - xr[u][i].w;
        unsigned h0, h1, l0, l1;
        asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\tv_fma_mixhi_f16 %0, %5, %8, 0\n\tv_fma_mixlo_f16 %1, %6, %8, 0\n\tv_fma_mixhi_f16 %1, %7, %8, 0\n\t"
            "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(sv));
        *reinterpret_cast<u32x2 *>(vbuf + ((tid + u * NT) * 8 + c * 2 * 4096) % kV) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(vbuf + ((tid + u * NT) * 8 + (c * 2 + 1) * 4096) % kV) = u32x2{l0, l1};
    };
    auto piece_one = [&](int u, int c, char *vbuf) __attribute__((always_inline)) {
        const float va = c ? xe[u].z - xe[u].y : xe[u].x - xe[u].z, vb = c ? xe[u].y - xe[u].w : xe[u].y + xe[u].z;
        unsigned ha, hb, la, lb;
        asm("v_fma_mixlo_f16 %0, %4, %6, 0\n\tv_fma_mixlo_f16 %1, %5, %6, 0\n\t"
            "v_fma_mixlo_f16 %2, %4, %6, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %3, %5, %6, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
            : "=&v"(ha), "=&v"(hb), "=&v"(la), "=&v"(lb) : "v"(va), "v"(vb), "v"(sv));
        char *dst = vbuf + ((tid + u * NT) * 2 + 36864) % kV;
        *reinterpret_cast<unsigned short *>(dst) = (unsigned short)ha;
        *reinterpret_cast<unsigned short *>(dst + 512) = (unsigned short)la;
        *reinterpret_cast<unsigned short *>(dst + 1024) = (unsigned short)hb;
        *reinterpret_cast<unsigned short *>(dst + 1536) = (unsigned short)lb;
    };
    auto b_read = [&](int slot, int buf, int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            bq[slot][q] = *reinterpret_cast<const f16x8 *>(lds + buf * kV + ((wave & 3) * 8192 + blk * 640 + q * 4096 + lane * 16) % kV);
    };
    // ---- prologue
    x_load(0);
    a_load(0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) piece(u, c, lds);
        piece_one(u, 0, lds), piece_one(u, 1, lds);
    }
    if (chunks > 1) x_load(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    b_read(0, 0, 0);
    const long long t1 = clock64();
    // ---- chunks
    constexpr int NPIECE = 6 * U;                 // staging pieces of a chunk: behind MFMAs 1, 1 + STEP, ...
    constexpr int STEP = 64 / NPIECE;
    for (int c = 0; c < chunks; ++c) {
        char *vnext = lds + ((c + 1) & 1) * kV;
        const int buf = c & 1;
#pragma unroll
        for (int blk = 0; blk < 12; ++blk) {
            if (blk == 11) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if (blk < 11) b_read((blk & 1) ^ 1, buf, blk + 1);
            else b_read(0, buf ^ 1, 0);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const int b = m & 1, t = m >> 1, ky = blk >> 2, j = blk & 3;
                __builtin_amdgcn_sched_barrier(0);
                acc[b * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ky & 1][b][t == 0 ? 1 : 0], bq[blk & 1][t == 1 ? 1 : 0], acc[b * 4 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int s = blk * 6 + m;
                if (j == 0 && m == 0) a_load((ky & 1) ^ 1, ky < 2 ? c : c + 1, ky < 2 ? ky + 1 : 0);
                if (s >= 1 && (s - 1) % STEP == 0 && (s - 1) / STEP < NPIECE) {
                    const int k = (s - 1) / STEP, u = k % U, q = k / U;
                    if (q < 4) piece(u, q, vnext);
                    else piece_one(u, q - 4, vnext);
                    if (q == 5 && u == U - 1 && c + 2 < chunks) x_load(c + 2);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t2 = clock64();
    // ---- epilogue: two passes
    f32x4 *ex = reinterpret_cast<f32x4 *>(lds);
    constexpr int EXW = WAVES * 16 * 64;          // f32x4 slots of the exchange: 128 / 64 KB
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        f32x2 mk[16];
#pragma unroll
        for (int n = 0; n < 16; ++n)
            mk[n] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, tid * 8u, base + (unsigned)((pass * 16 + n) * NT * 8) + (32u << 20), 0));
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                ex[((wave * 4 + j) * 4 + rq) * 64 + lane] = f32x4{acc[pass * 4 + j][4 * rq], acc[pass * 4 + j][4 * rq + 1], acc[pass * 4 + j][4 * rq + 2], acc[pass * 4 + j][4 * rq + 3]};
        __syncthreads();
#pragma unroll
        for (int rqi = 0; rqi < 2; ++rqi) {
            f32x4 o[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 p[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) p[c] = ex[((((c * (WAVES / 4) + (wave & (WAVES / 4 - 1))) % WAVES) * 4 + s) * 4 + rqi) * 64 % EXW + lane];
                o[s][0] = (p[0] + p[1] + p[2]) * sv, o[s][1] = (p[1] - p[2] - p[3]) * sv;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    f32x2 v = {o[y][0][e], o[y][1][e]};
                    const f32x2 m = mk[(rqi * 4 + e) * 2 + y];
                    v.x = m.x > 0.f ? v.x : 0.f, v.y = m.y > 0.f ? v.y : 0.f;
                    v.x = fmaxf(v.x, -1e30f), v.y = fmaxf(v.y, -1e30f);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ro, tid * 8u,
                                                          base + (unsigned)(((pass * 2 + rqi) * 8 + e * 2 + y) * NT * 8), 0);
                }
            }
        }
    }
    if (blockIdx.x == gridDim.x - 3 && tid == 0) {
        g_cycles[0] = t1 - t0, g_cycles[1] = t2 - t1, g_cycles[2] = clock64() - t2;
    }
}

template <int WAVES> static float run(int items8, int chunks, int reps, long long *cyc, int stag_shift = 0, int stag_ticks = 0) {
    constexpr int NT = WAVES * 64;
    const unsigned bytes = 1u << 30;
    float *in, *out;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMemset(in, 0, bytes);
    auto kern = item<WAVES>;
    const int lds = WAVES == 8 ? 131072 : 81920;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        printf("hipFuncSetAttribute failed\n");
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, lds);
    const int grid = WAVES == 8 ? items8 : 2 * items8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, out, in, bytes, chunks, 1.f, stag_shift, stag_ticks);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, out, in, bytes, chunks, 1.f, stag_shift, stag_ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(long long) * 3);
    cyc[3] = per_cu;
    hipFree(in), hipFree(out);
    return ms / reps * 1e3f;
}

int main() {
    // (work items of the eight-wave form, chunks of 16 channels): 64 -> 64 @ 1024^2 in 128-channel-equivalent items,
    // 128 -> 128 @ 512^2, 256 -> 256 @ 256^2, 512 -> 512 @ 128^2, and the same with fewer rounds
    const int cases[][2] = {{2048, 4}, {2048, 8}, {1024, 16}, {512, 32}, {256, 32}, {288, 16}};
    for (auto &cs : cases) {
        long long c8[4], c4[4];
        const float us8 = run<8>(cs[0], cs[1], 10, c8), us4 = run<4>(cs[0], cs[1], 10, c4);
        const double mfma_us = (double)cs[0] / 256 * cs[1] * 2 * 72 * 32 / 2.4e3;     // two waves per SIMD, 72 MFMAs of 32 cycles, 2.4 GHz
        printf("%5d items x %2d chunks: one 8-wave workgroup per CU %7.1f us (%lld / %lld / %lld cycles, %lld per CU), two 4-wave per CU %7.1f us "
               "(%lld / %lld / %lld, %lld per CU)  x%.3f; pure matrix work %.1f us\n",
               cs[0], cs[1], us8, c8[0], c8[1], c8[2], c8[3], us4, c4[0], c4[1], c4[2], c4[3], us8 / us4, mfma_us);
        // staggered: half an item of the four-wave form = us4 / (items per CU slot) / 2
        const double item_us = us4 / (2.0 * cs[0] / 512.0);
        for (int shift : {0, 3, 8})
            for (double frac : {0.3, 0.5}) {
                const int ticks = (int)(item_us * frac * 100.0);
                const float s4 = run<4>(cs[0], cs[1], 10, c4, shift, ticks);
                printf("      staggered (bit %d of the workgroup index, %.1f us): two 4-wave per CU %7.1f us  x%.3f  (%lld / %lld / %lld)\n",
                       shift, ticks / 100.0, s4, us8 / s4, c4[0], c4[1], c4[2]);
            }
        const float s8 = run<8>(cs[0], cs[1], 10, c8, 0, (int)(item_us * 100.0));
        printf("      staggered eight-wave form (bit 0, %.1f us): %7.1f us\n", item_us, s8);
    }
    return 0;
}
