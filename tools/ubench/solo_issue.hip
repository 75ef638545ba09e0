// Microbenchmark: ONE wave per SIMD (256-thread workgroup, one per CU) streaming
// v_mfma_f32_32x32x2_f32 over 16 accumulators with F filler instructions of one kind after every
// MFMA.  Prints cycles per MFMA: 64 means the fillers hide in the MFMA's shadow.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/solo_issue.hip -o /tmp/solo_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ long long g_t[4];

enum { kNone, kAdd, kPkAdd, kDsWrite128, kDsRead128, kBufLoad128, kMix, kBurstPk, kBurstAdd, kBurstPkDep };

template <int MODE, int F>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a * i, b * i};
    f32x4 w4 = {a, b, a, b};
    f32x4 rd[4] = {w4, w4, w4, w4};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 1 << 20, 0x00020000);
    u32x4 ld[4] = {};
    f32x4 *l4 = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < 8192; i += 256) l4[i] = w4;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int j = (i * F + f);
                if (MODE == kAdd) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7].x) : "v"(v[(j + 1) & 7].y));
                if (MODE == kPkAdd) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
                if (MODE == kDsWrite128) l4[(j & 7) * 256 + threadIdx.x] = w4;
                if (MODE == kDsRead128) {
                    f32x4 t = l4[(j & 7) * 256 + wave * 64 + (lane & 31)];
                    asm volatile("" :: "v"(t));       // discarded: only the issue matters
                }
                if (MODE >= kBurstPk) continue;
                if (MODE == kBufLoad128)
                    ld[j & 3] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(threadIdx.x * 16u), (unsigned)((j & 15) * 4096), 0);
            }
            // one burst of F vector instructions per 16 MFMAs
            if (MODE >= kBurstPk && i == 7) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    if (MODE == kBurstPk) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(v[(f + 1) & 7]));
                    if (MODE == kBurstAdd) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[f & 7].x) : "v"(v[(f + 1) & 7].y));
                    if (MODE == kBurstPkDep) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v[(f + 3) & 7]) : "v"(v[f & 7]), "v"(v[(f + 1) & 7]));
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) g_t[wave] = t1 - t0;
    float s = w4.x + rd[0].x + rd[1].y + rd[2].z + rd[3].w;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    for (int i = 0; i < 4; ++i) s += __builtin_bit_cast(f32x4, ld[i]).x;
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int MODE, int F> void run(const char *name) {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&in, 1 << 20);
    hipMemset(in, 0, 1 << 20);
    const int iters = 256;
    auto kern = k<MODE, F>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    kern<<<256, 256, 131072>>>(out, in, iters, 1.f, 1.f);
    kern<<<256, 256, 131072>>>(out, in, iters, 1.000001f, 0.999999f);
    hipDeviceSynchronize();
    long long t[4];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
    printf("%-22s x%d per MFMA: %6.1f cycles per MFMA, %7.1f per 16 MFMAs (waves %lld %lld %lld %lld)\n", name, F,
           (double)t[0] / (iters * 16.0), (double)t[0] / iters, t[0], t[1], t[2], t[3]);
    hipFree(out), hipFree(in);
}

// Skeleton of one 4-wave Winograd chunk: 64 MFMAs (4 k-steps x 16 accumulators) whose operands
// come from ds_read_b128 issued one k-step ahead, with the staging of the next chunk dealt out
// one piece per MFMA: 8 filter loads, 8 filter LDS writes, 8 patch loads, 64 plain adds (or 32
// packed adds), 8 patch LDS writes.  VAR: 0 plain adds, 1 packed adds, 2 no adds, 3 operand reads
// only.
template <int VAR>
__global__ __launch_bounds__(256) void chunk_k(float *out, const float *in, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 1 << 20, 0x00020000);
    f32x4 *l4 = reinterpret_cast<f32x4 *>(lds);
    f32x4 w4 = {a, b, a, b};
    for (int i = threadIdx.x; i < 8192; i += 256) l4[i] = w4;
    u32x4 wreg[8];
    f32x4 xreg[8];
    f32x4 vq[8];
    for (int i = 0; i < 8; ++i) wreg[i] = u32x4{0, 0, 0, 0}, xreg[i] = w4, vq[i] = w4;
    float t[32];
    for (int i = 0; i < 32; ++i) t[i] = a * i;
    f32x4 av[2][2], bv[2][2];
    const int rbase = wave * 64 + (lane & 31);
    av[0][0] = l4[rbase], av[0][1] = l4[rbase + 256], bv[0][0] = l4[rbase + 512], bv[0][1] = l4[rbase + 768];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int p = s * 16 + m;            // piece slot 0..63
                const int i = m & 1, j = (m >> 1) & 1, c = m >> 2;
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i][c], bv[s & 1][j][c], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m < 4) {                          // operands of the next k-step
                    const int o = ((s + 1) & 3) * 1024 + rbase;
                    if (m == 0) av[(s + 1) & 1][0] = l4[o];
                    if (m == 1) av[(s + 1) & 1][1] = l4[o + 256];
                    if (m == 2) bv[(s + 1) & 1][0] = l4[o + 512];
                    if (m == 3) bv[(s + 1) & 1][1] = l4[o + 768];
                }
                if (VAR == 3) continue;
                if (p >= 4 && p < 12) l4[4096 + (p - 4) * 256 + threadIdx.x] = __builtin_bit_cast(f32x4, wreg[p - 4]);
                if (p >= 12 && p < 20) wreg[p - 12] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(threadIdx.x * 16u), (unsigned)((p - 12) * 4096 + (it & 7) * 32768), 0);
                if (p == 20) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(xreg[q]));
                }
                if (VAR == 0 && p >= 20 && p < 52) {  // 64 plain adds, two per slot
                    const int q = (p - 20) * 2;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int u = q + e;
                        if (u < 32) t[u] = xreg[(u >> 2) & 7][u & 3] - xreg[((u >> 2) + 2) & 7][u & 3];
                        else vq[(u - 32) >> 2][(u - 32) & 3] = t[u - 32] + t[(u - 31) & 31];
                    }
                }
                if (VAR == 1 && p >= 20 && p < 52) {  // 32 packed adds, one per slot
                    const int u = p - 20;
                    f32x2 r, x0, x1;
                    if (u < 16) {
                        x0 = (u & 1) ? xreg[(u >> 1) & 7].zw : xreg[(u >> 1) & 7].xy;
                        x1 = (u & 1) ? xreg[((u >> 1) + 2) & 7].zw : xreg[((u >> 1) + 2) & 7].xy;
                        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x0), "v"(x1));
                        t[2 * u] = r.x, t[2 * u + 1] = r.y;
                    } else {
                        const int k = u - 16;
                        x0 = f32x2{t[2 * k], t[2 * k + 1]}, x1 = f32x2{t[(2 * k + 2) & 31], t[(2 * k + 3) & 31]};
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
                        if (k & 1) vq[k >> 1].zw = r; else vq[k >> 1].xy = r;
                    }
                }
                if (p >= 52 && p < 60) xreg[p - 52] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(threadIdx.x * 16u + 64), (unsigned)((p - 52) * 4096 + (it & 7) * 32768), 0));
                if (p >= 56) l4[6144 + (p - 56) * 256 + threadIdx.x] = vq[p - 56];
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) g_t[wave] = t1 - t0;
    float sum = w4.x;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += vq[i].x + xreg[i].y + __builtin_bit_cast(f32x4, wreg[i]).z;
    for (int i = 0; i < 32; ++i) sum += t[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum + lds[threadIdx.x];
}

template <int VAR> void run_chunk(const char *name) {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&in, 1 << 20);
    hipMemset(in, 0, 1 << 20);
    const int iters = 64;
    auto kern = chunk_k<VAR>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    kern<<<256, 256, 131072>>>(out, in, iters, 1.f, 1.f);
    kern<<<256, 256, 131072>>>(out, in, iters, 1.000001f, 0.999999f);
    hipDeviceSynchronize();
    long long t[4];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
    printf("%-30s: %7.1f cycles per 64-MFMA chunk = %5.1f per MFMA (waves %lld %lld %lld %lld)\n", name,
           (double)t[0] / iters, (double)t[0] / (iters * 64.0), t[0], t[1], t[2], t[3]);
    hipFree(out), hipFree(in);
}

int main() {
    run<kNone, 0>("bare MFMA");
    run<kAdd, 1>("v_add_f32");
    run<kAdd, 2>("v_add_f32");
    run<kAdd, 4>("v_add_f32");
    run<kAdd, 8>("v_add_f32");
    run<kPkAdd, 1>("v_pk_add_f32");
    run<kPkAdd, 2>("v_pk_add_f32");
    run<kPkAdd, 4>("v_pk_add_f32");
    run<kDsWrite128, 1>("ds_write_b128");
    run<kDsWrite128, 2>("ds_write_b128");
    run<kDsRead128, 1>("ds_read_b128");
    run<kDsRead128, 2>("ds_read_b128");
    run<kBufLoad128, 1>("buffer_load_b128");
    run<kBufLoad128, 2>("buffer_load_b128");
    run<kBurstPk, 4>("burst pk /16 MFMA");
    run<kBurstPk, 8>("burst pk /16 MFMA");
    run<kBurstPk, 16>("burst pk /16 MFMA");
    run<kBurstPk, 32>("burst pk /16 MFMA");
    run<kBurstAdd, 16>("burst add /16 MFMA");
    run<kBurstAdd, 32>("burst add /16 MFMA");
    run<kBurstPkDep, 16>("burst pk mods /16 MFMA");
    run_chunk<0>("chunk skeleton, plain adds");
    run_chunk<1>("chunk skeleton, pk adds");
    run_chunk<2>("chunk skeleton, no VALU");
    run_chunk<3>("chunk skeleton, reads only");
    return 0;
}
