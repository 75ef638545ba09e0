// Can a small-footprint kernel run in the shadow of the convolution workgroups?
//
// A conv_wino2 workgroup holds 128 of a CU's 160 KB of LDS and 2 x 232 of a SIMD's 512 registers per
// lane; what is left is 32 KB and 48 registers.  The Gram / SYMM kernels (68 / 30 KB, ~190 registers)
// therefore never share a CU with it and take the GPU for themselves (0.57 ms of a 6.3 ms tile).
// This probe launches a streaming + bf16-MFMA kernel built to fit into the leftovers (<= 48
// registers, 16 KB of LDS) on a second stream beside a run of real convolution launches and prints
// the three times: convolutions alone, probe alone, both together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc \
//         tools/ubench/shadow.hip style_transfer_amd/csrc/conv_wino4.hip -o build_ubench/shadow
#include "../../style_transfer_amd/csrc/conv_wino2.hip"

#include <cstdarg>
#include <vector>

namespace stx {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
const char *sw_env(const char *name) { return getenv(name); }     // (the harness reads the environment as it is)
void sw_reread() {}
int splitk_reduce_launch(hipStream_t, const ConvProblem &, int) { return 0; }
}  // namespace stx

typedef short bf16x8s __attribute__((ext_vector_type(8)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

// Streams `n_vec` 16-byte vectors (each workgroup a contiguous slice, MODE & 1: through LDS) and
// feeds them to v_mfma_f32_16x16x32_bf16 (MODE & 2), 4 accumulator registers.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void shadow_kernel(
    const f32x4s *__restrict__ src, size_t n_vec, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) f32x4s stage[1024];      // 16 KB
    const size_t per = n_vec / gridDim.x;
    const f32x4s *p = src + blockIdx.x * per;
    f32x4s acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
        f32x4s v0 = p[i], v1 = p[i + 256], v2 = p[i + 512], v3 = p[i + 768];
        if (MODE & 1) {
            stage[threadIdx.x] = v0, stage[threadIdx.x + 256] = v1;
            stage[threadIdx.x + 512] = v2, stage[threadIdx.x + 768] = v3;
            __syncthreads();
            v0 = stage[(threadIdx.x + 64) & 1023], v1 = stage[(threadIdx.x + 320) & 1023];
            v2 = stage[(threadIdx.x + 576) & 1023], v3 = stage[(threadIdx.x + 832) & 1023];
            __syncthreads();
        }
        if (MODE & 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, v0),
                                                          __builtin_bit_cast(bf16x8s, v1), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, v2),
                                                          __builtin_bit_cast(bf16x8s, v3), acc, 0, 0, 0);
        } else {
            acc += v0 + v1 + v2 + v3;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int MODE>
static void run(const char *label) {
    using namespace stx;
    const int K = 512, M = 512, H = 128, W = 128;
    const size_t xn = (size_t)K * H * W, yn = (size_t)M * H * W, wn = wino2_packed_floats(K, M);
    float *x, *y, *w, *sink;
    f32x4s *big;
    const size_t big_bytes = 268435456;                       // the 64 x 1024^2 blob of conv1_1
    hipMalloc(&x, xn * 4), hipMalloc(&y, yn * 4), hipMalloc(&w, wn * 4), hipMalloc(&sink, 64);
    hipMalloc(&big, big_bytes);
    hipMemset(x, 0, xn * 4), hipMemset(w, 0, wn * 4), hipMemset(big, 0, big_bytes);
    ConvProblem p{};
    p.x = x, p.w = w, p.y = y, p.K = K, p.M = M, p.H = H, p.W = W, p.ksize = 3, p.relu = 1;
    p.epilogue = kEpiForward;
    const ConvConfig cfg = wino2_config(0);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking), hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0), hipEventCreate(&a1), hipEventCreate(&b0), hipEventCreate(&b1);
    const int convs = 10, probes = 12;
    auto conv_run = [&]() {
        hipEventRecord(a0, sa);
        for (int i = 0; i < convs; ++i) wino2_launch(sa, cfg, p, 1);
        hipEventRecord(a1, sa);
    };
    auto probe_run = [&]() {
        hipEventRecord(b0, sb);
        for (int i = 0; i < probes; ++i) shadow_kernel<MODE><<<2048, 256, 0, sb>>>(big, big_bytes / 16, sink);
        hipEventRecord(b1, sb);
    };
    float ta, tb, ta2, tb2;
    conv_run(), hipDeviceSynchronize();
    probe_run(), hipDeviceSynchronize();
    conv_run(), hipDeviceSynchronize();
    hipEventElapsedTime(&ta, a0, a1);
    probe_run(), hipDeviceSynchronize();
    hipEventElapsedTime(&tb, b0, b1);
    conv_run(), probe_run(), hipDeviceSynchronize();
    hipEventElapsedTime(&ta2, a0, a1), hipEventElapsedTime(&tb2, b0, b1);
    printf("%-28s convolutions alone %.3f ms, probe alone %.3f ms (%.2f TB/s); together: convolutions %.3f ms, probe %.3f ms\n",
           label, ta, tb, (double)big_bytes * probes / tb / 1e9, ta2, tb2);
    hipFree(x), hipFree(y), hipFree(w), hipFree(sink), hipFree(big);
}

int main() {
    run<0>("stream + add");
    run<2>("stream + bf16 MFMA");
    run<3>("stream + LDS + bf16 MFMA");
    return 0;
}
