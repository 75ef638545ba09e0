// Microbenchmark: how long does a CU stand empty between two workgroups of one launch?
// Every workgroup stamps the constant 100 MHz clock when its first instruction runs and right before
// it ends, with the CU it ran on (HW_REG_HW_ID: CU 11:8, SH 12, SE 15:13; HW_REG_XCC_ID); the host sorts
// the stamps per CU and prints the distribution of (start of workgroup n+1) - (end of workgroup n).
// Resource shapes: threads per workgroup, dynamic LDS bytes, a 200-register body or a lean one.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long t0, t1; unsigned hw, xcc; };

template <int NT, bool FAT, bool STORE>
__global__ __launch_bounds__(NT) void k(Stamp *st, float *out, long long spin, float a) {
    const unsigned long long t0 = wall_clock64();
    extern __shared__ float lds[];
    f32x16 acc[FAT ? 12 : 1];
    for (int i = 0; i < (FAT ? 12 : 1); ++i) for (int r = 0; r < 16; ++r) acc[i][r] = a * i;
    const long long c0 = clock64();
    while (clock64() - c0 < spin) {
        for (int i = 0; i < (FAT ? 12 : 1); ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < (FAT ? 12 : 1); ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (STORE) {                         // an epilogue with 64 KB of stores per workgroup in flight at the end
        for (int i = 0; i < 16384 / NT; ++i) out[((size_t)blockIdx.x * 16384 + i * NT + threadIdx.x)] = lds[(threadIdx.x + i) % NT];
    } else if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), hw, xcc};
    }
}

template <int NT, bool FAT, bool STORE>
void run(const char *name, int lds_bytes, int blocks, long long spin) {
    Stamp *st; float *out;
    hipMalloc(&st, blocks * sizeof(Stamp));
    hipMalloc(&out, (size_t)blocks * 16384 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<NT, FAT, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int rep = 0; rep < 2; ++rep) k<NT, FAT, STORE><<<blocks, NT, lds_bytes>>>(st, out, spin, 1.f);
    hipDeviceSynchronize();
    std::vector<Stamp> h(blocks);
    hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<Stamp>> per_cu;
    for (auto &s : h) per_cu[(s.xcc & 0xf) << 16 | (s.hw & 0xff00)] .push_back(s);
    std::vector<double> gaps, lens;
    size_t most = 0;
    for (auto &kv : per_cu) {
        auto &v = kv.second;
        most = std::max(most, v.size());
        std::sort(v.begin(), v.end(), [](const Stamp &x, const Stamp &y) { return x.t0 < y.t0; });
        for (size_t i = 0; i < v.size(); ++i) {
            lens.push_back((v[i].t1 - v[i].t0) * 0.01);
            // (with two workgroups resident the next one may start before this one ends: negative = overlap)
            if (i + 1 < v.size()) gaps.push_back(((double)v[i + 1].t0 - (double)v[i].t1) * 0.01);
        }
    }
    std::sort(gaps.begin(), gaps.end());
    std::sort(lens.begin(), lens.end());
    unsigned long long first = ~0ull, last = 0;
    for (auto &s : h) { first = std::min(first, s.t0); last = std::max(last, s.t1); }
    printf("%-44s %5d wgs on %3zu CUs (most %2zu): body %.2f us; gap median %.2f us, p10 %.2f, p90 %.2f, max %.2f; launch %.1f us\n", name, blocks,
           per_cu.size(), most, lens[lens.size() / 2], gaps.empty() ? 0. : gaps[gaps.size() / 2], gaps.empty() ? 0. : gaps[gaps.size() / 10],
           gaps.empty() ? 0. : gaps[gaps.size() * 9 / 10], gaps.empty() ? 0. : gaps.back(), (last - first) * 0.01);
    hipFree(st); hipFree(out);
}

int main() {
    const long long spin = 40000;    // ~18 us
    run<512, true, false>("512 threads, 128 KB LDS, 200 registers", 131072, 4096, spin);
    run<512, true, true>("  + 64 KB of stores at the end", 131072, 4096, spin);
    run<512, true, false>("512 threads, 64 KB LDS, 200 registers", 65536, 4096, spin);
    run<512, true, false>("512 threads, 1 KB LDS, 200 registers", 1024, 4096, spin);
    run<256, true, false>("256 threads, 64 KB LDS, 200 registers", 65536, 8192, spin);
    run<256, true, true>("  + 64 KB of stores at the end", 65536, 8192, spin);
    run<512, false, false>("512 threads, 128 KB LDS, lean", 131072, 4096, spin);
    run<256, false, false>("256 threads, 1 KB LDS, lean", 1024, 8192, spin);
    run<64, false, false>("64 threads, 1 KB LDS, lean", 1024, 32768, spin);
    return 0;
}
