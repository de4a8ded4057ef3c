// valu_mix.hip — how do full-rate (v_xor_b32, ~2 cyc) and half-rate (v_bcnt_u32_b32, ~4 cyc) ops share a SIMD?
// P1: every wave runs G xors then G bcnts (independent registers), G = 1..32.
// P2: even waves run only xors, odd waves only bcnts.
// Reports SIMD-cycles per (xor + bcnt) pair; the ideal is 2 + 4 = 6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int G, int SPLIT>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t seed) {
    uint32_t a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = i; }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (SPLIT) {
        // 32 pairs' worth per iteration, but this wave issues only one kind, twice as many of it
        if (wave & 1) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i]));
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i]));
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) {
#pragma unroll
                for (int i = 0; i < G; ++i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[g * G + i]) : "v"(b[(g * G + i + 7) & 31]));
#pragma unroll
                for (int i = 0; i < G; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(b[g * G + i]) : "v"(a[g * G + i]));
            }
        }
    }
    if (threadIdx.x == 0) {
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int G, int SPLIT>
int run(const char* name, int iters) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    unsigned long long* clk; CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8));
    std::vector<unsigned long long> hclk(2 * cus * 8);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bpc : {2, 4, 5, 8}) {
        const int grid = cus * bpc;
        hipLaunchKernelGGL((k<G, SPLIT>), dim3(grid), dim3(256), 0, 0, out, clk, 50, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<G, SPLIT>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(hclk.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int g = 0; g < grid; ++g) if (hclk[2 * g + 1]) f.push_back((double)hclk[2 * g] / (double)hclk[2 * g + 1] * 100e6);
        std::sort(f.begin(), f.end());
        const double ghz = f.empty() ? 0 : f[f.size() / 2] * 1e-9;
        const double pairs = (double)grid * 4 * iters * 32.0;      // (xor,bcnt) pairs executed (wave-level)
        const double per_s = pairs / (ms * 1e-3);
        printf("%-10s waves/SIMD=%d %8.3f ms  clk %.3f GHz  %.2f SIMD-cycles per (xor+bcnt) pair\n", name, bpc, ms, ghz,
               ghz * 1e9 / (per_s / (cus * 4.0)));
    }
    CK(hipFree(out)); CK(hipFree(clk));
    return 0;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    if (run<1, 0>("G=1", iters)) return 1;
    if (run<2, 0>("G=2", iters)) return 1;
    if (run<4, 0>("G=4", iters)) return 1;
    if (run<8, 0>("G=8", iters)) return 1;
    if (run<16, 0>("G=16", iters)) return 1;
    if (run<32, 0>("G=32", iters)) return 1;
    if (run<32, 1>("split", iters)) return 1;
    return 0;
}
