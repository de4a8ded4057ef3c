// valu_phase.hip — can two waves on one SIMD overlap full-rate (xor) and half-rate (bcnt) work if their phases are
// offset?  512-thread workgroups: wave w and wave w+4 share a SIMD.
//  M0: every wave loops [G xors][G bcnts]          M1: waves >= 4 loop [G bcnts][G xors] (half a period later)
//  M2: G = 1 alternation (best single-stream order)  M3: waves < 4 only xors, waves >= 4 only bcnts (reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int G>
__device__ __forceinline__ void xors(uint32_t (&a)[32], uint32_t (&b)[32], int g) {
#pragma unroll
    for (int i = 0; i < G; ++i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[g * G + i]) : "v"(b[(g * G + i + 7) & 31]));
}
template <int G>
__device__ __forceinline__ void bcnts(uint32_t (&a)[32], uint32_t (&b)[32], int g) {
#pragma unroll
    for (int i = 0; i < G; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(b[g * G + i]) : "v"(a[g * G + i]));
}

template <int G, int MODE>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = i; }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool hi = wave >= 4;
    if (MODE == 3) {
        if (hi) { for (int it = 0; it < iters; ++it) { bcnts<32>(a, b, 0); bcnts<32>(a, b, 0); } }
        else    { for (int it = 0; it < iters; ++it) { xors<32>(a, b, 0); xors<32>(a, b, 0); } }
    } else if (MODE == 1 && hi) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) { bcnts<G>(a, b, g); xors<G>(a, b, g); }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) { xors<G>(a, b, g); bcnts<G>(a, b, g); }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int G, int MODE>
int run(const char* name, int iters) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 512 * cus * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bpc : {1, 2, 4}) {
        const int grid = cus * bpc;
        hipLaunchKernelGGL((k<G, MODE>), dim3(grid), dim3(512), 0, 0, out, 50, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<G, MODE>), dim3(grid), dim3(512), 0, 0, out, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double pairs = (double)grid * 8 * iters * 32.0;
        printf("%-28s waves/SIMD=%d %8.3f ms   %.3f ns*SIMD per (xor+bcnt) pair  (= %.2f cycles @2.4GHz)\n", name, bpc * 2, ms,
               ms * 1e6 / (pairs / (cus * 4.0)), ms * 1e6 / (pairs / (cus * 4.0)) * 2.4);
    }
    CK(hipFree(out));
    return 0;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    if (run<32, 0>("M0 G=32 same phase", iters)) return 1;
    if (run<32, 1>("M1 G=32 hi waves shifted", iters)) return 1;
    if (run<16, 0>("M0 G=16 same phase", iters)) return 1;
    if (run<16, 1>("M1 G=16 hi waves shifted", iters)) return 1;
    if (run<8, 0>("M0 G=8 same phase", iters)) return 1;
    if (run<8, 1>("M1 G=8 hi waves shifted", iters)) return 1;
    if (run<1, 0>("M2 G=1 alternate", iters)) return 1;
    if (run<1, 1>("M2' G=1 hi waves b-then-x", iters)) return 1;
    if (run<32, 3>("M3 split", iters)) return 1;
    return 0;
}
