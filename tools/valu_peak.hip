// valu_peak.hip — measures the integer VALU issue rate the Hamming path is bounded by on this chip:
// independent v_xor_b32 + v_bcnt_u32_b32 chains from registers, no memory traffic in the loop.
// Prints lane-ops/s for several waves-per-SIMD occupancies.  Usage: ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t q[8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { q[i] = threadIdx.x * 2654435761u + i * 40503u + seed; acc[i] = i; }
    uint32_t s = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) {          // xor (scalar operand) + accumulating bcnt: the real inner loop's mix
                    uint32_t x = q[i] ^ s;
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
                } else if (MODE == 1) {   // bcnt only
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                } else {                  // xor only
                    asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                }
            }
            s = s * 1664525u + 1013904223u;
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char* name, int ops_per_inner) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bpc = 1; bpc <= 8; ++bpc) {       // blocks of 256 threads per CU = waves per SIMD
        const int iters = 20000;
        const int grid = cus * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double laneops = (double)grid * 256 * iters * 64.0 * ops_per_inner;
        printf("%-10s waves/SIMD=%d  %.3f ms  %.3e lane-ops/s  (%.1f lanes/clk/CU @2.4GHz)\n", name, bpc, ms,
               laneops / (ms * 1e-3), laneops / (ms * 1e-3) / cus / 2.4e9);
    }
    CK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  gcn=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
    if (run<0>("xor+bcnt", 2)) return 1;
    if (run<1>("bcnt", 1)) return 1;
    if (run<2>("xor", 1)) return 1;
    return 0;
}
