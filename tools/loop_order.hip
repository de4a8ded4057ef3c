// loop_order.hip — instruction-ORDER experiments for the real inner loop (8 query rows x 2 train rows = 16 distances
// per block, train words as SGPR operands), no memory traffic.  Same instruction multiset in every pattern:
// per distance 8 v_xor_b32 + 8 v_bcnt_u32_b32 + 1 v_lshl_or_b32 + 0.5 v_min3_u32.
// Reports SIMD-cycles per 64 distances (one wave-distance) and distances/s for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define XOR(dst, s, v) asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(dst) : "s"(s), "v"(v))
#define BCNT0(dst, x) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(dst) : "v"(x))
#define BCNT(acc, x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x))
#define PACK(d, t) asm volatile("v_lshl_or_b32 %0, %0, 22, %1" : "+v"(d) : "s"(t))
#define MIN3(b, x, y) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(b) : "v"(x), "v"(y))

// PAT 0: compiler-scheduled (non-volatile asm for bcnt, C for the rest)  — what the kernel does today
// PAT 1: per distance: x0 b0 x1 b1 ... x7 b7 (strict alternation, each bcnt right after its xor)
// PAT 2: per distance: 8 xors then the 8-bcnt chain
// PAT 3: two distances interleaved, strict alternation: xA0 bB.. (xor of one distance between the bcnts of the other)
// PAT 4: per distance: x0 x1 b0 x2 b1 ... (xor one ahead of its bcnt: no back-to-back dependency)
template <int PAT>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t seed) {
    uint32_t q[8][8], best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = ~0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) q[j][i] = threadIdx.x * 2654435761u + (j * 8 + i) * 40503u + seed; }
    uint32_t s = seed;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        uint32_t A[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { A[i] = s; s = s * 1664525u + 1013904223u; }
        const uint32_t t = (uint32_t)it * 2;
        if (PAT == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t d0, d1, x;
                x = q[j][0] ^ A[0]; asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d0) : "v"(x));
                x = q[j][0] ^ A[8]; asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d1) : "v"(x));
#pragma unroll
                for (int kk = 1; kk < 8; ++kk) {
                    x = q[j][kk] ^ A[kk]; asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d0) : "v"(x));
                    x = q[j][kk] ^ A[8 + kk]; asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d1) : "v"(x));
                }
                uint32_t k0 = (d0 << 22) | t, k1 = (d1 << 22) | (t + 1);
                best[j] = min(min(best[j], k0), k1);
            }
        } else if (PAT == 1 || PAT == 2 || PAT == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t d[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint32_t x[8];
                    if (PAT == 1) {
                        XOR(x[0], A[r * 8], q[j][0]); BCNT0(d[r], x[0]);
#pragma unroll
                        for (int kk = 1; kk < 8; ++kk) { XOR(x[kk], A[r * 8 + kk], q[j][kk]); BCNT(d[r], x[kk]); }
                    } else if (PAT == 2) {
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) XOR(x[kk], A[r * 8 + kk], q[j][kk]);
                        BCNT0(d[r], x[0]);
#pragma unroll
                        for (int kk = 1; kk < 8; ++kk) BCNT(d[r], x[kk]);
                    } else {
                        XOR(x[0], A[r * 8], q[j][0]);
                        XOR(x[1], A[r * 8 + 1], q[j][1]);
                        BCNT0(d[r], x[0]);
#pragma unroll
                        for (int kk = 2; kk < 8; ++kk) { XOR(x[kk], A[r * 8 + kk], q[j][kk]); BCNT(d[r], x[kk - 1]); }
                        BCNT(d[r], x[7]);
                    }
                }
                PACK(d[0], t); PACK(d[1], t + 1);
                MIN3(best[j], d[0], d[1]);
            }
        } else if (PAT == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t d0, d1, x0[8], x1[8];
                XOR(x0[0], A[0], q[j][0]);
                XOR(x1[0], A[8], q[j][0]); BCNT0(d0, x0[0]);
#pragma unroll
                for (int kk = 1; kk < 8; ++kk) {
                    XOR(x0[kk], A[kk], q[j][kk]);
                    if (kk == 1) BCNT0(d1, x1[0]); else BCNT(d1, x1[kk - 1]);
                    XOR(x1[kk], A[8 + kk], q[j][kk]);
                    BCNT(d0, x0[kk]);
                }
                BCNT(d1, x1[7]);
                PACK(d0, t); PACK(d1, t + 1);
                MIN3(best[j], d0, d1);
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
    for (int j = 0; j < 8; ++j) r += best[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int PAT>
int run(const char* name, int iters) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    unsigned long long* clk; CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8));
    std::vector<unsigned long long> hclk(2 * cus * 8);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bpc : {2, 3, 4, 5}) {
        const int grid = cus * bpc;
        hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), 0, 0, out, clk, 50, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(hclk.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int g = 0; g < grid; ++g) if (hclk[2 * g + 1]) f.push_back((double)hclk[2 * g] / (double)hclk[2 * g + 1] * 100e6);
        std::sort(f.begin(), f.end());
        const double ghz = f.empty() ? 0 : f[f.size() / 2] * 1e-9;
        const double wdist = (double)grid * 4 * iters * 16.0;      // wave-distances (64 distances each)
        const double per_s = wdist / (ms * 1e-3);
        printf("%-22s waves/SIMD=%d %8.3f ms  clk %.3f GHz  %.1f SIMD-cycles per 64 distances   %.3e distances/s\n", name, bpc, ms, ghz,
               ghz * 1e9 / (per_s / (cus * 4.0)), per_s * 64.0);
    }
    CK(hipFree(out)); CK(hipFree(clk));
    return 0;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    if (run<0>("0 compiler-scheduled", iters)) return 1;
    if (run<1>("1 x b x b (dependent)", iters)) return 1;
    if (run<2>("2 8x then 8b", iters)) return 1;
    if (run<3>("3 two-dist alternate", iters)) return 1;
    if (run<4>("4 xor one ahead", iters)) return 1;
    return 0;
}
