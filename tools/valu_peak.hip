// valu_peak.hip — instruction-rate microbenchmark for the integer VALU ops the Hamming path is built from.
// For each op: a long unrolled stream of independent instructions from registers (no memory traffic), at 1..8 waves
// per SIMD.  Reports wave64-instructions/s, the in-kernel shader clock (s_memtime / s_memrealtime, 100 MHz) and the
// resulting cycles per wave-instruction per SIMD.  Usage: ./valu_peak [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum Mode { XOR_BCNT, BCNT, XOR_E32, XOR_E64, LSHL_OR, MIN3, MIN_E32, ADD3, BFI, ALIGNBIT, CMP_CNDMASK, CMP16_CNDMASK, LSHLREV_OR, REAL_MIX, REAL_MIX2, N_MODES };
static const char* kNames[] = {"xor+bcnt", "bcnt(vop3)", "xor(e32)", "xor(e64)", "lshl_or", "min3_u32", "min_u32(e32)", "add3_u32", "bfi_b32",
                               "alignbit", "cmp+cndmask", "cmp16+cndmsk", "lshlrev+or", "mix17.5", "mix_cmp16"};
// wave-instructions per inner step (8 lanes of the unrolled i-loop, see below)
static const int kOps[] = {16, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 16, 16, 140, 144};

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t seed) {
    uint32_t q[8], acc[8], b2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { q[i] = threadIdx.x * 2654435761u + i * 40503u + seed; acc[i] = i; b2[i] = ~0u; }
    uint32_t s = seed;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == REAL_MIX || MODE == REAL_MIX2) {
                // one train row against 8 query rows' worth of chains: 8 x (8 xor + 8 bcnt) + index bookkeeping
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint32_t d;
                    uint32_t x = q[0] ^ (s + j);
                    if (MODE == REAL_MIX) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x));
                    else asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "s"(s << 16));
#pragma unroll
                    for (int kk = 1; kk < 8; ++kk) { x = q[kk] ^ (s + j + kk); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x)); }
                    if (MODE == REAL_MIX) {
                        asm volatile("v_lshl_or_b32 %0, %0, 22, %1" : "+v"(d) : "s"(s));
                        if (j & 1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(acc[j >> 1]) : "v"(d), "v"(b2[j >> 1]));
                        else b2[j >> 1] = d;
                    } else {
                        asm volatile("v_cmp_lt_u16 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[j]) : "v"(d) : "vcc");
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == XOR_BCNT) { uint32_t x = q[i] ^ s; asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(x)); }
                    else if (MODE == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == XOR_E32) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == XOR_E64) asm volatile("v_xor_b32_e64 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(q[i]), "v"(b2[i]));
                    else if (MODE == MIN_E32) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(q[i]), "v"(b2[i]));
                    else if (MODE == BFI) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(q[i]), "v"(b2[i]));
                    else if (MODE == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(acc[i]) : "v"(q[i]));
                    else if (MODE == CMP_CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(q[i]) : "vcc");
                    else if (MODE == CMP16_CNDMASK) asm volatile("v_cmp_lt_u16 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(q[i]) : "vcc");
                    else if (MODE == LSHLREV_OR) asm volatile("v_lshlrev_b32 %0, 3, %0\n\tv_or_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(q[i]));
                }
            }
            s = s * 1664525u + 1013904223u;
        }
    }
    if (threadIdx.x == 0) {
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i] + b2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(int iters) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    unsigned long long* clk; CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8));
    std::vector<unsigned long long> hclk(2 * cus * 8);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int per_step = kOps[MODE];
    if (MODE == REAL_MIX || MODE == REAL_MIX2) iters /= 8;
    for (int bpc : {1, 2, 3, 4, 5, 8}) {       // blocks of 256 threads per CU = waves per SIMD
        const int grid = cus * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, clk, 50, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(hclk.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int g = 0; g < grid; ++g) if (hclk[2 * g + 1]) f.push_back((double)hclk[2 * g] / (double)hclk[2 * g + 1] * 100e6);
        std::sort(f.begin(), f.end());
        const double ghz = f.empty() ? 0 : f[f.size() / 2] * 1e-9;
        const double winstr = (double)grid * 4 /*waves*/ * iters * 4.0 * per_step;     // wave-instructions executed
        const double per_s = winstr / (ms * 1e-3);
        // cycles per wave-instruction per SIMD = clock / (wave-instr/s per SIMD)
        const double cyc = ghz * 1e9 / (per_s / (cus * 4.0));
        printf("%-13s waves/SIMD=%d  %8.3f ms  %.3e winstr/s  clk %.3f GHz  %.2f cyc/winstr/SIMD  (%.3e lane-ops/s)\n", kNames[MODE], bpc,
               ms, per_s, ghz, cyc, per_s * 64.0);
    }
    CK(hipFree(out)); CK(hipFree(clk));
    return 0;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  gcn=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
    if (run<XOR_BCNT>(iters)) return 1;
    if (run<BCNT>(iters)) return 1;
    if (run<XOR_E32>(iters)) return 1;
    if (run<XOR_E64>(iters)) return 1;
    if (run<LSHL_OR>(iters)) return 1;
    if (run<MIN3>(iters)) return 1;
    if (run<MIN_E32>(iters)) return 1;
    if (run<ADD3>(iters)) return 1;
    if (run<BFI>(iters)) return 1;
    if (run<ALIGNBIT>(iters)) return 1;
    if (run<CMP_CNDMASK>(iters)) return 1;
    if (run<CMP16_CNDMASK>(iters)) return 1;
    if (run<LSHLREV_OR>(iters)) return 1;
    if (run<REAL_MIX>(iters)) return 1;
    if (run<REAL_MIX2>(iters)) return 1;
    return 0;
}
