// valu_busy.hip — three ALU-only loops at 8 waves per SIMD, one kernel each, meant to be run under
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
// so that the product kernel's VALU-busy counters can be read against known reference points:
//   k_pure_xor   only v_xor_b32            (the 2-cycle class of profiles/r01_valu_class.txt)
//   k_pure_bcnt  only v_bcnt_u32_b32       (the 4-cycle class)
//   k_pair_mix   rounds 1-2's inner-loop body: 8 x (v_xor_b32 ; s_nop 0 ; v_bcnt_u32_b32) per distance + v_min3_u32
//                per two distances, operands in registers / SGPRs (constant "train rows"), nothing else in the loop
//   k_prio_mix   round 3's body: the same instructions, two chains per phase, popcounts + min3 at s_setprio 3, xors at 0
// Prints each kernel's wall time and SIMD cycles per wave64 VALU instruction at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_pure_xor(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = a[i] * 3 + 1; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_pure_bcnt(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = a[i] * 3 + 1; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the product's fold2_min (lcm_kernels.hip): one query row (8 VGPRs) against two train rows (16 SGPRs)
__device__ __forceinline__ void fold2_min(uint32_t& best, const uint32_t (&q)[8], const uint32_t (&s)[16]) {
    uint32_t d0, d1, x;
    asm volatile(
        "v_xor_b32_e32 %3, %4, %20\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, 0\n\t"
        "v_xor_b32_e32 %3, %5, %21\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %6, %22\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %7, %23\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %8, %24\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %9, %25\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %10, %26\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %11, %27\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %12, %20\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, 0\n\t"
        "v_xor_b32_e32 %3, %13, %21\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %14, %22\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %15, %23\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %16, %24\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %17, %25\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %18, %26\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %19, %27\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_min3_u32 %0, %0, %1, %2"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x)
        : "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]),
          "s"(s[8]), "s"(s[9]), "s"(s[10]), "s"(s[11]), "s"(s[12]), "s"(s[13]), "s"(s[14]), "s"(s[15]),
          "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
}

// round 3's order of the same work (lcm_kernels.hip, LCM_INNER_PRIO): two xors at priority 0, the two popcounts and the
// minimum update at priority 3 — the other waves' half-rate xors issue beside this wave's quarter-rate popcounts
__device__ __forceinline__ void fold2_min_prio(uint32_t& best, const uint32_t (&q)[8], const uint32_t (&s)[16]) {
    uint32_t d0, d1, x0, x1;
    asm volatile(
        "v_xor_b32_e32 %3, %5, %21\n\tv_xor_b32_e32 %4, %13, %21\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, 0\n\tv_bcnt_u32_b32 %2, %4, 0\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %6, %22\n\tv_xor_b32_e32 %4, %14, %22\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %7, %23\n\tv_xor_b32_e32 %4, %15, %23\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %8, %24\n\tv_xor_b32_e32 %4, %16, %24\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %9, %25\n\tv_xor_b32_e32 %4, %17, %25\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %10, %26\n\tv_xor_b32_e32 %4, %18, %26\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %11, %27\n\tv_xor_b32_e32 %4, %19, %27\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %12, %28\n\tv_xor_b32_e32 %4, %20, %28\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\t"
        "v_min3_u32 %0, %0, %1, %2\n\ts_setprio 0"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x0), "=&v"(x1)
        : "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]),
          "s"(s[8]), "s"(s[9]), "s"(s[10]), "s"(s[11]), "s"(s[12]), "s"(s[13]), "s"(s[14]), "s"(s[15]),
          "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
}

template <bool PRIO>
__device__ __forceinline__ void mix_body(uint32_t* out, int iters, uint32_t seed) {
    uint32_t q[8][8], best[8], s[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        best[j] = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < 8; ++w) q[j][w] = (threadIdx.x + 256 * j) * 2654435761u + w * 40503u + seed;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = __builtin_amdgcn_readfirstlane(seed * (k + 3) + blockIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (PRIO) fold2_min_prio(best[j], q[j], s); else fold2_min(best[j], q[j], s); }
    }
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += best[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256, 6) void k_pair_mix(uint32_t* out, int iters, uint32_t seed) { mix_body<false>(out, iters, seed); }
__global__ __launch_bounds__(256, 6) void k_prio_mix(uint32_t* out, int iters, uint32_t seed) { mix_body<true>(out, iters, seed); }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 0.05;      // rough duration of each kernel
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out = nullptr;
    CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct { const char* name; void (*fn)(uint32_t*, int, uint32_t); int bpc; double winstr_per_iter; double cyc_guess; } ks[] = {
        {"k_pure_xor  (64 v_xor_b32 per iteration)", k_pure_xor, 8, 64.0, 2.1},
        {"k_pure_bcnt (64 v_bcnt_u32_b32 per iteration)", k_pure_bcnt, 8, 64.0, 4.2},
        {"k_pair_mix  (8 rows x [16 xor + 16 bcnt + 1 min3] per iteration)", k_pair_mix, 6, 8 * 33.0, 3.3},
        {"k_prio_mix  (the same work, popcounts at s_setprio 3, xors at 0)", k_prio_mix, 6, 8 * 33.0, 2.3},
    };
    for (auto& k : ks) {
        const int grid = cus * k.bpc;       // 256-thread workgroups: bpc per CU = bpc waves per SIMD
        const int iters = (int)(seconds * 2.4e9 / (k.winstr_per_iter * k.bpc * k.cyc_guess));
        hipLaunchKernelGGL(k.fn, dim3(grid), dim3(256), 0, 0, out, 100, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k.fn, dim3(grid), dim3(256), 0, 0, out, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-68s %d waves/SIMD  %8.3f ms  %5.2f SIMD-cycles per wave64 VALU instruction (at 2.4 GHz)\n", k.name, k.bpc, ms,
               ms * 1e-3 * 2.4e9 / (k.winstr_per_iter * k.bpc * iters));
    }
    return 0;
}
