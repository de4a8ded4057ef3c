// valu_class.hip — issue cost (SIMD cycles per wave64 instruction, 8 waves/SIMD, independent registers) of candidate
// integer VALU instructions on gfx950: which are "full rate" (2 cycles) and which "half rate" (4 cycles)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPS(X) \
  X(0, "v_xor_b32", "v_xor_b32_e32 %0, %1, %0") \
  X(1, "v_and_b32", "v_and_b32_e32 %0, %1, %0") \
  X(2, "v_or_b32", "v_or_b32_e32 %0, %1, %0") \
  X(3, "v_add_u32", "v_add_u32_e32 %0, %1, %0") \
  X(4, "v_sub_u32", "v_sub_u32_e32 %0, %1, %0") \
  X(5, "v_mov_b32", "v_mov_b32_e32 %0, %1") \
  X(6, "v_min_u32", "v_min_u32_e32 %0, %1, %0") \
  X(7, "v_max_u32", "v_max_u32_e32 %0, %1, %0") \
  X(8, "v_min_u16", "v_min_u16_e32 %0, %1, %0") \
  X(9, "v_lshlrev_b32", "v_lshlrev_b32_e32 %0, 3, %0") \
  X(10, "v_lshrrev_b32", "v_lshrrev_b32_e32 %0, 3, %0") \
  X(11, "v_cndmask_b32", "v_cndmask_b32_e32 %0, %0, %1, vcc") \
  X(12, "v_cmp_lt_u32", "v_cmp_lt_u32_e32 vcc, %1, %0") \
  X(13, "v_cmp_lt_u16", "v_cmp_lt_u16_e32 vcc, %1, %0") \
  X(14, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %1, %0") \
  X(15, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1") \
  X(16, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2") \
  X(17, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2") \
  X(18, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2") \
  X(19, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1") \
  X(20, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 1") \
  X(21, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
  X(22, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
  X(23, "v_sad_u32", "v_sad_u32 %0, %0, %1, %2") \
  X(24, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
  X(25, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2") \
  X(26, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
  X(27, "v_add_f32", "v_add_f32_e32 %0, %1, %0") \
  X(28, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
  X(29, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
  X(30, "v_mul_u32_u24", "v_mul_u32_u24_e32 %0, %1, %0") \
  X(31, "v_xnor_b32", "v_xnor_b32_e32 %0, %1, %0") \
  X(32, "v_not_b32", "v_not_b32_e32 %0, %1") \
  X(33, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 9") \
  X(34, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2") \
  X(35, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %1, %2") \
  X(36, "v_dot8_u32_u4", "v_dot8_u32_u4 %0, %0, %1, %2") \
  X(37, "v_max3_u32", "v_max3_u32 %0, %0, %1, %2") \
  X(38, "v_med3_u32", "v_med3_u32 %0, %0, %1, %2") \
  X(39, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 9")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16], b[16], c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = a[i] * 3 + 1; c[i] = ~a[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#define X(id, name, txt) if (OP == id) asm volatile(txt : "+v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
            OPS(X)
#undef X
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int iters) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    static uint32_t* out = nullptr;
    if (!out) CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%-16s", name);
    for (int bpc : {1, 2, 4, 8}) {
        const int grid = cus * bpc;
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, out, 100, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, out, iters, 7u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double winstr_per_simd = (double)bpc * iters * 64.0;
        printf("  w=%d: %5.2f cyc", bpc, ms * 1e-3 * 2.4e9 / winstr_per_simd);
    }
    printf("   (cycles @2.4 GHz nominal)\n");
    return 0;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 100000;
#define X(id, name, txt) if (run<id>(name, iters)) return 1;
    OPS(X)
#undef X
    return 0;
}
