// fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the loop-search kernel.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B reads exactly HALF the bytes of a wide coalesced 16 B/lane
// stream, and "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// Each kernel reads a 1 GiB buffer exactly once (far beyond the 32 MiB of L2), so bytes read = 2^30:
//   k_wide16      lane l of a wave reads the 16 bytes at 16 l of consecutive 1 KiB blocks        (the guide's reference: expect 1/2)
//   k_rowpair     lane l reads the 32-byte row l of consecutive 2 KiB blocks as two 16-byte loads (how query rows are loaded)
//   k_scalar64    every wave reads consecutive 64-byte lines with s_load_dwordx16                 (how stored rows are streamed)
//   k_gather32    lane l reads one 32-byte row at a pseudo-random row index, two 16-byte loads    (the argmin re-scan)
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./tools/fetch_calib     then tools/fetch_calib_summary.py
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef const uint32_t __attribute__((address_space(4))) * sptr_t;

__global__ __launch_bounds__(256) void k_wide16(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_rowpair(const uint4* __restrict__ p, size_t n_rows, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (size_t)gridDim.x * 256) {
        const uint4 lo = p[r * 2], hi = p[r * 2 + 1];
        acc += lo.x ^ lo.w ^ hi.y ^ hi.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_scalar64(const uint32_t* p, size_t n_lines, uint32_t* out) {
    // one 64-byte line per wave per trip, wave-uniform address -> s_load_dwordx16
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * 256) >> 6;
    uint32_t acc = 0;
    for (size_t l = __builtin_amdgcn_readfirstlane((uint32_t)wave); l < n_lines; l += n_waves) {
        sptr_t s = (sptr_t)(p + l * 16);
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = s[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather32(const uint4* __restrict__ p, size_t n_rows, uint32_t* out) {
    // every row read exactly once, in a scrambled order: row = (i * odd) mod 2^k  (n_rows is a power of two)
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (size_t)gridDim.x * 256) {
        const size_t r = (i * 2654435761ull) & (n_rows - 1);
        const uint4 lo = p[r * 2], hi = p[r * 2 + 1];
        acc += lo.x ^ lo.w ^ hi.y ^ hi.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t* buf = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0x5A, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k_wide16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_rowpair, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 32, out);
    hipLaunchKernelGGL(k_scalar64, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 64, out);
    hipLaunchKernelGGL(k_gather32, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 32, out);
    CK(hipDeviceSynchronize());
    printf("each kernel read %zu bytes once\n", bytes);
    return 0;
}
