#!/usr/bin/env python3
"""Generates tools/prio_bench.hip (round 3): pure-asm loop bodies of the product's inner loop — 16 chains (8 query rows x 2
stored rows) of 8 v_xor_b32 + 8 v_bcnt_u32_b32, one v_min3_u32 per two chains — in which the WAVE PRIORITY differs
between a wave's xor phase and its bcnt phase (s_setprio).  Found with tools/order_bench (part 3): with the bcnt phase
at the higher priority the SIMD keeps its 4-cycle pipe fed from the high-priority waves and issues the other waves'
2-cycle xors beside them; SQ_ACTIVE_INST_VALU2 counts those shared quad-cycles.

Registers are hard-coded low (v24..v63) so that 8 workgroups of 256 threads fit a CU: `w` in the output really is the
number of waves per SIMD (the kernel asserts its occupancy).  Variant language: a body is built per GROUP of n chains;
for each of the 8 words: n xors, [hi], n bcnts, [lo]; then the group's min3 ops (before or after the last [lo])."""
import sys

D0, X0, B0 = 24, 40, 56          # distance accumulators (16), xor temporaries (<= 16), running minima (8)

def xor(c, k, slot):   # chain c = 2 j + r : query row j (all reuse the 8 query VGPR operands: timing only), stored row r
    r = c & 1
    return f"v_xor_b32_e32 v{X0 + slot}, %{2 + r * 8 + k}, %{18 + k}"
def bcnt(c, k, slot):
    return f"v_bcnt_u32_b32 v{D0 + c}, v{X0 + slot}, {'0' if k == 0 else 'v%d' % (D0 + c)}"
def min3(c):
    return f"v_min3_u32 v{B0 + c // 2}, v{B0 + c // 2}, v{D0 + c}, v{D0 + c + 1}"

def grouped(n, hi="s_setprio 3", lo="s_setprio 0", tail="hi", mid=None, tail_ops=True):
    """tail: 'hi' = the group's min3 ops run before the last lo (at the bcnt priority), 'lo' = after it"""
    o = []
    for g in range(0, 16, n):
        for k in range(8):
            for i, c in enumerate(range(g, g + n)): o.append(xor(c, k, i))
            if hi: o.append(hi)
            for i, c in enumerate(range(g, g + n)):
                o.append(bcnt(c, k, i))
                if mid and i + 1 < n: o.append(mid)
            last = (k == 7)
            if last and tail_ops and tail == "hi":
                for c in range(g, g + n, 2): o.append(min3(c))
            if lo: o.append(lo)
            if last and tail_ops and tail == "lo":
                for c in range(g, g + n, 2): o.append(min3(c))
    return o

def grouped_min16(n, where="lo"):
    """the running minimum as one 2-cycle v_min_u16 per distance (a distance is <= 256) instead of half a 4-cycle v_min3_u32"""
    o = []
    for g in range(0, 16, n):
        for k in range(8):
            for i, c in enumerate(range(g, g + n)): o.append(xor(c, k, i))
            o.append("s_setprio 3")
            for i, c in enumerate(range(g, g + n)): o.append(bcnt(c, k, i))
            m = [f"v_min_u16_e32 v{B0 + c // 2}, v{B0 + c // 2}, v{D0 + c}" for c in range(g, g + n)]
            if k == 7 and where == "hi": o += m
            o.append("s_setprio 0")
            if k == 7 and where == "lo": o += m
    return o
def pipelined(n, hi="s_setprio 3", lo="s_setprio 0", tail_ops=True):
    """software-pipelined by one word: the xors of word k + 1 are issued BEFORE the bcnts of word k (two sets of temporaries),
    so a bcnt never waits for an xor issued just before it"""
    o = []
    for g in range(0, 16, n):
        for i, c in enumerate(range(g, g + n)): o.append(xor(c, 0, i))
        for k in range(8):
            cur, nxt = (k & 1) * n, ((k + 1) & 1) * n
            if k + 1 < 8:
                for i, c in enumerate(range(g, g + n)): o.append(xor(c, k + 1, nxt + i))
            if hi: o.append(hi)
            for i, c in enumerate(range(g, g + n)): o.append(bcnt(c, k, cur + i))
            if k == 7 and tail_ops:
                for c in range(g, g + n, 2): o.append(min3(c))
            if lo: o.append(lo)
    return o

def seq_nop(tail_ops=True):
    """round 1/2's product order: (x, s_nop 0, b) per word, chain after chain, min3 per two chains"""
    o = []
    for c in range(16):
        for k in range(8):
            o += [xor(c, k, 0), "s_nop 0", bcnt(c, k, 0)]
        if tail_ops and (c & 1): o.append(min3(c - 1))
    return o

VARIANTS = [
    ("A0 x nop b (round 2)", seq_nop()),
    ("A1 xx p3 bb p0, min3 lo", grouped(2, tail="lo")),
    ("A2 xx p3 bb p0, min3 hi", grouped(2, tail="hi")),
    ("A3 x4 p3 b4 p0, min3 hi", grouped(4)),
    ("A4 x8 p3 b8 p0, min3 hi", grouped(8)),
    ("A5 x16 p3 b16 p0", grouped(16)),
    ("A6 xx p3 bb p0 pipelined", pipelined(2)),
    ("A7 x4 p3 b4 p0 pipelined", pipelined(4)),
    ("A8 x8 p3 b8 p0 pipelined", pipelined(8)),
    ("A9 xx p1 bb p0", grouped(2, "s_setprio 1", "s_setprio 0")),
    ("A10 xx p3 bb (p0 only at end)", grouped(2, "s_setprio 3", None) + ["s_setprio 0"]),
    ("A11 xx nop bb (no prio)", grouped(2, "s_nop 0", None)),
    ("A12 x4 p3 b4 nop-sep p0", grouped(4, mid="s_nop 0")),
    ("A13 xx p3 bb p0 + min_u16 lo", grouped_min16(2)),
    ("A14 xx p3 bb + min_u16 hi p0", grouped_min16(2, "hi")),
    ("A15 x4 p3 b4 p0 + min_u16 lo", grouped_min16(4)),
    # without the min3 ops: the pure mix
    ("B0 x nop b, pure", seq_nop(False)),
    ("B1 xx p3 bb p0, pure", grouped(2, tail_ops=False)),
    ("B2 x4 p3 b4 p0, pure", grouped(4, tail_ops=False)),
    ("B3 x8 p3 b8 p0, pure", grouped(8, tail_ops=False)),
    ("B4 xx pipelined, pure", pipelined(2, tail_ops=False)),
    ("B5 x4 pipelined, pure", pipelined(4, tail_ops=False)),
    ("B6 x8 pipelined, pure", pipelined(8, tail_ops=False)),
]

clob_regs = list(range(D0, D0 + 16)) + list(range(X0, X0 + 16)) + list(range(B0, B0 + 8))
clob = ", ".join(f'"v{r}"' for r in clob_regs)
src = ['// GENERATED by tools/gen_prio_bench.py — do not edit.',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>', '#include <cstdlib>', '#include <vector>', '#include <algorithm>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)',
       'template <int V> __global__ __launch_bounds__(256, 2) void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t seed) {',
       '  uint32_t q[8];',
       '  for (int i = 0; i < 8; ++i) q[i] = threadIdx.x * 2654435761u + i * 40503u + seed;',
       '  uint32_t s = seed, r0v = 0, r1v = 0;',
       '  asm volatile("' + "\\n\\t".join(f"v_mov_b32 v{B0 + j}, -1" for j in range(8)) + '" ::: ' + ", ".join(f'"v{B0 + j}"' for j in range(8)) + ');',
       '  unsigned long long t0 = 0, r0 = 0;',
       '  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }',
       '  for (int it = 0; it < iters; ++it) {',
       '    uint32_t A[16];',
       '    for (int i = 0; i < 16; ++i) { A[i] = s; s = s * 1664525u + 1013904223u; }',
       '    const uint32_t t = (uint32_t)it * 2, t1 = t + 1;']
for vi, (name, ins) in enumerate(VARIANTS):
    txt = "\\n\\t".join(ins)
    src.append(f'    if (V == {vi}) asm volatile("{txt}" : "+v"(r0v), "+v"(r1v) : ' +
               ", ".join(f'"s"(A[{i}])' for i in range(16)) + ", " + ", ".join(f'"v"(q[{i}])' for i in range(8)) +
               f', "s"(t), "s"(t1) : {clob});')
src += ['  }',
        '  if (threadIdx.x == 0) { unsigned long long t1c = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); clk[blockIdx.x * 2] = t1c - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }',
        '  uint32_t acc = r0v + r1v;',
        '  asm volatile("' + "\\n\\t".join(f"v_add_u32 %0, %0, v{B0 + j}" for j in range(8)) + '" : "+v"(acc) :: ' + ", ".join(f'"v{B0 + j}"' for j in range(8)) + ');',
        '  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;',
        '}',
        'template <int V> int run(const char* name, int iters) {',
        '  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;',
        '  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<V>, 256, 0));',
        '  uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));',
        '  unsigned long long* clk; CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8)); std::vector<unsigned long long> hclk(2 * cus * 8);',
        '  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));',
        '  printf("%-30s occ %d ", name, occ);',
        '  for (int bpc : {2, 4, 5, 6, 7, 8}) {',
        '    if (bpc > occ) { printf("  w%d: -", bpc); continue; }',
        '    const int grid = cus * bpc;',
        '    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, clk, 50, 1u); CK(hipDeviceSynchronize());',
        '    CK(hipEventRecord(a)); hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 7u); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));',
        '    float ms; CK(hipEventElapsedTime(&ms, a, b));',
        '    CK(hipMemcpy(hclk.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));',
        '    std::vector<double> f; for (int g = 0; g < grid; ++g) if (hclk[2 * g + 1]) f.push_back((double)hclk[2 * g] / (double)hclk[2 * g + 1] * 100e6);',
        '    std::sort(f.begin(), f.end()); const double ghz = f.empty() ? 0 : f[f.size() / 2] * 1e-9;',
        '    const double per_s = (double)grid * 4 * iters * 16.0 / (ms * 1e-3);',
        '    printf("  w%d: %5.1f cyc %.3fT/s", bpc, ghz * 1e9 / (per_s / (cus * 4.0)), per_s * 64.0 * 1e-12);',
        '  }',
        '  printf("\\n"); CK(hipFree(out)); CK(hipFree(clk)); return 0;',
        '}',
        'int main(int argc, char** argv) {',
        '  int iters = argc > 1 ? atoi(argv[1]) : 50000;',
        '  const int first = argc > 2 ? atoi(argv[2]) : 0, last = argc > 3 ? atoi(argv[3]) : 9999;',
        '  printf("SIMD-cycles per 64 distances (and chip-wide Tdist/s) at w waves per SIMD\\n");']
for vi, (name, _) in enumerate(VARIANTS):
    src.append(f'  if ({vi} >= first && {vi} <= last && run<{vi}>("{name}", iters)) return 1;')
src += ['  return 0;', '}']
open(sys.argv[1] if len(sys.argv) > 1 else "tools/prio_bench.hip", "w").write("\n".join(src) + "\n")
