#!/usr/bin/env python3
"""Generates tools/order_bench.hip: pure-asm loop bodies (16 distances = 16 chains of 8 xor + 8 bcnt, then 16 packs and
8 min3) in different instruction orders, to find the order the gfx950 VALU issues fastest.  Temporaries live in
hard-coded VGPRs v40.. (clobbered), query words are 8 VGPR operands, train words 16 SGPR operands."""
import sys

NQ = 8  # query rows per body (all reuse the same 8 query VGPR operands: timing only)

def xor(c, k):     # chain c = j*2 + r ; SGPR operand index = r*8 + k -> %(2+..)
    r = c & 1
    return f"v_xor_b32_e32 v{60 + c}, %{2 + r * 8 + k}, %{18 + k}"
def bcnt(c, k):
    return f"v_bcnt_u32_b32 v{40 + c}, v{60 + c}, {'0' if k == 0 else 'v%d' % (40 + c)}"
def tail():
    out = []
    for c in range(16):
        out.append(f"v_lshl_or_b32 v{40 + c}, v{40 + c}, 22, %{26 + (c & 1)}")
    for j in range(8):
        out.append(f"v_min3_u32 v{80 + j}, v{80 + j}, v{40 + 2 * j}, v{41 + 2 * j}")
    return out

def seq(nop=False):
    o = []
    for c in range(16):
        for k in range(8):
            o.append(xor(c, k))
            if nop: o.append("s_nop 0")
            o.append(bcnt(c, k))
    return o
def grouped(n, nop=False):
    o = []
    for g in range(0, 16, n):
        for k in range(8):
            for c in range(g, g + n): o.append(xor(c, k))
            if nop: o.append("s_nop 0")
            for c in range(g, g + n): o.append(bcnt(c, k))
    return o
def alt(n):
    """n chains; classes alternate x,b,x,b; each bcnt consumes the xor issued n... slots earlier"""
    o = []
    for g in range(0, 16, n):
        pend = []
        for k in range(8):
            for c in range(g, g + n):
                o.append(xor(c, k))
                pend.append((c, k))
                if len(pend) >= n:
                    o.append(bcnt(*pend.pop(0)))
        for p in pend: o.append(bcnt(*p))
    return o
def skew1():
    o = []
    for c in range(16):
        o.append(xor(c, 0))
        for k in range(1, 8):
            o.append(xor(c, k)); o.append(bcnt(c, k - 1))
        o.append(bcnt(c, 7))
    return o
def fused_tail(n):
    """grouped(n) but each group's packs/min3 right after the group (as the kernel must do per asm statement)"""
    o = []
    for g in range(0, 16, n):
        for k in range(8):
            for c in range(g, g + n): o.append(xor(c, k))
            for c in range(g, g + n): o.append(bcnt(c, k))
        for c in range(g, g + n): o.append(f"v_lshl_or_b32 v{40 + c}, v{40 + c}, 22, %{26 + (c & 1)}")
        for c in range(g, g + n, 2): o.append(f"v_min3_u32 v{80 + c // 2}, v{80 + c // 2}, v{40 + c}, v{41 + c}")
    return o, True

def seqn(ax=None, ab=None, tail_nop=None, every=1):
    o = []
    n = 0
    for c in range(16):
        for k in range(8):
            o.append(xor(c, k)); n += 1
            if ax and n % every == 0: o.append(ax)
            o.append(bcnt(c, k))
            if ab: o.append(ab)
    t = []
    for c in range(16):
        t.append(f"v_lshl_or_b32 v{40 + c}, v{40 + c}, 22, %{26 + (c & 1)}")
        if tail_nop: t.append(tail_nop)
    for j in range(8):
        t.append(f"v_min3_u32 v{80 + j}, v{80 + j}, v{40 + 2 * j}, v{41 + 2 * j}")
        if tail_nop: t.append(tail_nop)
    return (o + t, True)
def xnb_skew(ax):
    """x_k+1 issued before b_k, nop after each xor"""
    o = []
    for c in range(16):
        o.append(xor(c, 0)); o.append(ax)
        for k in range(1, 8):
            o.append(xor(c, k)); o.append(ax); o.append(bcnt(c, k - 1))
        o.append(bcnt(c, 7))
    return o
VARIANTS = [
    # ---- part 1: instruction ORDER, no nops (profiles/r01_order_bench_orders.txt)
    ("V1 seq x b (fold2)", seq()),
    ("V3 2 chains xx bb", grouped(2)),
    ("V4 4 chains x4 b4", grouped(4)),
    ("V5 8 chains x8 b8", grouped(8)),
    ("V6 16 chains x16 b16", grouped(16)),
    ("V7 2 chains xx nop bb", grouped(2, True)),
    ("V8 1 chain skewed", skew1()),
    ("V9 alt 2 (x b' x b)", alt(2)),
    ("V11 alt 4", alt(4)),
    ("V12 alt 8", alt(8)),
    ("V13 2 chains + own tail", fused_tail(2)),
    ("V14 4 chains + own tail", fused_tail(4)),
    # ---- part 2: NOP placement on the sequential order (profiles/r01_order_bench_nops.txt)
    ("N0 x b", seqn()),
    ("N1 x nop0 b", seqn("s_nop 0")),
    ("N2 x nop1 b", seqn("s_nop 1")),
    ("N3 x nop0 b nop0", seqn("s_nop 0", "s_nop 0")),
    ("N4 x b nop0", seqn(None, "s_nop 0")),
    ("N5 x nop0 b +tail nops", seqn("s_nop 0", None, "s_nop 0")),
    ("N6 x s_mov b", seqn("s_mov_b32 s90, 0")),
    ("N7 x s_setprio b", seqn("s_setprio 0")),
    ("N8 x nop0 b nop0 +tail", seqn("s_nop 0", "s_nop 0", "s_nop 0")),
    ("N9 x v_nop b", seqn("v_nop")),
    ("N10 skew x nop x nop b", xnb_skew("s_nop 0")),
    ("N11 x s_sleep0 b", seqn("s_sleep 0")),
    ("N12 x nop2 b", seqn("s_nop 2")),
]
def grouped_prio(n, hi_b=True, nop=False):
    """n chains: n xors, s_setprio, n bcnts, s_setprio — the wave's priority differs between its xor and bcnt phases
    (hi_b: bcnt phase at priority 3, xor phase at 0; else the other way round)"""
    o = []
    pb, px = ("s_setprio 3", "s_setprio 0") if hi_b else ("s_setprio 0", "s_setprio 3")
    for g in range(0, 16, n):
        for k in range(8):
            for c in range(g, g + n): o.append(xor(c, k))
            o.append(pb)
            for i, c in enumerate(range(g, g + n)):
                o.append(bcnt(c, k))
                if nop and i + 1 < n: o.append("s_nop 0")
            o.append(px)
    return o
def seq_alt_prio():
    """x, s_setprio, b with the priority alternating pair by pair (one SOPP per pair, as the s_nop rule)"""
    o = []
    n = 0
    for c in range(16):
        for k in range(8):
            o.append(xor(c, k)); o.append("s_setprio 3" if n & 1 else "s_setprio 0"); o.append(bcnt(c, k)); n += 1
    return o
VARIANTS += [
    # ---- part 3 (round 3): does the wave PRIORITY steer which waves' xors share a quad-cycle?  (SQ_ACTIVE_INST_VALU2 shows
    #      two xors of DIFFERENT waves sharing 25 % of the product kernel's quad-cycles; a wave alone never pairs its own)
    ("P1 x prio3 b prio0", seqn("s_setprio 3", "s_setprio 0")),
    ("P2 x prio0 b prio3", seqn("s_setprio 0", "s_setprio 3")),
    ("P3 xx prio3 bb prio0", grouped_prio(2)),
    ("P4 x4 prio3 b4 prio0", grouped_prio(4)),
    ("P5 x8 prio3 b8 prio0", grouped_prio(8)),
    ("P6 xx prio0 bb prio3", grouped_prio(2, False)),
    ("P7 x4 prio0 b4 prio3", grouped_prio(4, False)),
    ("P8 xx prio3 b nop b prio0", grouped_prio(2, True, True)),
    ("P9 x4 prio3 b nop.. prio0", grouped_prio(4, True, True)),
    ("P10 x prio(alt) b", seq_alt_prio()),
    ("P11 x nop0 b, odd blocks prio3", seqn("s_nop 0")),
    ("P12 x b, odd blocks prio3", seqn()),
]
def grouped_prio_lv(n, hi, lo, tail_hi=False, own_tail=True):
    """grouped_prio with chosen priority levels; tail_hi: the 4-cycle pack / min ops of the tail run at the bcnt priority"""
    o = []
    for g in range(0, 16, n):
        for k in range(8):
            for c in range(g, g + n): o.append(xor(c, k))
            o.append(f"s_setprio {hi}")
            for c in range(g, g + n): o.append(bcnt(c, k))
            if not (tail_hi and k == 7): o.append(f"s_setprio {lo}")
        if tail_hi:
            for c in range(g, g + n): o.append(f"v_lshl_or_b32 v{40 + c}, v{40 + c}, 22, %{26 + (c & 1)}")
            for c in range(g, g + n, 2): o.append(f"v_min3_u32 v{80 + c // 2}, v{80 + c // 2}, v{40 + c}, v{41 + c}")
            o.append(f"s_setprio {lo}")
    return (o, True) if tail_hi else o
def no_tail(ins):
    return (ins, True)
VARIANTS += [
    # ---- part 4 (round 3): where is the floor?  (8 x 2 + 8 x 4 = 48 cycles if every xor shares its quad-cycle)
    ("Q1 xx p3 bb p0, NO tail", no_tail(grouped_prio(2))),
    ("Q2 x nop0 b, NO tail", no_tail(seqn("s_nop 0")[0][:16 * 8 * 3])),
    ("Q3 xx p1 bb p0", grouped_prio_lv(2, 1, 0)),
    ("Q4 xx p2 bb p1", grouped_prio_lv(2, 2, 1)),
    ("Q5 xx p3 bb+tail p0", grouped_prio_lv(2, 3, 0, True)),
    ("Q6 x4 p3 b4+tail p0", grouped_prio_lv(4, 3, 0, True)),
    ("Q7 x8 p3 b8+tail p0", grouped_prio_lv(8, 3, 0, True)),
    ("Q8 x16 p3 b16 p0", grouped_prio(16)),
]
STATIC_PRIO = {"P11 x nop0 b, odd blocks prio3", "P12 x b, odd blocks prio3"}
def body(v):
    name, ins = v
    own_tail = False
    if isinstance(ins, tuple): ins, own_tail = ins
    if not own_tail: ins = ins + tail()
    return ins

src = ['// GENERATED by tools/gen_order_bench.py — do not edit.',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>', '#include <cstdlib>', '#include <vector>', '#include <algorithm>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)',
       'template <int V> __global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t seed) {',
       '  constexpr bool STATIC_PRIO_V = ' + " || ".join(f"V == {i}" for i, v in enumerate(VARIANTS) if v[0] in STATIC_PRIO) + ';',
       '  uint32_t q[8];',
       '  for (int i = 0; i < 8; ++i) q[i] = threadIdx.x * 2654435761u + i * 40503u + seed;',
       '  uint32_t s = seed, r0v = 0, r1v = 0;',
       '  asm volatile("' + "\\n\\t".join(f"v_mov_b32 v{80 + j}, -1" for j in range(8)) + '" ::: ' + ", ".join(f'"v{80 + j}"' for j in range(8)) + ');',
       '  if (STATIC_PRIO_V && (blockIdx.x & 1)) asm volatile("s_setprio 3");',
       '  unsigned long long t0 = 0, r0 = 0;',
       '  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }',
       '  for (int it = 0; it < iters; ++it) {',
       '    uint32_t A[16];',
       '    for (int i = 0; i < 16; ++i) { A[i] = s; s = s * 1664525u + 1013904223u; }',
       '    const uint32_t t = (uint32_t)it * 2, t1 = t + 1;']
clob = ", ".join(f'"v{r}"' for r in list(range(40, 56)) + list(range(60, 76)) + list(range(80, 88))) + ', "s90"'
for vi, v in enumerate(VARIANTS):
    ins = body(v)
    txt = "\\n\\t".join(ins)
    src.append(f'    if (V == {vi}) asm volatile("{txt}" : "+v"(r0v), "+v"(r1v) : ' +
               ", ".join(f'"s"(A[{i}])' for i in range(16)) + ", " + ", ".join(f'"v"(q[{i}])' for i in range(8)) +
               f', "s"(t), "s"(t1) : {clob});')
src += ['  }',
        '  if (threadIdx.x == 0) { unsigned long long t1c = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); clk[blockIdx.x * 2] = t1c - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }',
        '  uint32_t acc = r0v + r1v;',
        '  asm volatile("' + "\\n\\t".join(f"v_add_u32 %0, %0, v{80 + j}" for j in range(8)) + '" : "+v"(acc) :: ' + ", ".join(f'"v{80 + j}"' for j in range(8)) + ');',
        '  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;',
        '}',
        'template <int V> int run(const char* name, int iters) {',
        '  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;',
        '  uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));',
        '  unsigned long long* clk; CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8)); std::vector<unsigned long long> hclk(2 * cus * 8);',
        '  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));',
        '  printf("%-26s", name);',
        '  for (int bpc : {2, 3, 4, 5, 6, 7, 8}) {',
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
        '  printf("SIMD-cycles per 64 distances (and chip-wide Tdist/s) at w waves per SIMD\\n");']
src.append('  const int first = argc > 2 ? atoi(argv[2]) : 0;')
for vi, v in enumerate(VARIANTS):
    src.append(f'  if ({vi} >= first && run<{vi}>("{v[0]}", iters)) return 1;')
src += ['  return 0;', '}']
open(sys.argv[1] if len(sys.argv) > 1 else "tools/order_bench.hip", "w").write("\n".join(src) + "\n")
