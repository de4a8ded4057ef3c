// lcm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the loop-closure matcher.
//
// What is computed (reference: LoopClosingSystem::matchFeatures / detectLoops, include/loop_closing.hpp:40,48;
// rules README.md:116-126): for one query frame and one stored ("train") frame, every query row's FIRST minimum
// 256-bit Hamming distance over the train rows (cv::BFMatcher NORM_HAMMING k=1 semantics), then the pair's
// min-of-mins, the count of matches with d <= max(ratio*min, floor), and one 8-byte lcm_score record.
//
// This is integer VALU work: 8 x v_xor_b32 + 8 x v_bcnt_u32_b32 (accumulating) per distance, no MFMA.
//
// k_score_rowlane — "row-per-lane, train rows through the scalar unit" (variants 0 / 1, the default):
//   * every lane OWNS up to QPT query rows in VGPRs (8 dwords each), loaded once per work item with coalesced
//     16-byte loads (row-major 32-byte rows: lane l reads row l — the 32-byte descriptor database layout);
//   * the train rows are wave-uniform: they are read with s_load_dwordx16 from the constant address space into
//     SGPRs and used directly as the scalar operand of v_xor_b32 — no LDS traffic, no VGPRs, no cross-lane work
//     in the inner loop;
//   * per-query running minimum DISTANCE folded with v_min3_u32; the throughput modes carry no train index in the
//     inner loop.  The first train row attaining the minimum (cv::BFMatcher's trainIdx: strict-'<' ascending scan of OpenCV's
//     batchDistance) comes from 8-row group keys kept in lane-private LDS words + a re-scan of the winning group
//     (ARGMIN_MODE 1, the bulk search), or from a packed key dist << 22 | train_idx per distance (ARGMIN_MODE 2, the
//     pair mode, where a launch is too small to hide the re-scan's dependent loads);
//   * per pair, the min-of-mins and the good-match count are LDS-atomic reductions (ds_min_u32 / ds_add_u32) over
//     the workgroup's lanes.
//
// k_score_trainlane — "train-row-per-lane, queries staged in LDS" (the mapping BASELINE.json's north_star sketches):
//   lanes own train rows (coalesced loads), query rows are broadcast from LDS, per-query min/argmin is a
//   wavefront __shfl reduction per query.  Kept for A/B measurement; see DESIGN.md for the numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcm_kernels.h"

#ifndef LCM_INNER_PRIO
#define LCM_INNER_PRIO 1        // 0: round 1/2's (v_xor, s_nop 0, v_bcnt) order, for A/B builds (`make nop` in this directory)
#endif

namespace lcm {

typedef const uint32_t __attribute__((address_space(4))) * sptr_t;   // constant AS => SMEM (s_load) when uniform
typedef const int32_t __attribute__((address_space(4))) * siptr_t;

// How the instruction stream is shaped for the gfx950 VALU (tools/order_bench, tools/prio_bench; profiles/r01_valu_class.txt,
// r01_order_bench_*.txt, r03_prio_bench.txt, r03_valu_issue.json):
//   * v_xor_b32 is a half-rate ("2-cycle") wave64 instruction, v_bcnt_u32_b32 / v_min3_u32 / v_lshl_or_b32 are quarter-rate
//     ("4-cycle") ones.  A SIMD issues at most one quarter-rate instruction per quad-cycle, and it can issue a half-rate
//     instruction of ANOTHER wave in the same quad-cycle (counter SQ_ACTIVE_INST_VALU2); one wave alone never pairs its own.
//   * Which waves' instructions meet is the arbiter's choice, and left alone it chooses badly: every bare order of the
//     mix costs 4.19 cycles per instruction (73 SIMD-cycles per 64 distances); rounds 1-2 put an `s_nop 0` between xor and
//     bcnt, which let a quarter of the quad-cycles carry two instructions (54 cycles per 64 distances).
//   * Round 3: the arbiter serves waves by PRIORITY.  Each wave raises its priority (s_setprio 3) for its popcounts and
//     the running-minimum update, and drops it (s_setprio 0) for its xors: the quarter-rate pipe is then fed every
//     quad-cycle from whichever waves have popcounts ready, and the other waves' xors ride in the same quad-cycles.  The
//     8 xors per distance all but vanish from the cost: 37 SIMD-cycles per 64 distances against the 35 of the quarter-
//     rate instructions alone (8 bcnt + 0.5 min3 + 0.27 bookkeeping, 4 cycles each) — +47 % on the whole search.
//     Everything that is not an xor must sit in the high-priority phase: the v_min3 issued after the s_setprio 0 cost 32 %.
//   * Two chains (the same query row against two stored rows) per phase, two temporaries; longer phases (4, 8, 16
//     chains) and software-pipelined xors measure the same or worse.
// One query row (8 VGPRs) against TWO train rows (16 SGPRs, rows t and t+1): both distances, both packed keys and the
// fold into the running minimum, as ONE asm statement so that the instruction order is exactly the one below.
__device__ __forceinline__ void fold2(uint32_t& best, const uint32_t (&q)[8], const uint32_t* s, uint32_t t0, uint32_t t1) {
#if LCM_INNER_PRIO
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
        "v_lshl_or_b32 %1, %1, 22, %29\n\t"
        "v_lshl_or_b32 %2, %2, 22, %30\n\t"
        "v_min3_u32 %0, %0, %1, %2\n\ts_setprio 0"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x0), "=&v"(x1)
        : "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]),
          "s"(s[8]), "s"(s[9]), "s"(s[10]), "s"(s[11]), "s"(s[12]), "s"(s[13]), "s"(s[14]), "s"(s[15]),
          "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "s"(t0), "s"(t1));
#else
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
        "v_lshl_or_b32 %1, %1, 22, %28\n\t"
        "v_lshl_or_b32 %2, %2, 22, %29\n\t"
        "v_min3_u32 %0, %0, %1, %2"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x)
        : "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]),
          "s"(s[8]), "s"(s[9]), "s"(s[10]), "s"(s[11]), "s"(s[12]), "s"(s[13]), "s"(s[14]), "s"(s[15]),
          "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "s"(t0), "s"(t1));
#endif
}

// Same, but only the minimum DISTANCE is tracked (no train index): what the loop search needs — a LoopCandidate
// (include/loop_closing.hpp:22-27) carries a match count and a similarity, never a train index, and the match list
// of a detected loop comes from the pair-mode kernel on demand (README.md:101 "Re-match features on identified loop
// frames").  Saves the two v_lshl_or_b32 per pair of distances.
__device__ __forceinline__ void fold2_min(uint32_t& best, const uint32_t (&q)[8], const uint32_t* s) {
#if LCM_INNER_PRIO
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
#else
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
#endif
}

// Re-scan step (ARGMIN kernels): the distance between this lane's query row and ONE train row that the lane picks
// itself (byte offset `off` from the wave-uniform frame base).  The two loads and their wait are one asm statement
// (8 temporary VGPRs, nothing for the compiler to hoist or interleave across rows: its own version of this loop wanted
// 157 VGPRs and spilled the query rows of the main loop); the 8 xor / popcount-accumulate pairs are plain C++.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t rescan_distance(const uint32_t (&q)[8], const uint32_t* base, uint32_t off) {
    // the whole 32-byte row in ONE round trip (two 16-byte loads, one wait): the re-scan is a chain of dependent loads,
    // and with short stored frames (a few hundred rows) its latency — not its arithmetic — is what the argmin costs
    u32x4 lo, hi;
    asm volatile(
        "global_load_dwordx4 %0, %2, %3\n\t"
        "global_load_dwordx4 %1, %2, %3 offset:16\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(lo), "=&v"(hi)
        : "v"(off), "s"(base)
        : "memory");
    uint32_t d = __popc(q[0] ^ lo.x);
    d += __popc(q[1] ^ lo.y); d += __popc(q[2] ^ lo.z); d += __popc(q[3] ^ lo.w);
    d += __popc(q[4] ^ hi.x); d += __popc(q[5] ^ hi.y); d += __popc(q[6] ^ hi.z); d += __popc(q[7] ^ hi.w);
    return d;
}

// ---------------------------------------------------------------------------------------------------
// Variants 0 / 1: row-per-lane.  ARGMIN = false tracks the best DISTANCE per query row; ARGMIN = true also finds the
// FIRST train row that attains it (cv::BFMatcher's strict-'<' ascending scan) at the same inner-loop cost:
//
//   * the inner loop is the distance-only one in both cases (fold2_min: 8 x (2 v_xor at priority 0, 2 v_bcnt at
//     priority 3) + v_min3 per two train rows) — no index arithmetic per distance;
//   * every ARGMIN_GROUP (8) train rows, each lane folds `running_min << 22 | group` into a PRIVATE LDS word per
//     query row with ds_min_u32 (1 v_lshl_or_b32 + 1 LDS atomic per 8 distances; the LDS instruction does not
//     hold the VALU).  The running minimum never rises, so the word ends up holding (best distance, FIRST group in
//     which the running minimum reached it);
//   * after the scan, the lane re-reads just that group's 8 rows (per-lane global loads, L2-resident: the frame
//     was streamed a moment ago) and takes min(dist << 22 | row): the first row of the first group with the best
//     distance = the first minimum.  8 of ~2000 rows = 0.4 % extra distances; each row is ONE
//     round trip (two 16-byte loads), because with short stored frames the chain of dependent loads, not the
//     arithmetic, is what the argmin costs (256-row frames: +100 % with 2 round trips x 16 rows, +6 % now).
//
// The words are lane-private (index j * THREADS + tid): LDS serves as 8 extra registers with a free min-ALU, no
// barrier is involved.
// ---------------------------------------------------------------------------------------------------
constexpr int ARGMIN_GROUP = 8;       // train rows per (dist, group) key; a multiple of the 4 rows of one loop trip

// ARGMIN_MODE: 0 = distances only; 1 = argmin by group keys + re-scan (bulk: throughput, the re-scan's dependent loads
// hide behind the other waves); 2 = argmin by a packed key per distance (v_lshl_or_b32 per distance, +8 % VALU, no
// re-scan: the pair mode, where a launch is a few dozen workgroups and the re-scan's ~250 dependent loads per lane
// would be most of its latency).
//
// PACKED (bulk search, 256 x 8 only): the workgroup owns 2048 consecutive rows of a VIRTUAL row space in which the
// query frames of a chunk follow one another without gaps (ScoreArgs::pk_*), so a 2000-row ORB frame no longer leaves
// 48 of the 2048 lane slots idle (measured worth: 2.5 %, profiles/r02_idle_lanes.txt).  A lane's 8 rows may then
// belong to different query frames (and a frame above 2048 rows spans several workgroups): each (lane, j) keeps the
// packed position of its row's frame in a lane-private LDS word, the per-pair reductions move to k_finalize_bulk, and the epilogue is 8 stores per lane per stored frame
// (best distance, or best key in the argmin mode) — skipped where the stored frame is not eligible for that row's
// query frame (the frames are packed by descending eligibility, so that is the last few slots of a column only).
template <int THREADS, int QPT, int ARGMIN_MODE, bool WRITE_KEYS, bool PACKED = false>
// Waves per SIMD the register budget is cut for: 6 (80 VGPRs) for the throughput kernels, 8 for the 6-rows-per-lane A/B,
// 5 (96 VGPRs) for the per-distance-key kernel of the pair mode — two values more than 80 are live in it, its launches
// are a few dozen workgroups, and occupancy is not what they wait for.
__global__ __launch_bounds__(THREADS, (PACKED && QPT == 6) ? 8 : (ARGMIN_MODE == 2 ? 5 : 6)) void k_score_rowlane(ScoreArgs a) {
    // ARGMIN_MODE = 0 with WRITE_KEYS = true writes the best DISTANCE per query row (split mode, see k_finalize_pairs)
    constexpr bool ARGMIN = ARGMIN_MODE != 0;
    constexpr bool GROUPED = ARGMIN_MODE == 1;
    constexpr bool KEYED = ARGMIN_MODE == 2;
    constexpr int DSHIFT = ARGMIN ? KEY_SHIFT : 0;      // best[j] >> DSHIFT is the best distance (epilogue)
    __shared__ uint32_t red_min[2];
    __shared__ uint32_t red_sum[2];
    __shared__ uint32_t red_idx[2];
    __shared__ uint32_t lane_key[(GROUPED || PACKED) ? THREADS * QPT : 1];   // GROUPED: (dist, group) keys; PACKED: also the epilogue's staging
    __shared__ uint32_t lane_meta[PACKED ? THREADS * QPT : 1];     // PACKED: packed position of the row's frame, 0xFFFFFFFF = idle

    const int tid = threadIdx.x;
    if (tid == 0) { red_min[0] = red_min[1] = 0xFFFFFFFFu; red_sum[0] = red_sum[1] = 0u; red_idx[0] = red_idx[1] = 0u; }
    __syncthreads();

    WorkItem it;
    int nq;
    size_t q_word0;                            // first dword of this item's query rows in q_rows
    uint32_t pair_t_row = 0;
    int pair_nt = -1;                          // >= 0: pair mode (one train segment given by the item itself)
    if (PACKED) {                              // q_frame = column of the virtual row space, out_offset = its first packed position
        it = a.items[blockIdx.x];
        nq = (int)min((uint32_t)(THREADS * QPT), a.pk_vstart[a.pk_n] - it.q_frame * (uint32_t)(THREADS * QPT));
        q_word0 = 0;
    } else if (a.pair_items) {
        const PairItem pi = a.pair_items[blockIdx.x];
        it.q_frame = 0; it.slot_begin = 0; it.n_slots = 1; it.out_offset = pi.out_offset;
        nq = (int)(pi.nq_nt & 0xFFFu);
        pair_nt = (int)(pi.nq_nt >> 12);
        pair_t_row = pi.t_row;
        q_word0 = (size_t)pi.q_row * 8;
    } else if (a.items) {
        it = a.items[blockIdx.x];
        nq = a.q_counts[it.q_frame];
        q_word0 = (size_t)it.q_frame * a.q_stride_words;
    } else {                                   // implicit item, derived from blockIdx (see ScoreArgs)
        uint32_t b = blockIdx.x, qi = 0, total = a.imp_total, pair0 = 0;
        int rows = a.imp_nq;
        if (a.imp_nbatch) {                    // micro-batch: which query does this workgroup belong to (<= 16: scan)
            while (qi + 1 < a.imp_nbatch && b >= a.bat_wg[qi + 1]) ++qi;
            b -= a.bat_wg[qi];
            total = a.bat_elig[qi]; pair0 = a.bat_pair[qi]; rows = a.bat_nq[qi];
        }
        const uint32_t c = b % a.imp_chunks, g = b / a.imp_chunks;
        it.q_frame = qi * a.imp_chunks + c;
        it.slot_begin = g * a.imp_spi;
        it.n_slots = min(a.imp_spi, total - it.slot_begin);
        it.out_offset = (pair0 + it.slot_begin) * a.imp_chunks + c;
        nq = min((int)a.imp_chunk_rows, rows - (int)(c * a.imp_chunk_rows));
        q_word0 = (size_t)it.q_frame * a.q_stride_words;
    }

    // ---- load this lane's query rows: row = j*THREADS + tid (consecutive lanes -> consecutive 32-byte rows)
    uint32_t q[QPT][8];
    auto valid = [&](int j) { return j * THREADS + tid < nq; };     // recomputed where needed: keeps VGPRs <= 80
    if (PACKED) {
        // Phase 1 (a rolled loop, on purpose): the packed position of each of this lane's rows goes to its LDS word.
        // Rows ascend with j, so the position only moves forward from the column's first one (it.out_offset).  Kept
        // out of the unrolled code below so that no position stays in a register across the scan: the main loop has
        // none to spare.
        {
            uint32_t c = it.out_offset;
            const uint32_t v0 = it.q_frame * (uint32_t)(THREADS * QPT);
#pragma unroll 1
            for (int j = 0; j < QPT; ++j) {
                const uint32_t v = v0 + (uint32_t)(j * THREADS + tid);
                uint32_t meta = 0xFFFFFFFFu;
                if (j * THREADS + tid < nq) {
                    while (a.pk_vstart[c + 1] <= v) ++c;            // v < pk_vstart[pk_n]: stops at c < pk_n
                    meta = c;
                }
                lane_meta[j * THREADS + tid] = meta;
            }
        }
        // Phase 2: load the rows (row r of frame pk_qframe[c]; a frame above 2048 rows spans columns)
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            __builtin_amdgcn_sched_barrier(0);      // one row at a time: interleaving the 8 address chains costs registers
            uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
            const uint32_t c = lane_meta[j * THREADS + tid];
            if (c != 0xFFFFFFFFu) {
                const uint32_t r = it.q_frame * (uint32_t)(THREADS * QPT) + (uint32_t)(j * THREADS + tid) - a.pk_vstart[c];
                const uint4* qb = reinterpret_cast<const uint4*>(a.q_rows + (size_t)a.pk_qframe[c] * a.q_stride_words);
                lo = qb[(size_t)r * 2]; hi = qb[(size_t)r * 2 + 1];
            }
            q[j][0] = lo.x; q[j][1] = lo.y; q[j][2] = lo.z; q[j][3] = lo.w;
            q[j][4] = hi.x; q[j][5] = hi.y; q[j][6] = hi.z; q[j][7] = hi.w;
        }
    } else {
        const uint4* qbase = reinterpret_cast<const uint4*>(a.q_rows + q_word0);
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int row = j * THREADS + tid;
            uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
            if (row < nq) { lo = qbase[row * 2]; hi = qbase[row * 2 + 1]; }
            q[j][0] = lo.x; q[j][1] = lo.y; q[j][2] = lo.z; q[j][3] = lo.w;
            q[j][4] = hi.x; q[j][5] = hi.y; q[j][6] = hi.z; q[j][7] = hi.w;
        }
    }

    for (uint32_t s = 0; s < it.n_slots; ++s) {
        const uint32_t slot = it.slot_begin + s;
        const int nt = pair_nt >= 0 ? pair_nt : ((siptr_t)a.db_counts)[slot];
        const uint32_t* Tbase = pair_nt >= 0 ? a.db_rows + (size_t)pair_t_row * 8 : a.db_rows + (size_t)slot * a.db_stride_words;
        sptr_t T = (sptr_t)Tbase;

        uint32_t best[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) best[j] = 0xFFFFFFFFu;
        if (GROUPED) {
#pragma unroll
            for (int j = 0; j < QPT; ++j) lane_key[j * THREADS + tid] = 0xFFFFFFFFu;
        }
        // (dist, group) fold of this lane's running minima into its private LDS words
        auto fold_group = [&](uint32_t g) {
#if LCM_INNER_PRIO
            __builtin_amdgcn_s_setprio(3);       // quarter-rate packs: with the popcounts' priority (see fold2)
#endif
#pragma unroll
            for (int j = 0; j < QPT; ++j) atomicMin(&lane_key[j * THREADS + tid], (best[j] << KEY_SHIFT) | g);
#if LCM_INNER_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        // Train rows are stored padded to a multiple of 4 rows with copies of the LAST real row: a copy has the
        // same distance and a higher index than the row it copies, so it can never win the (dist, idx) minimum.
        // Two 16-dword SGPR buffers (2 rows each) ping-pong: the s_load of the next 2 rows is in flight while the
        // VALU works on the current 2 (SMEM returns out of order, so the only legal wait is lgkmcnt(0)).
        if (nt > 0) {
            uint32_t A[16], B[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) A[k] = T[k];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            for (int t = 0; t < nt; t += 4) {
#pragma unroll
                for (int k = 0; k < 16; ++k) B[k] = T[(t + 2) * 8 + k];
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the VALU block it overlaps
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    if (KEYED) fold2(best[j], q[j], A, (uint32_t)t, (uint32_t)(t + 1)); else fold2_min(best[j], q[j], A);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): B landed while A was being consumed
#pragma unroll
                for (int k = 0; k < 16; ++k) A[k] = T[(t + 4) * 8 + k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    if (KEYED) fold2(best[j], q[j], B, (uint32_t)(t + 2), (uint32_t)(t + 3)); else fold2_min(best[j], q[j], B);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (GROUPED && (t & (ARGMIN_GROUP - 4)) == (ARGMIN_GROUP - 4)) fold_group((uint32_t)t / ARGMIN_GROUP);
                __builtin_amdgcn_s_waitcnt(0xC07F);  // A (rows t+4, t+5) landed while B was being consumed
            }
            if (GROUPED) {
                // the last, possibly partial group (a repeat of an already folded group changes nothing) ...
                fold_group((uint32_t)(nt - 1) / ARGMIN_GROUP);
                // ... then the re-scan of each query row's winning group for the first row that attains the minimum
                const uint32_t* Tg = Tbase;                                              // wave-uniform frame base
                const uint32_t last_off = (uint32_t)(nt - 1) * 32u;
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    // one query row at a time, the group's rows one after the other: the re-scan is < 1 % of the work
                    // and must not cost the main loop a register (64 of its 80 hold the query rows)
                    __builtin_amdgcn_sched_barrier(0);
                    if (!valid(j)) continue;
                    const uint32_t row0 = (lane_key[j * THREADS + tid] & KEY_IDX_MASK) * ARGMIN_GROUP;
                    uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll 1
                    for (uint32_t r = 0; r < (uint32_t)ARGMIN_GROUP; ++r) {
                        // rows past the frame's end re-read the last row under a HIGHER index: they cannot win
                        const uint32_t d = rescan_distance(q[j], Tg, min((row0 + r) * 32u, last_off));
                        kmin = min(kmin, (d << KEY_SHIFT) | (row0 + r));
                    }
                    lane_key[j * THREADS + tid] = kmin;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (GROUPED && !PACKED) {   // from here on best[j] is the packed key dist << 22 | first train row (0xFFFFFFFF: no train rows)
#pragma unroll
            for (int j = 0; j < QPT; ++j) best[j] = lane_key[j * THREADS + tid];
        }

        if (PACKED) {        // best distance / key of every (eligible pair, query row); k_finalize_bulk forms the records
            // staged through the lane's LDS words and written by a ROLLED loop: a handful of live registers instead of
            // eight address chains (the scan's 64 query-row registers stay untouched, nothing spills)
            if (!GROUPED) {
#pragma unroll
                for (int j = 0; j < QPT; ++j) lane_key[j * THREADS + tid] = best[j];
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t col0 = it.q_frame * (uint32_t)(THREADS * QPT);      // first virtual row of the column (wave-uniform)
            asm volatile("" : "+s"(col0));       // opaque here: `col0 + tid` must not be hoisted above the scan, where it would cost a register
#pragma unroll 1
            for (int j = 0; j < QPT; ++j) {
                const uint32_t c = lane_meta[j * THREADS + tid];
                if (c == 0xFFFFFFFFu) continue;
                const uint32_t r = col0 + (uint32_t)(j * THREADS + tid) - a.pk_vstart[c];
                if (slot < a.pk_elig[c]) {
                    const size_t w = ((size_t)a.pk_pairs[c] + (slot - a.pk_slot0)) * a.pk_stride + r;
                    // distance-only: a best distance is <= 256 (0xFFFF: the stored frame is empty) -> 2-byte scratch words
                    if (GROUPED) a.pk_dist[w] = lane_key[j * THREADS + tid];
                    else reinterpret_cast<uint16_t*>(a.pk_dist)[w] = (uint16_t)lane_key[j * THREADS + tid];
                }
            }
            continue;
        }
        // ---- pair epilogue: min-of-mins, ratio filter count, one score record
        const size_t out = (size_t)it.out_offset + s;
        if (WRITE_KEYS) {
#pragma unroll
            for (int j = 0; j < QPT; ++j)
                if (valid(j)) a.keys[out * a.keys_stride + j * THREADS + tid] = best[j];
        }
        // split mode (per-row best DISTANCES for k_finalize_pairs): this workgroup sees only a chunk of the query
        // frame, so there is no pair record to form here
        if (WRITE_KEYS && !ARGMIN) continue;
        // min-of-mins and good-match count: per-lane partials folded with LDS atomics (ds_min_u32 / ds_add_u32), one
        // word per pair parity.  Deliberately not a shuffle tree: this runs once per 4M distances, and the atomics
        // need no extra VGPRs, which keeps the kernel inside the 80-register budget of 6 waves/SIMD without spills.
        const int par = s & 1;
        uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < QPT; ++j) if (valid(j)) dmin = min(dmin, best[j] >> DSHIFT);
        atomicMin(&red_min[par], dmin);
        __syncthreads();
        dmin = red_min[par];
        if (tid == 0) { red_min[par ^ 1] = 0xFFFFFFFFu; red_sum[par ^ 1] = 0u; red_idx[par ^ 1] = 0u; }   // next pair's words (idle until B)
        uint32_t thr = (uint32_t)a.ratio * dmin;
        thr = max(thr, (uint32_t)a.dist_floor);
        uint32_t cnt = 0, isum = 0;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const bool good = valid(j) && (best[j] >> DSHIFT) <= thr;
            cnt += good ? 1u : 0u;
            if (ARGMIN) isum += good ? (best[j] & KEY_IDX_MASK) : 0u;
        }
        atomicAdd(&red_sum[par], cnt);
        if (ARGMIN && a.idx_sums) atomicAdd(&red_idx[par], isum);
        __syncthreads();
        cnt = red_sum[par];
        if (tid == 0) {
            const bool empty = (nq <= 0) || (nt <= 0);
            uint2 rec;
            rec.x = empty ? 0u : cnt;
            rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((uint32_t)(nt & 0xFFFF) << 16);
            if (a.scores) reinterpret_cast<uint2*>(a.scores)[out] = rec;
            if (ARGMIN && a.idx_sums) a.idx_sums[out] = empty ? 0u : red_idx[par];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Variants 2 / 3 (k_score_trainlane): the mapping BASELINE.json's north_star sketches — lanes own TRAIN rows (coalesced 16-byte loads, 8 rows
// per lane = a whole 2048-row stored frame in a workgroup's registers), QUERY rows staged in LDS in 256-row tiles and
// broadcast to all lanes with ds_read_b128, per-query min(/argmin) by a wavefront __shfl_xor reduction, waves combined
// with ds_min_u32.  Kept selectable (lcm_set_kernel_variant(2) / 3) so that the choice of variant 0 rests on a
// measurement: this mapping needs ~1.2 (distances) to ~2.2 (keys) extra 4-cycle VALU ops per distance for the
// cross-lane reduction, see DESIGN.md §4.
// ---------------------------------------------------------------------------------------------------
// one broadcast query row (8 VGPRs, same value in every lane) against TWO of this lane's train rows
__device__ __forceinline__ void fold2_vv(uint32_t& best, const uint32_t (&q)[8], const uint32_t (&t0)[8], const uint32_t (&t1)[8]) {
#if LCM_INNER_PRIO
    uint32_t d0, d1, x0, x1;
    asm volatile(
        "v_xor_b32_e32 %3, %5, %13\n\tv_xor_b32_e32 %4, %5, %21\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, 0\n\tv_bcnt_u32_b32 %2, %4, 0\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %6, %14\n\tv_xor_b32_e32 %4, %6, %22\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %7, %15\n\tv_xor_b32_e32 %4, %7, %23\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %8, %16\n\tv_xor_b32_e32 %4, %8, %24\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %9, %17\n\tv_xor_b32_e32 %4, %9, %25\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %10, %18\n\tv_xor_b32_e32 %4, %10, %26\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %11, %19\n\tv_xor_b32_e32 %4, %11, %27\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %12, %20\n\tv_xor_b32_e32 %4, %12, %28\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\t"
        "v_min3_u32 %0, %0, %1, %2\n\ts_setprio 0"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x0), "=&v"(x1)
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t0[4]), "v"(t0[5]), "v"(t0[6]), "v"(t0[7]),
          "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]), "v"(t1[4]), "v"(t1[5]), "v"(t1[6]), "v"(t1[7]));
#else
    uint32_t d0, d1, x;
    asm volatile(
        "v_xor_b32_e32 %3, %4, %12\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, 0\n\t"
        "v_xor_b32_e32 %3, %5, %13\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %6, %14\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %7, %15\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %8, %16\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %9, %17\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %10, %18\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %11, %19\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %4, %20\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, 0\n\t"
        "v_xor_b32_e32 %3, %5, %21\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %6, %22\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %7, %23\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %8, %24\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %9, %25\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %10, %26\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %11, %27\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_min3_u32 %0, %0, %1, %2"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x)
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t0[4]), "v"(t0[5]), "v"(t0[6]), "v"(t0[7]),
          "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]), "v"(t1[4]), "v"(t1[5]), "v"(t1[6]), "v"(t1[7]));
#endif
}
// same, folding packed keys dist << 22 | train row (k0, k1 = this lane's two row indices)
__device__ __forceinline__ void fold2_vv_keys(uint32_t& best, const uint32_t (&q)[8], const uint32_t (&t0)[8], const uint32_t (&t1)[8],
                                              uint32_t k0, uint32_t k1) {
#if LCM_INNER_PRIO
    uint32_t d0, d1, x0, x1;
    asm volatile(
        "v_xor_b32_e32 %3, %5, %13\n\tv_xor_b32_e32 %4, %5, %21\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, 0\n\tv_bcnt_u32_b32 %2, %4, 0\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %6, %14\n\tv_xor_b32_e32 %4, %6, %22\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %7, %15\n\tv_xor_b32_e32 %4, %7, %23\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %8, %16\n\tv_xor_b32_e32 %4, %8, %24\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %9, %17\n\tv_xor_b32_e32 %4, %9, %25\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %10, %18\n\tv_xor_b32_e32 %4, %10, %26\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %11, %19\n\tv_xor_b32_e32 %4, %11, %27\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\ts_setprio 0\n\t"
        "v_xor_b32_e32 %3, %12, %20\n\tv_xor_b32_e32 %4, %12, %28\n\ts_setprio 3\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\t"
        "v_lshl_or_b32 %1, %1, 22, %29\n\t"
        "v_lshl_or_b32 %2, %2, 22, %30\n\t"
        "v_min3_u32 %0, %0, %1, %2\n\ts_setprio 0"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x0), "=&v"(x1)
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t0[4]), "v"(t0[5]), "v"(t0[6]), "v"(t0[7]),
          "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]), "v"(t1[4]), "v"(t1[5]), "v"(t1[6]), "v"(t1[7]),
          "v"(k0), "v"(k1));
#else
    uint32_t d0, d1, x;
    asm volatile(
        "v_xor_b32_e32 %3, %4, %12\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, 0\n\t"
        "v_xor_b32_e32 %3, %5, %13\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %6, %14\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %7, %15\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %8, %16\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %9, %17\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %10, %18\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %11, %19\n\ts_nop 0\n\tv_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_xor_b32_e32 %3, %4, %20\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, 0\n\t"
        "v_xor_b32_e32 %3, %5, %21\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %6, %22\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %7, %23\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %8, %24\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %9, %25\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %10, %26\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_xor_b32_e32 %3, %11, %27\n\ts_nop 0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"
        "v_lshl_or_b32 %1, %1, 22, %28\n\t"
        "v_lshl_or_b32 %2, %2, 22, %29\n\t"
        "v_min3_u32 %0, %0, %1, %2"
        : "+v"(best), "=&v"(d0), "=&v"(d1), "=&v"(x)
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
          "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t0[4]), "v"(t0[5]), "v"(t0[6]), "v"(t0[7]),
          "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]), "v"(t1[4]), "v"(t1[5]), "v"(t1[6]), "v"(t1[7]),
          "v"(k0), "v"(k1));
#endif
}

template <bool ARGMIN, bool WRITE_KEYS>
__global__ __launch_bounds__(256) void k_score_trainlane(ScoreArgs a) {
    constexpr int THREADS = 256, TPT = 8, QTILE = 256;
    constexpr int DSHIFT = ARGMIN ? KEY_SHIFT : 0;
    __shared__ uint4 qtile[QTILE * 2];                 // 256 query rows x 32 bytes
    __shared__ uint32_t best_q[THREADS * TPT];         // per query row: best distance / key of this pair
    __shared__ uint32_t red_min, red_sum;

    const int tid = threadIdx.x;
    WorkItem it;
    int nq;
    if (a.items) {
        it = a.items[blockIdx.x];
        nq = a.q_counts[it.q_frame];
    } else {
        const uint32_t c = blockIdx.x % a.imp_chunks, g = blockIdx.x / a.imp_chunks;
        it.q_frame = c;
        it.slot_begin = g * a.imp_spi;
        it.n_slots = min(a.imp_spi, a.imp_total - it.slot_begin);
        it.out_offset = it.slot_begin * a.imp_chunks + c;
        nq = min((int)a.imp_chunk_rows, a.imp_nq - (int)(c * a.imp_chunk_rows));
    }
    const uint4* qbase = reinterpret_cast<const uint4*>(a.q_rows + (size_t)it.q_frame * a.q_stride_words);

    for (uint32_t s = 0; s < it.n_slots; ++s) {
        const uint32_t slot = it.slot_begin + s;
        const int nt = a.db_counts[slot];
        const uint4* tbase = reinterpret_cast<const uint4*>(a.db_rows + (size_t)slot * a.db_stride_words);
        // this lane's train rows j*256 + tid; rows past the end are copies of the LAST row (same distance, higher
        // index: can never win), so no lane needs masking in the inner loop
        uint32_t t[TPT][8];
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            int row = j * THREADS + tid;
            row = row < nt ? row : max(nt - 1, 0);
            const uint4 lo = tbase[row * 2], hi = tbase[row * 2 + 1];
            t[j][0] = lo.x; t[j][1] = lo.y; t[j][2] = lo.z; t[j][3] = lo.w;
            t[j][4] = hi.x; t[j][5] = hi.y; t[j][6] = hi.z; t[j][7] = hi.w;
        }
#pragma unroll
        for (int j = 0; j < TPT; ++j) best_q[j * THREADS + tid] = 0xFFFFFFFFu;
        if (tid == 0) { red_min = 0xFFFFFFFFu; red_sum = 0u; }

        for (int q0 = 0; q0 < nq && nt > 0; q0 += QTILE) {
            __syncthreads();                                        // previous tile fully consumed (and best_q init visible)
            if (q0 + tid < nq) { qtile[tid * 2] = qbase[(q0 + tid) * 2]; qtile[tid * 2 + 1] = qbase[(q0 + tid) * 2 + 1]; }
            __syncthreads();
            const int nrows = min(QTILE, nq - q0);
            for (int i = 0; i < nrows; ++i) {
                const uint4 lo = qtile[i * 2], hi = qtile[i * 2 + 1];      // same address in every lane: LDS broadcast
                const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                uint32_t b = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < TPT; j += 2) {
                    if (ARGMIN) fold2_vv_keys(b, q, t[j], t[j + 1], (uint32_t)(j * THREADS + tid), (uint32_t)((j + 1) * THREADS + tid));
                    else fold2_vv(b, q, t[j], t[j + 1]);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) b = min(b, (uint32_t)__shfl_xor((int)b, o, 64));   // wavefront reduction
                if ((tid & 63) == 0) atomicMin(&best_q[q0 + i], b);                                    // 4 waves -> 1
            }
        }
        __syncthreads();
        // ---- pair epilogue (same record as variant 0)
        const size_t out = (size_t)it.out_offset + s;
        uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int row = j * THREADS + tid;
            if (row < nq) {
                const uint32_t k = best_q[row];
                if (WRITE_KEYS) a.keys[out * a.keys_stride + row] = k;
                dmin = min(dmin, k >> DSHIFT);
            }
        }
        atomicMin(&red_min, dmin);
        __syncthreads();
        dmin = red_min;
        uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int row = j * THREADS + tid;
            cnt += (row < nq && (best_q[row] >> DSHIFT) <= thr) ? 1u : 0u;
        }
        atomicAdd(&red_sum, cnt);
        __syncthreads();
        if (tid == 0) {
            const bool empty = (nq <= 0) || (nt <= 0);
            uint2 rec;
            rec.x = empty ? 0u : red_sum;
            rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((uint32_t)(nt & 0xFFFF) << 16);
            reinterpret_cast<uint2*>(a.scores)[out] = rec;
        }
        __syncthreads();                                            // red_* / best_q are rewritten by the next slot
    }
}

template <int THREADS, int QPT>
static hipError_t launch_rowlane(const ScoreArgs& a, uint32_t n_items, bool write_keys, bool argmin, hipStream_t st) {
    if (n_items == 0) return hipSuccess;
    // 78-80 VGPRs (QPT = 8) => 6 waves per SIMD by the register file alone: no other occupancy control is needed
    const unsigned lds = 0;
    if (write_keys && argmin)        // pair mode / match lists: a key per distance, no re-scan (latency)
        hipLaunchKernelGGL((k_score_rowlane<THREADS, QPT, 2, true>), dim3(n_items), dim3(THREADS), lds, st, a);
    else if (write_keys)             // split mode: best distance per query row
        hipLaunchKernelGGL((k_score_rowlane<THREADS, QPT, 0, true>), dim3(n_items), dim3(THREADS), lds, st, a);
    else if (argmin)                 // bulk argmin: group keys + re-scan (throughput)
        hipLaunchKernelGGL((k_score_rowlane<THREADS, QPT, 1, false>), dim3(n_items), dim3(THREADS), lds, st, a);
    else
        hipLaunchKernelGGL((k_score_rowlane<THREADS, QPT, 0, false>), dim3(n_items), dim3(THREADS), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Split mode for SHORT databases (online queries against fewer than ~1000 stored frames): one workgroup per pair
// leaves the chip under-filled, so a pair's query rows are cut into chunks of 256 * QPT rows (QPT = 4, 2 or 1 rows per
// lane -> 2, 4 or 8 workgroups per pair), every chunk writes its rows' best distances, and this kernel — one
// workgroup per pair — folds them into the usual score record (min-of-mins, ratio filter, count).
// ---------------------------------------------------------------------------------------------------
// One WAVE per pair: the pair's <= 2048 per-row distances are read once (8 coalesced 16-byte loads per lane) and stay
// in registers for both passes; wave reductions are shuffles.
__global__ __launch_bounds__(256) void k_finalize_pairs(FinalizeArgs a, uint32_t n_pairs) {
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;                                // whole wave: no barrier follows
    int nq = a.nq;
    uint32_t slot = a.slot_begin + pair;
    if (a.n_batch) {
        uint32_t b = 0;
        while (b + 1 < a.n_batch && pair >= a.bat_pair[b + 1]) ++b;
        nq = a.bat_nq[b];
        slot = pair - a.bat_pair[b];
    }
    const uint4* d = reinterpret_cast<const uint4*>(a.dist + (size_t)pair * a.padded_rows);     // padded_rows % 256 == 0
    uint32_t v[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r0 = (i * 64 + lane) * 4;                     // rows r0 .. r0 + 3; rows >= nq were never written: masked
        uint4 x = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (r0 < nq) x = d[i * 64 + lane];
        v[4 * i + 0] = x.x;
        v[4 * i + 1] = r0 + 1 < nq ? x.y : 0xFFFFFFFFu;
        v[4 * i + 2] = r0 + 2 < nq ? x.z : 0xFFFFFFFFu;
        v[4 * i + 3] = r0 + 3 < nq ? x.w : 0xFFFFFFFFu;
    }
    uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 32; ++k) dmin = min(dmin, v[k]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
    const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) cnt += (v[k] != 0xFFFFFFFFu && v[k] <= thr) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
    if (lane == 0) {
        const int nt = a.db_counts[slot];
        const bool empty = (nq <= 0) || (nt <= 0) || dmin == 0xFFFFFFFFu;
        uint2 rec;
        rec.x = empty ? 0u : cnt;
        rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((uint32_t)(nt & 0xFFFF) << 16);
        reinterpret_cast<uint2*>(a.scores)[pair] = rec;
    }
}

// 32 query rows per workgroup, 8 threads per row: thread (row, part) folds segments part, part + 8, ... (a handful of
// independent loads instead of one thread walking all segments one dependent load after the other — that walk was most
// of a single matchFeatures call's fold time), the 8 partial minima meet in LDS.
__global__ __launch_bounds__(256) void k_fold_pair_keys(FoldArgs a) {
    __shared__ uint32_t part_min[8][32];
    const PairDesc p = a.pairs[blockIdx.y];
    const uint32_t rr = threadIdx.x & 31u, part = threadIdx.x >> 5;
    const uint32_t r = blockIdx.x * 32u + rr;
    const uint32_t CR = a.chunk_rows ? a.chunk_rows : (uint32_t)MAX_FUSED_QUERY_ROWS;
    uint32_t best = 0xFFFFFFFFu;
    if (r < p.nq) {
        const uint32_t c = r / CR, lr = r % CR;
        const uint32_t* src = a.seg_keys + ((size_t)p.first_item + (size_t)c * p.n_seg) * CR + lr;
        for (uint32_t g = part; g < p.n_seg; g += 8) best = min(best, src[(size_t)g * CR] + g * p.seg_rows);
    }
    part_min[part][rr] = best;
    __syncthreads();
    if (part == 0 && r < p.nq) {
#pragma unroll
        for (int k = 1; k < 8; ++k) best = min(best, part_min[k][rr]);
        a.final_keys[p.out_row0 + r] = best;
    }
}

hipError_t launch_fold_pair_keys(const FoldArgs& a, uint32_t max_nq, hipStream_t st) {
    if (a.n_pairs == 0 || max_nq == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fold_pair_keys, dim3((max_nq + 31) / 32, a.n_pairs), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Upload by kernel: n16 16-byte words from pinned host memory (mapped into the device's address space) to device memory.
// For the pair mode's staging block (<= a few hundred KB): on the compute queue a copy is followed by the kernel that
// needs it after a normal kernel-to-kernel gap, where a DMA-engine copy costs a cross-engine hand-over first.
__global__ __launch_bounds__(256) void k_upload_u4(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

hipError_t launch_upload(void* d_dst, const void* h_src_pinned, size_t bytes, hipStream_t st) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(k_upload_u4, dim3((n16 + 255) / 256), dim3(256), 0, st, (uint4*)d_dst, (const uint4*)h_src_pinned, n16);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs& a, uint32_t n_pairs, hipStream_t st) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_pairs, dim3((n_pairs + 3) / 4), dim3(256), 0, st, a, n_pairs);
    return hipGetLastError();
}

hipError_t launch_score_packed(const ScoreArgs& a, uint32_t n_items, bool argmin, hipStream_t st) {
    if (n_items == 0) return hipSuccess;
    if (a.pk_col_rows == 1536) {          // EXPERIMENT: 6 rows per lane, 8 waves per SIMD
        if (argmin) hipLaunchKernelGGL((k_score_rowlane<256, 6, 1, false, true>), dim3(n_items), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_score_rowlane<256, 6, 0, false, true>), dim3(n_items), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (argmin) hipLaunchKernelGGL((k_score_rowlane<256, 8, 1, false, true>), dim3(n_items), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_score_rowlane<256, 8, 0, false, true>), dim3(n_items), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_score_pairs_small(const ScoreArgs& a, uint32_t n_items, hipStream_t st) {
    return launch_rowlane<256, 2>(a, n_items, true, true, st);
}

// Split-mode launch: 256-thread workgroups holding `qpt` query rows per lane, writing best distances to a.keys.
hipError_t launch_score_split(const ScoreArgs& a, uint32_t n_items, int qpt, hipStream_t st) {
    switch (qpt) {
        case 4: return launch_rowlane<256, 4>(a, n_items, true, false, st);
        case 2: return launch_rowlane<256, 2>(a, n_items, true, false, st);
        case 1: return launch_rowlane<256, 1>(a, n_items, true, false, st);
        case 16: return launch_rowlane<64, 8>(a, n_items, true, false, st);       // 512-row chunks, one wave per workgroup
        case 32: return launch_rowlane<128, 8>(a, n_items, true, false, st);      // 1024-row chunks, two waves
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------
// On-device loop test (README.md:123-126) over the score array, candidates compacted IN PAIR ORDER — (query frame
// ascending, stored frame ascending) = (current id, matched id) ascending, the order detectLoops reports — so the host
// never sorts: k_loop_count (verdict per pair, candidates per 256-pair block), k_block_scan (exclusive prefix of the
// block counts, one workgroup), k_loop_emit (verdict again, rank inside the block by wave ballots, write).
// The division is IEEE double (no fast-math), so the verdict is bit-identical to the host's lcm_loop_test.
// HBM-bound: 2 x 8 bytes read per pair + 24 bytes written per candidate.
// ---------------------------------------------------------------------------------------------------
struct CandidateRec { int32_t cur, matched, num; int32_t pad; double sim; };
static_assert(sizeof(CandidateRec) == 24, "lcm_loop_candidate layout");

__device__ __forceinline__ bool loop_verdict(const LoopTestArgs& a, uint32_t p, CandidateRec& r) {
    if (p >= a.n_pairs) return false;
    // query frame of pair p: last c with offsets[c] <= p
    uint32_t lo = 0, hi = a.n_q;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.offsets[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t c = lo, slot = p - a.offsets[c];
    const uint2 rec = reinterpret_cast<const uint2*>(a.scores)[p];
    const uint32_t good = rec.x;
    const int den = min(a.q_kp[c], a.db_kp[slot]);
    if (den <= 0 || (int64_t)good < (int64_t)a.min_matches) return false;
    const double sim = (double)good / (double)den;
    if (!(sim > a.sim_threshold)) return false;
    r.cur = a.q_ids[c]; r.matched = a.db_ids[slot]; r.num = (int32_t)good; r.pad = 0; r.sim = sim;
    return true;
}

__global__ __launch_bounds__(256) void k_loop_count(LoopTestArgs a) {
    CandidateRec r;
    const bool pass = loop_verdict(a, blockIdx.x * 256u + threadIdx.x, r);
    const int n = __syncthreads_count(pass ? 1 : 0);
    if (threadIdx.x == 0) a.block_counts[blockIdx.x] = (uint32_t)n;
}

// counts[0..n) -> exclusive prefix sums in place; *total = sum.  One workgroup of 1024 threads, chunk by chunk.
__global__ __launch_bounds__(1024) void k_block_scan(uint32_t* counts, uint32_t n, uint32_t* total) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? counts[i] : 0u;
        uint32_t x = v;                                   // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wave_sum[wave] = x;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (i < n) counts[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) carry = before + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_loop_emit(LoopTestArgs a) {
    __shared__ uint32_t wave_n[4];
    CandidateRec r;
    const bool pass = loop_verdict(a, blockIdx.x * 256u + threadIdx.x, r);
    const uint64_t m = __ballot(pass);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_n[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!pass) return;
    uint32_t k = a.block_counts[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) k += wave_n[w];
    if (k < a.cap) reinterpret_cast<CandidateRec*>(a.out)[k] = r;
}

hipError_t launch_loop_count(const LoopTestArgs& a, hipStream_t st) {
    if (a.n_pairs == 0) return hipSuccess;
    const uint32_t n_blocks = (a.n_pairs + 255) / 256;
    hipLaunchKernelGGL(k_loop_count, dim3(n_blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, st, a.block_counts, n_blocks, a.counter);
    return hipGetLastError();
}

hipError_t launch_loop_emit(const LoopTestArgs& a, hipStream_t st) {
    if (a.n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_loop_emit, dim3((a.n_pairs + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cross_score(CrossArgs a) {
    __shared__ uint32_t ck[MAX_FUSED_QUERY_ROWS];      // per query row: the key it keeps, 0xFFFFFFFF = no match
    __shared__ uint32_t red_min, red_sum, red_idx;
    const CrossDesc p = a.descs[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t* fk = a.keys + (size_t)p.f_slot * MAX_FUSED_QUERY_ROWS;
    const uint32_t* bk = a.keys + (size_t)p.b_slot * MAX_FUSED_QUERY_ROWS;
    for (uint32_t q = tid; q < p.nq; q += 256) ck[q] = 0xFFFFFFFFu;
    if (tid == 0) { red_min = 0xFFFFFFFFu; red_sum = 0u; red_idx = 0u; }
    __syncthreads();
    if (p.nt > 0) {
        if (a.mode == 1) {
            for (uint32_t q = tid; q < p.nq; q += 256) {
                const uint32_t f = fk[q];
                if ((bk[f & KEY_IDX_MASK] & KEY_IDX_MASK) == q) ck[q] = f;
            }
        } else {
            for (uint32_t t = tid; t < p.nt; t += 256) {
                const uint32_t b = bk[t];
                atomicMin(&ck[b & KEY_IDX_MASK], (b & ~KEY_IDX_MASK) | t);
            }
        }
    }
    __syncthreads();
    uint32_t dmin = 0xFFFFFFFFu;
    for (uint32_t q = tid; q < p.nq; q += 256) if (ck[q] != 0xFFFFFFFFu) dmin = min(dmin, ck[q] >> KEY_SHIFT);
    atomicMin(&red_min, dmin);
    __syncthreads();
    dmin = red_min;
    const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
    uint32_t cnt = 0, isum = 0;
    for (uint32_t q = tid; q < p.nq; q += 256) {
        const uint32_t k = ck[q];
        const bool good = k != 0xFFFFFFFFu && (k >> KEY_SHIFT) <= thr;
        cnt += good ? 1u : 0u;
        isum += good ? (k & KEY_IDX_MASK) : 0u;
    }
    atomicAdd(&red_sum, cnt);
    atomicAdd(&red_idx, isum);
    __syncthreads();
    if (tid == 0) {
        const bool empty = p.nq == 0 || p.nt == 0 || dmin == 0xFFFFFFFFu;     // nothing survived the cross-check
        uint2 rec;
        rec.x = empty ? 0u : red_sum;
        rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((p.nt & 0xFFFFu) << 16);
        reinterpret_cast<uint2*>(a.scores)[p.out] = rec;
        if (a.idx_sums) a.idx_sums[p.out] = empty ? 0u : red_idx;
    }
}

hipError_t launch_cross_score(const CrossArgs& a, uint32_t n_pairs, hipStream_t st) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cross_score, dim3(n_pairs), dim3(256), 0, st, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void k_pad_rows(uint32_t* rows, const int32_t* counts, uint32_t stride_rows) {
    const int n = counts[blockIdx.x];
    if (n <= 0 || (uint32_t)n > stride_rows) return;        // empty frame, or a slot the caller never filled
    uint32_t* f = rows + (size_t)blockIdx.x * stride_rows * 8;
    const int r = n + (int)(threadIdx.x >> 3), w = (int)(threadIdx.x & 7);      // up to 8 rows x 8 dwords
    const int end = ((n + 3) & ~3) + 4;
    if (r < end && (uint32_t)r < stride_rows) f[(size_t)r * 8 + w] = f[(size_t)(n - 1) * 8 + w];
}

hipError_t launch_pad_rows(uint32_t* rows, const int32_t* counts, uint32_t stride_rows, uint32_t n_frames, hipStream_t st) {
    if (n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pad_rows, dim3(n_frames), dim3(64), 0, st, rows, counts, stride_rows);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_merge_shards(MergeArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_total) return;
    uint32_t r = 0;
    while (r + 1 < a.world && i >= a.shard_base[r + 1]) ++r;
    const uint32_t li = i - a.shard_base[r];
    const uint32_t* off_r = a.shard_offsets + (size_t)r * (a.n_q + 1);
    uint32_t lo = 0, hi = a.n_q;                      // last c with off_r[c] <= li
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off_r[mid] <= li) lo = mid; else hi = mid;
    }
    const uint32_t k = li - off_r[lo];
    const uint32_t dst = a.offsets[lo] + r + k * a.world;
    if (a.elem_words == 1) reinterpret_cast<uint32_t*>(a.merged)[dst] = reinterpret_cast<const uint32_t*>(a.gathered)[i];
    else reinterpret_cast<uint2*>(a.merged)[dst] = reinterpret_cast<const uint2*>(a.gathered)[i];
}

hipError_t launch_merge_shards(const MergeArgs& a, hipStream_t st) {
    if (a.n_total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_shards, dim3((a.n_total + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// variant 0: row-per-lane, best distance per query row (default for lcm_all_vs_all and the online queries)
// variant 1: row-per-lane, + the first train row attaining it by 8-row group keys and a re-scan (lcm_all_vs_all_argmin)
// write_keys: a packed key per distance and per-row keys written out (pair mode / match lists), whatever the variant
// variant 2 / 3: north_star's train-row-per-lane mapping, distances only / keys (stored frames of <= 2048 rows)
// All selectable so they can be measured on the same workload (bench.py --variant N).
hipError_t launch_score(const ScoreArgs& a, uint32_t n_items, int max_query_rows, bool write_keys, int variant,
                        hipStream_t st) {
    if (variant >= 4) variant = 0;        // the matrix-core variants cover the bulk search only (lcm_api.cpp: mfma_bulk)
    if (variant >= 2 && !a.pair_items && max_query_rows <= 2048 && a.db_stride_words != 0 && a.db_stride_words <= 2048 * 8) {
        if (n_items == 0) return hipSuccess;
        const bool argmin = write_keys || variant == 3;
        if (write_keys) hipLaunchKernelGGL((k_score_trainlane<true, true>), dim3(n_items), dim3(256), 0, st, a);
        else if (argmin) hipLaunchKernelGGL((k_score_trainlane<true, false>), dim3(n_items), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_score_trainlane<false, false>), dim3(n_items), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    // variant 4 (matrix cores) covers the bulk search only (lcm_api.cpp: mfma_bulk); everything else runs variant 0
    const bool argmin = write_keys || variant == 1 || variant == 3;
    if (max_query_rows <= 512) return launch_rowlane<64, 8>(a, n_items, write_keys, argmin, st);
    if (max_query_rows <= 1024) return launch_rowlane<128, 8>(a, n_items, write_keys, argmin, st);
    if (max_query_rows <= 1536) return launch_rowlane<192, 8>(a, n_items, write_keys, argmin, st);
    if (max_query_rows <= 2048) return launch_rowlane<256, 8>(a, n_items, write_keys, argmin, st);
    return hipErrorInvalidValue;
}

}  // namespace lcm
