// lcm_mfma.hip — OPT-IN matrix-core variant of the loop search (lcm_set_kernel_variant(4)); see lcm_kernels.h.
//
// Not the product default: BASELINE.json's north_star asks for XOR + popcount on the vector ALU ("no MFMA"), which
// is what variants 0 / 1 in lcm_kernels.hip do.  This file turns DESIGN.md's estimate of what that rule costs into a
// measurement, bit-exact against the same tests.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcm_kernels.h"

namespace lcm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- bits -> +1 / -1 int8 operand image -----------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_expand_pm1(const uint32_t* rows, const int32_t* counts, uint32_t stride_words,
                                                    uint32_t tiles_per_frame, uint8_t* pm1) {
    const uint32_t tile = blockIdx.x, frame = blockIdx.y;
    const int n = counts[frame];
    if ((int)(tile * 32) >= n) return;                          // the kernel never reads tiles past a frame's rows
    const uint32_t ks = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t r = min(tile * 32 + (lane & 31), (uint32_t)(n - 1));       // pad rows repeat the last row
    const uint32_t h = lane >> 5;
    const uint32_t word = rows[(size_t)frame * stride_words + (size_t)r * 8 + ks];          // bits [32 ks, 32 ks + 32)
    const uint32_t bits = (word >> (16 * h)) & 0xFFFFu;
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (((bits >> (4 * i + b)) & 1u) ? 0x01u : 0xFFu) << (8 * b);
        o[i] = v;
    }
    uint4* dst = reinterpret_cast<uint4*>(pm1 + ((size_t)frame * tiles_per_frame + tile) * PM1_TILE_BYTES) + ks * 64 + lane;
    *dst = make_uint4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_expand_pm1(const uint32_t* rows, const int32_t* counts, uint32_t stride_words, uint32_t n_frames,
                             uint32_t tiles_per_frame, uint8_t* pm1, hipStream_t st) {
    if (n_frames == 0 || tiles_per_frame == 0) return hipSuccess;
    hipLaunchKernelGGL(k_expand_pm1, dim3(tiles_per_frame, n_frames), dim3(512), 0, st, rows, counts, stride_words, tiles_per_frame, pm1);
    return hipGetLastError();
}

// ---- the MFMA scoring kernel -----------------------------------------------------------------------------------------
__device__ __forceinline__ int max16(const v16i& c) {
    int m = max(max(c[0], c[1]), c[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) m = max(max(m, c[i]), c[i + 1]);
    return max(m, c[15]);
}

// ARGMIN epilogue: the packed key (distance << 22 | row) of query row `qrow` of frame `qf` over the 32 stored rows of
// tile `tile` of slot `slot` — exact XOR + popcount on the packed rows, four rows per round trip.  The tile is the
// first one that reached the best dot product, so its minimum IS the frame's first minimum.
__device__ __forceinline__ uint32_t rescan_tile(const MfmaArgs& a, uint32_t qf, uint32_t qrow, uint32_t slot, uint32_t tile, int nt) {
    const uint4* qp = reinterpret_cast<const uint4*>(a.q_rows + (size_t)qf * a.q_stride_words + (size_t)qrow * 8);
    const uint4 qlo = qp[0], qhi = qp[1];
    const uint4* tp = reinterpret_cast<const uint4*>(a.db_rows + (size_t)slot * a.db_stride_words);
    const uint32_t r0 = tile * 32u, r1 = min(r0 + 32u, (uint32_t)nt);
    uint32_t key = 0xFFFFFFFFu;
    for (uint32_t r = r0; r < r1; r += 4) {
        uint4 lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rr = min(r + (uint32_t)k, r1 - 1);          // past the end: the last row again (higher index, cannot win)
            lo[k] = tp[(size_t)rr * 2]; hi[k] = tp[(size_t)rr * 2 + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t d = __popc(qlo.x ^ lo[k].x) + __popc(qlo.y ^ lo[k].y) + __popc(qlo.z ^ lo[k].z) + __popc(qlo.w ^ lo[k].w) +
                               __popc(qhi.x ^ hi[k].x) + __popc(qhi.y ^ hi[k].y) + __popc(qhi.z ^ hi[k].z) + __popc(qhi.w ^ hi[k].w);
            key = min(key, (d << KEY_SHIFT) | (r + (uint32_t)k));
        }
    }
    return key;
}

__global__ __launch_bounds__(256, 4) void k_score_mfma(MfmaArgs a) {
    __shared__ uint4 atile[2][512];                             // two 8 KiB stored-frame tiles (A operands)
    const MfmaItem it = a.items[blockIdx.x];
    if (it.n_slots == 0) return;                                // padding item of the XCD-interleaved order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = a.q_counts[it.q_frame];
    const uint32_t qt0 = it.q_chunk * 8 + (uint32_t)wave * 2;   // this wave's two query tiles
    const uint32_t nq_tiles = (uint32_t)(nq + 31) / 32;

    // B operands: the wave's 2 query tiles x 8 k-steps, resident for the whole item (tiles past the frame: zeros)
    v4i b0[8], b1[8];
    {
        const uint4* qb = reinterpret_cast<const uint4*>(a.q_pm1 + ((size_t)it.q_frame * a.q_tiles_per_frame + qt0) * PM1_TILE_BYTES);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 x = make_uint4(0, 0, 0, 0), y = make_uint4(0, 0, 0, 0);
            if (qt0 < nq_tiles) x = qb[ks * 64 + lane];
            if (qt0 + 1 < nq_tiles) y = qb[512 + ks * 64 + lane];
            b0[ks] = v4i{(int)x.x, (int)x.y, (int)x.z, (int)x.w};
            b1[ks] = v4i{(int)y.x, (int)y.y, (int)y.z, (int)y.w};
        }
    }

    for (uint32_t s = 0; s < it.n_slots; ++s) {
        const uint32_t slot = it.slot_begin + s;
        const int nt = a.db_counts[slot];
        const uint32_t nt_tiles = (uint32_t)(nt + 31) / 32;
        const uint4* tb = reinterpret_cast<const uint4*>(a.db_pm1 + (size_t)slot * a.db_tiles_per_frame * PM1_TILE_BYTES);
        int best0 = -0x7FFFFFFF, best1 = -0x7FFFFFFF;
        uint32_t bt0 = 0, bt1 = 0;                               // first tile that reached best0 / best1 (argmin form)
        if (nt_tiles > 0) {
            // stage tile 0, then: compute tile t from LDS while tile t + 1 travels global -> registers -> LDS
            uint4 g0 = tb[tid], g1 = tb[256 + tid];
            atile[0][tid] = g0; atile[0][256 + tid] = g1;
            __syncthreads();
            for (uint32_t t = 0; t < nt_tiles; ++t) {
                const int cur = (int)(t & 1);
                if (t + 1 < nt_tiles) { g0 = tb[(size_t)(t + 1) * 512 + tid]; g1 = tb[(size_t)(t + 1) * 512 + 256 + tid]; }
                v16i acc0 = {0}, acc1 = {0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const uint4 x = atile[cur][ks * 64 + lane];
                    const v4i av = v4i{(int)x.x, (int)x.y, (int)x.z, (int)x.w};
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b0[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b1[ks], acc1, 0, 0, 0);
                }
                const int m0 = max16(acc0), m1 = max16(acc1);
                if (m0 > best0) { best0 = m0; bt0 = t; }        // strict: the FIRST tile keeps the title
                if (m1 > best1) { best1 = m1; bt1 = t; }
                if (t + 1 < nt_tiles) { atile[cur ^ 1][tid] = g0; atile[cur ^ 1][256 + tid] = g1; }
                __syncthreads();                                 // tile t + 1 visible; tile t's buffer free for t + 2
            }
        }
        // a query column's 32 stored rows of a tile sit in 16 registers of lane l and 16 of lane l + 32
        {
            const int o0 = __shfl_xor(best0, 32, 64), o1 = __shfl_xor(best1, 32, 64);
            const uint32_t p0 = (uint32_t)__shfl_xor((int)bt0, 32, 64), p1 = (uint32_t)__shfl_xor((int)bt1, 32, 64);
            if (o0 > best0 || (o0 == best0 && p0 < bt0)) { best0 = o0; bt0 = p0; }
            if (o1 > best1 || (o1 == best1 && p1 < bt1)) { best1 = o1; bt1 = p1; }
        }
        if (a.argmin) {
            // both half-waves now hold (best, first tile) of both query tiles: lanes 0-31 finish tile 0's rows, 32-63 tile 1's
            uint32_t* out = a.dist + (size_t)(it.out_offset + s - a.pair_base) * MAX_FUSED_QUERY_ROWS;
            const uint32_t r = qt0 * 32 + (uint32_t)lane;                        // = r0 for lane < 32, r1 for lane >= 32
            if (r < (uint32_t)nq) out[r] = nt > 0 ? rescan_tile(a, it.q_frame, r, slot, lane < 32 ? bt0 : bt1, nt) : 0xFFFFFFFFu;
            continue;
        }
        if (lane < 32) {
            uint32_t* out = a.dist + (size_t)(it.out_offset + s - a.pair_base) * MAX_FUSED_QUERY_ROWS;
            const uint32_t r0 = qt0 * 32 + (uint32_t)lane, r1 = r0 + 32;
            // <q, t> = 256 - 2 d  =>  d = (256 - dot) / 2, exact
            if (r0 < (uint32_t)nq) out[r0] = nt > 0 ? (uint32_t)(256 - best0) >> 1 : 0xFFFFFFFFu;
            if (r1 < (uint32_t)nq) out[r1] = nt > 0 ? (uint32_t)(256 - best1) >> 1 : 0xFFFFFFFFu;
        }
    }
}

hipError_t launch_score_mfma(const MfmaArgs& a, uint32_t n_items, hipStream_t st) {
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(k_score_mfma, dim3(n_items), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- the same search on the block-scaled fp4 matrix instruction (variant 5) -------------------------------------------
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands: +1 = 0x2, -1 = 0xA, every block scale 1.0 (E8M0 0x7F).
// 64 bits of a descriptor per instruction at the cycles the int8 form spends on 32, and half the operand bytes:
// a tile of 32 rows is 4 k-steps x 64 lanes x 16 bytes = 4 KiB.  Products are +-1 and sums stay below 2^24: the f32
// accumulator holds <q, t> exactly.  One workgroup = 512 query rows (each wave keeps 4 query tiles = 64 VGPRs).
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_expand_fp4(const uint32_t* rows, const int32_t* counts, uint32_t stride_words,
                                                    uint32_t tiles_per_frame, uint8_t* img) {
    const uint32_t tile = blockIdx.x, frame = blockIdx.y;
    const int n = counts[frame];
    if ((int)(tile * 32) >= n) return;
    const uint32_t ks = threadIdx.x >> 6, lane = threadIdx.x & 63;             // 4 k-steps of 64 bits
    const uint32_t r = min(tile * 32 + (lane & 31), (uint32_t)(n - 1));
    const uint32_t h = lane >> 5;
    const uint32_t word = rows[(size_t)frame * stride_words + (size_t)r * 8 + ks * 2 + h];   // bits [64 ks + 32 h, +32)
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v |= (((word >> (8 * i + b)) & 1u) ? 0x2u : 0xAu) << (4 * b);
        o[i] = v;
    }
    uint4* dst = reinterpret_cast<uint4*>(img + ((size_t)frame * tiles_per_frame + tile) * FP4_TILE_BYTES) + ks * 64 + lane;
    *dst = make_uint4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_expand_fp4(const uint32_t* rows, const int32_t* counts, uint32_t stride_words, uint32_t n_frames,
                             uint32_t tiles_per_frame, uint8_t* img, hipStream_t st) {
    if (n_frames == 0 || tiles_per_frame == 0) return hipSuccess;
    hipLaunchKernelGGL(k_expand_fp4, dim3(tiles_per_frame, n_frames), dim3(256), 0, st, rows, counts, stride_words, tiles_per_frame, img);
    return hipGetLastError();
}

__device__ __forceinline__ float max16f(const v16f& c) {
    float m = fmaxf(fmaxf(c[0], c[1]), c[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, c[i]), c[i + 1]);
    return fmaxf(m, c[15]);
}

__global__ __launch_bounds__(256, 3) void k_score_mfma_fp4(MfmaArgs a) {
    __shared__ uint4 atile[2][256];                             // two 4 KiB stored-frame tiles
    const MfmaItem it = a.items[blockIdx.x];
    if (it.n_slots == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = a.q_counts[it.q_frame];
    const uint32_t qt0 = it.q_chunk * 16 + (uint32_t)wave * 4;  // this wave's four query tiles (chunk = 512 rows)
    const uint32_t nq_tiles = (uint32_t)(nq + 31) / 32;
    constexpr int SCALE_ONE = 0x7F7F7F7F;                       // E8M0 127 = 2^0 in every byte

    v8i b[4][4];
    {
        const uint4* qb = reinterpret_cast<const uint4*>(a.q_pm1 + ((size_t)it.q_frame * a.q_tiles_per_frame + qt0) * FP4_TILE_BYTES);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 x = make_uint4(0, 0, 0, 0);
                if (qt0 + t < nq_tiles) x = qb[t * 256 + ks * 64 + lane];
                b[t][ks] = v8i{(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
            }
    }

    for (uint32_t s = 0; s < it.n_slots; ++s) {
        const uint32_t slot = it.slot_begin + s;
        const int nt = a.db_counts[slot];
        const uint32_t nt_tiles = (uint32_t)(nt + 31) / 32;
        const uint4* tb = reinterpret_cast<const uint4*>(a.db_pm1 + (size_t)slot * a.db_tiles_per_frame * FP4_TILE_BYTES);
        float best[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
        uint32_t bt[4] = {0, 0, 0, 0};                           // first tile that reached best[q] (argmin form)
        if (nt_tiles > 0) {
            uint4 g = tb[tid];
            atile[0][tid] = g;
            __syncthreads();
            for (uint32_t t = 0; t < nt_tiles; ++t) {
                const int cur = (int)(t & 1);
                if (t + 1 < nt_tiles) g = tb[(size_t)(t + 1) * 256 + tid];
                v16f acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint4 x = atile[cur][ks * 64 + lane];
                    const v8i av = v8i{(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b[q][ks], acc[q], 4, 4, 0, SCALE_ONE, 0, SCALE_ONE);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float mq = max16f(acc[q]);
                    if (mq > best[q]) { best[q] = mq; bt[q] = t; }      // strict: the FIRST tile keeps the title
                }
                if (t + 1 < nt_tiles) atile[cur ^ 1][tid] = g;
                __syncthreads();
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ob = __shfl_xor(best[q], 32, 64);
            const uint32_t ot = (uint32_t)__shfl_xor((int)bt[q], 32, 64);
            if (ob > best[q] || (ob == best[q] && ot < bt[q])) { best[q] = ob; bt[q] = ot; }
        }
        if (a.argmin) {
            // lanes 0-31 finish query tiles 0 and 1 of this wave, lanes 32-63 tiles 2 and 3
            uint32_t* out = a.dist + (size_t)(it.out_offset + s - a.pair_base) * MAX_FUSED_QUERY_ROWS;
            const uint32_t half = (uint32_t)lane >> 5, l31 = (uint32_t)lane & 31u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t tb_ = half ? bt[2 + k] : bt[k];
                const uint32_t r = (qt0 + 2u * half + (uint32_t)k) * 32 + l31;
                if (r < (uint32_t)nq) out[r] = nt > 0 ? rescan_tile(a, it.q_frame, r, slot, tb_, nt) : 0xFFFFFFFFu;
            }
            continue;
        }
        if (lane < 32) {
            uint32_t* out = a.dist + (size_t)(it.out_offset + s - a.pair_base) * MAX_FUSED_QUERY_ROWS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r = (qt0 + (uint32_t)q) * 32 + (uint32_t)lane;
                if (r < (uint32_t)nq) out[r] = nt > 0 ? (uint32_t)(256 - (int)best[q]) >> 1 : 0xFFFFFFFFu;
            }
        }
    }
}

hipError_t launch_score_mfma_fp4(const MfmaArgs& a, uint32_t n_items, hipStream_t st) {
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(k_score_mfma_fp4, dim3(n_items), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Which pair scratch slot `local` of a fold launch holds (FinalizeBulkArgs): global pair index p (where its record goes),
// stored slot, and the query frame's row count.
__device__ __forceinline__ void fold_locate(const FinalizeBulkArgs& a, uint32_t local, uint32_t& p, uint32_t& slot, int& nq) {
    if (a.pk_pairs) {                                           // packed route: group of query frames x range of stored slots
        uint32_t lo = 0, hi = a.pk_n;                           // last k with pk_pairs[k] <= local
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.pk_pairs[mid] <= local) lo = mid; else hi = mid;
        }
        const uint32_t c = a.pk_cidx[lo];
        slot = a.slot0 + (local - a.pk_pairs[lo]);
        p = a.offsets[c] + slot;
        nq = a.nq[c];
        return;
    }
    p = a.pair_base + local;
    uint32_t lo = 0, hi = a.n_q;                                // last c with offsets[c] <= p
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.offsets[mid] <= p) lo = mid; else hi = mid;
    }
    nq = a.nq[lo];
    slot = p - a.offsets[lo];
}

// ---- per-pair fold of the best distances ---------------------------------------------------------------------------
// One WAVE per pair: a pair's <= 2048 per-row words are read once, 8 coalesced 16-byte loads per lane (1 KiB per wave
// load), and stay in 32 registers for both passes (min-of-mins, then the ratio-filter count); the two wave reductions
// are shuffles.  HBM/L2-bound: 4 B read per (pair, query row) + 8 (12) B written per pair.
__global__ __launch_bounds__(256) void k_finalize_bulk(FinalizeBulkArgs a, uint32_t n_pairs) {
    const int lane = threadIdx.x & 63;
    const uint32_t local = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (local >= n_pairs) return;                               // whole wave: no barrier follows
    uint32_t p, slot;
    int nq;
    fold_locate(a, local, p, slot, nq);
    const uint32_t stride = a.stride ? a.stride : (uint32_t)MAX_FUSED_QUERY_ROWS;
    const uint4* d = reinterpret_cast<const uint4*>(a.dist + (size_t)local * stride);
    const int sh = a.key_shift;
    const uint32_t idx_mask = sh ? ((1u << sh) - 1u) : 0u;
    uint32_t dmin = 0xFFFFFFFFu, cnt = 0, isum = 0;
    if (nq <= MAX_FUSED_QUERY_ROWS) {
        uint32_t v[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = (i * 64 + lane) * 4;                 // rows r0 .. r0 + 3; rows >= nq hold stale words: masked
            uint4 x = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (r0 < nq) x = d[i * 64 + lane];
            v[4 * i + 0] = x.x;
            v[4 * i + 1] = r0 + 1 < nq ? x.y : 0xFFFFFFFFu;
            v[4 * i + 2] = r0 + 2 < nq ? x.z : 0xFFFFFFFFu;
            v[4 * i + 3] = r0 + 3 < nq ? x.w : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) dmin = min(dmin, v[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : v[k] >> sh);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const bool good = v[k] != 0xFFFFFFFFu && (v[k] >> sh) <= thr;
            cnt += good ? 1u : 0u;
            isum += good ? (v[k] & idx_mask) : 0u;
        }
    } else {
        // a query frame above 2048 rows (packed route only): two passes over its words, the second from the caches
        const uint32_t* w = a.dist + (size_t)local * stride;
        for (int r = lane; r < nq; r += 64) { const uint32_t x = w[r]; if (x != 0xFFFFFFFFu) dmin = min(dmin, x >> sh); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
        for (int r = lane; r < nq; r += 64) {
            const uint32_t x = w[r];
            const bool good = x != 0xFFFFFFFFu && (x >> sh) <= thr;
            cnt += good ? 1u : 0u;
            isum += good ? (x & idx_mask) : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { cnt += (uint32_t)__shfl_xor((int)cnt, o, 64); isum += (uint32_t)__shfl_xor((int)isum, o, 64); }
    if (lane == 0) {
        const int nt = a.db_counts[slot];
        const bool empty = (nq <= 0) || (nt <= 0) || dmin == 0xFFFFFFFFu;
        uint2 rec;
        rec.x = empty ? 0u : cnt;
        rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((uint32_t)(nt & 0xFFFF) << 16);
        reinterpret_cast<uint2*>(a.scores)[p] = rec;
        if (a.idx_sums) a.idx_sums[p] = empty ? 0u : isum;
    }
}

// The same fold over 2-byte words (the packed route's distance-only scratch: a best distance is <= 256, 0xFFFF = the
// stored frame is empty): 4 coalesced 16-byte loads per lane instead of 8, half the scratch traffic.
__global__ __launch_bounds__(256) void k_finalize_bulk_u16(FinalizeBulkArgs a, uint32_t n_pairs) {
    const int lane = threadIdx.x & 63;
    const uint32_t local = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (local >= n_pairs) return;                               // whole wave: no barrier follows
    uint32_t p, slot;
    int nq;
    fold_locate(a, local, p, slot, nq);
    const uint32_t stride = a.stride ? a.stride : (uint32_t)MAX_FUSED_QUERY_ROWS;
    const uint16_t* w = reinterpret_cast<const uint16_t*>(a.dist) + (size_t)local * stride;
    uint32_t dmin = 0xFFFFu, cnt = 0;
    if (nq <= MAX_FUSED_QUERY_ROWS) {
        uint32_t v[32];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r0 = (i * 64 + lane) * 8;                 // rows r0 .. r0 + 7; rows >= nq hold stale words: masked
            uint4 x = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (r0 < nq) x = reinterpret_cast<const uint4*>(w)[i * 64 + lane];
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[8 * i + 2 * k] = r0 + 2 * k < nq ? (xs[k] & 0xFFFFu) : 0xFFFFu;
                v[8 * i + 2 * k + 1] = r0 + 2 * k + 1 < nq ? (xs[k] >> 16) : 0xFFFFu;
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) dmin = min(dmin, v[k]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
#pragma unroll
        for (int k = 0; k < 32; ++k) cnt += (v[k] != 0xFFFFu && v[k] <= thr) ? 1u : 0u;
    } else {
        for (int r = lane; r < nq; r += 64) dmin = min(dmin, (uint32_t)w[r]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        const uint32_t thr = max((uint32_t)a.ratio * dmin, (uint32_t)a.dist_floor);
        for (int r = lane; r < nq; r += 64) { const uint32_t x = w[r]; cnt += (x != 0xFFFFu && x <= thr) ? 1u : 0u; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
    if (lane == 0) {
        const int nt = a.db_counts[slot];
        const bool empty = (nq <= 0) || (nt <= 0) || dmin == 0xFFFFu;
        uint2 rec;
        rec.x = empty ? 0u : cnt;
        rec.y = (empty ? 0xFFFFu : (dmin & 0xFFFFu)) | ((uint32_t)(nt & 0xFFFF) << 16);
        reinterpret_cast<uint2*>(a.scores)[p] = rec;
    }
}

hipError_t launch_finalize_bulk(const FinalizeBulkArgs& a, uint32_t n_pairs, hipStream_t st) {
    if (n_pairs == 0) return hipSuccess;
    if (a.word_bytes == 2) {
        if (a.key_shift != 0 || a.idx_sums) return hipErrorInvalidValue;      // packed keys are 4-byte words
        hipLaunchKernelGGL(k_finalize_bulk_u16, dim3((n_pairs + 3) / 4), dim3(256), 0, st, a, n_pairs);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_finalize_bulk, dim3((n_pairs + 3) / 4), dim3(256), 0, st, a, n_pairs);
    return hipGetLastError();
}

}  // namespace lcm
