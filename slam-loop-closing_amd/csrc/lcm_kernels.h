// lcm_kernels.h — launch interface between the C-ABI host code (lcm_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcm {

constexpr int KEY_SHIFT = 22;                       // packed key = dist << 22 | train_idx  (dist <= 256 -> 9 bits)
constexpr uint32_t KEY_IDX_MASK = (1u << KEY_SHIFT) - 1;
constexpr int MAX_FUSED_QUERY_ROWS = 2048;          // one workgroup holds a whole query frame in registers
constexpr int MAX_QUERY_BATCH = 16;                 // online queries scored by one launch (lcm_query_submit_batch)

// One unit of work: query frame `q_frame` against stored slots [slot_begin, slot_begin + n_slots),
// score records written to scores[out_offset ...].
struct WorkItem {
    uint32_t q_frame;
    uint32_t slot_begin;
    uint32_t n_slots;
    uint32_t out_offset;
};

// Pair mode (matchFeatures / match lists): one unit = <= 2048 query rows starting at row q_row of q_rows against nt
// train rows starting at row t_row of db_rows (a SEGMENT of a train matrix: several items share one pair so that a
// single 2000 x 2000 match spreads over ~64 workgroups).  t_row and every segment length but a matrix's last are
// multiples of 4; the last segment ends where the matrix's padding rows (copies of its last row) begin.
struct PairItem {
    uint32_t q_row;
    uint32_t t_row;
    uint32_t nq_nt;              // nq (12 bits, <= 2048 -> stored as nq) | nt << 12 (20 bits)
    uint32_t out_offset;         // keys[out_offset * keys_stride + local query row]
};

struct ScoreArgs {
    const uint32_t* q_rows;      // query frames: frame f at q_rows + f * q_stride_words, rows of 8 dwords
    const int32_t*  q_counts;    // rows per query frame
    uint32_t        q_stride_words;
    const uint32_t* db_rows;     // stored frames, same layout
    const int32_t*  db_counts;
    uint32_t        db_stride_words;
    const WorkItem* items;       // NULL => implicit items (online queries: nothing to upload), see imp_* below
    const PairItem* pair_items;  // non-NULL => pair mode: overrides items / implicit items; keys must be non-NULL
    void*           scores;      // lcm_score records (8 bytes each)
    uint32_t*       keys;        // optional: best packed key per query row, keys[pair * keys_stride + row]
    uint32_t        keys_stride;
    uint32_t*       idx_sums;    // optional (ARGMIN kernels): per pair, sum of the train indices of its GOOD matches mod 2^32
    int32_t         ratio;
    int32_t         dist_floor;
    // Implicit items (items == NULL): ONE query frame of imp_nq rows at q_rows, cut into imp_chunks chunks of
    // imp_chunk_rows rows (1 chunk = whole frame), against stored slots [0, imp_total) in runs of imp_spi slots.
    // Workgroup b: chunk c = b % imp_chunks, run g = b / imp_chunks, out_offset = g * imp_spi * imp_chunks + c.
    uint32_t        imp_chunks, imp_chunk_rows, imp_spi, imp_total;
    int32_t         imp_nq;
    // Implicit BATCH (items == NULL, imp_nbatch > 0): imp_nbatch query frames (micro-batched online queries), query b
    // of bat_nq[b] rows stored at chunk index b * imp_chunks of q_rows, scored against stored slots [0, bat_elig[b]);
    // its workgroups are [bat_wg[b], bat_wg[b + 1]) and its records start at pair index bat_pair[b].  imp_chunks,
    // imp_chunk_rows and imp_spi are shared by the whole batch.
    // PACKED bulk mode (launch_score_packed): the query rows of a GROUP of query frames form one virtual row space cut
    // into 2048-row workgroups, so a 2000-row frame no longer leaves 48 of 2048 lane slots idle.  A launch (chunk) covers
    // one group against one RANGE of stored slots [pk_slot0, ...): the scratch holds only that range's pairs, so its
    // size does not limit how many frames share the row space.  Work item: q_frame = workgroup index w in that space
    // (virtual rows [2048 w, 2048 w + 2048)), slots as usual (inside the chunk's range).  Per query frame k of the group
    // (pk_n of them): pk_vstart[k] = its first virtual row (pk_vstart[pk_n] = total), pk_qframe[k] = its index in
    // q_rows, pk_elig[k] = its eligible stored slots (global), pk_pairs[k] = its first pair in this chunk's pk_dist
    // (pk_pairs[pk_n] = the chunk's pair count).  Output: the best distance (as a uint16; ARGMIN: the packed key, a
    // uint32) of every (pair, query row) -> word (pk_pairs[k] + slot - pk_slot0) * pk_stride + row of pk_dist;
    // launch_finalize_bulk folds them.
    // Query frames may exceed 2048 rows here (they span columns); every other route stops at MAX_FUSED_QUERY_ROWS.
    const uint32_t* pk_vstart;
    const uint32_t* pk_qframe;
    const uint32_t* pk_elig;
    const uint32_t* pk_pairs;
    uint32_t*       pk_dist;
    uint32_t        pk_n;
    uint32_t        pk_col_rows;  // rows per column (workgroup): 2048 (8 rows per lane) or 1536 (6 rows per lane)
    uint32_t        pk_stride;    // words of pk_dist per pair: 2048, or the largest query frame rounded up when above that
    uint32_t        pk_slot0;     // first stored slot of the chunk's range
    uint32_t        imp_nbatch;
    uint32_t        bat_wg[MAX_QUERY_BATCH + 1];
    uint32_t        bat_pair[MAX_QUERY_BATCH + 1];
    int32_t         bat_nq[MAX_QUERY_BATCH];
    uint32_t        bat_elig[MAX_QUERY_BATCH];
};

// Launch the pair-scoring kernel over n_items work items.  max_query_rows = largest row count of any query
// frame referenced (<= MAX_FUSED_QUERY_ROWS).  variant: 0 = row-per-lane/scalar-broadcast (default).
hipError_t launch_score(const ScoreArgs& a, uint32_t n_items, int max_query_rows, bool write_keys, int variant,
                        hipStream_t st);
// PACKED bulk mode (see ScoreArgs::pk_*): 256-thread workgroups of 8 rows per lane; argmin = group keys + re-scan.
hipError_t launch_score_packed(const ScoreArgs& a, uint32_t n_items, bool argmin, hipStream_t st);

// Split mode (short databases): chunks of 256 * qpt query rows per workgroup write per-row best distances into
// a.keys; launch_finalize folds each pair's rows into its lcm_score record.
hipError_t launch_score_split(const ScoreArgs& a, uint32_t n_items, int qpt, hipStream_t st);
struct FinalizeArgs {
    const uint32_t* dist;        // best distance per (pair, row): dist[pair * padded_rows + row]
    uint32_t        padded_rows;
    int32_t         nq;          // real query rows (single query)
    const int32_t*  db_counts;   // stored row counts; pair p is slot slot_begin + p
    uint32_t        slot_begin;
    void*           scores;      // lcm_score per pair
    int32_t         ratio, dist_floor;
    // batch (n_batch > 0): pair p belongs to the query b with bat_pair[b] <= p < bat_pair[b + 1], has bat_nq[b] query
    // rows and is stored slot p - bat_pair[b]
    uint32_t        n_batch;
    uint32_t        bat_pair[MAX_QUERY_BATCH + 1];
    int32_t         bat_nq[MAX_QUERY_BATCH];
};
hipError_t launch_finalize(const FinalizeArgs& a, uint32_t n_pairs, hipStream_t st);

// Pair mode, second step: fold the per-segment keys of every pair (segment-local train indices) into one key per query
// row with GLOBAL train indices: final[out_row0 + r] = min over segments g of (seg_keys[item(c, g)][r mod 2048] + g *
// seg_rows), c = r / 2048.  min over (dist, global index) keys keeps the FIRST minimum across segments.
struct PairDesc {
    uint32_t first_item;         // item (c, g) of this pair is first_item + c * n_seg + g
    uint32_t n_seg;
    uint32_t seg_rows;
    uint32_t nq;
    uint32_t out_row0;           // first row of this pair in the folded key array
};
struct FoldArgs {
    const uint32_t* seg_keys;    // per item: chunk_rows keys
    uint32_t        chunk_rows;  // query rows per item: 2048 (throughput shape) or 512 (latency shape); 0 means 2048
    const PairDesc* pairs;
    uint32_t*       final_keys;
    uint32_t        n_pairs;
};
hipError_t launch_fold_pair_keys(const FoldArgs& a, uint32_t max_nq, hipStream_t st);
// bytes (rounded up to 16; both buffers must have that room, 16-byte aligned) from PINNED host memory to device memory, by
// a kernel on `st` instead of a DMA-engine copy (latency-critical small uploads)
hipError_t launch_upload(void* d_dst, const void* h_src_pinned, size_t bytes, hipStream_t st);
// Pair mode, first step, LATENCY shape: items of <= 512 query rows on 256-thread workgroups of 2 rows per lane.  One
// matchFeatures call is a few hundred thousand distances per wave whatever the shape, and a wave alone on its SIMD issues
// one VALU instruction per ~8 cycles: what shortens the call is MORE waves with less work each, not fewer instructions
// (2000 x 2000: 252 workgroups x 4 waves x 64 distances per lane instead of 63 x 4 x 256).
hipError_t launch_score_pairs_small(const ScoreArgs& a, uint32_t n_items, hipStream_t st);

// On-device loop test over a finished score array (BASELINE.json configs[3] "fused on-device filter + loop test"):
// pair p belongs to query frame c = upper_bound(offsets, p) - 1 and stored slot p - offsets[c]; a candidate is
// similarity = good / min(kp_q, kp_t) > sim_threshold (IEEE double, strict) and good >= min_matches.  Candidates
// are compacted in pair order (count per 256-pair block, prefix scan, emit), which is (current id, matched id)
// ascending: the host does not sort.
struct LoopTestArgs {
    const void*     scores;        // lcm_score records
    const uint32_t* offsets;       // n_q + 1 pair offsets per query frame
    const int32_t*  q_ids;         // per query frame
    const int32_t*  q_kp;          // keypoint count per query frame
    const int32_t*  db_ids;        // per stored slot
    const int32_t*  db_kp;
    void*           out;           // lcm_loop_candidate records (24 bytes)
    uint32_t*       counter;       // number of candidates found (may exceed cap: count only)
    uint32_t*       block_counts;  // scratch: ceil(n_pairs / 256) words (candidates per block, then their prefix)
    uint32_t        n_q, n_pairs, cap;
    int32_t         min_matches;
    double          sim_threshold;
};
// Two steps, so that the host can size the candidate buffer from the count (and refuse a too-small `cap` before any
// worst-case allocation): launch_loop_count leaves the number of candidates in *counter (and the per-block prefix in
// block_counts); launch_loop_emit then writes candidate k < cap to out[k].
hipError_t launch_loop_count(const LoopTestArgs& a, hipStream_t st);
hipError_t launch_loop_emit(const LoopTestArgs& a, hipStream_t st);

// Cross-check (lcm_params.cross_check, BFMatcher crossCheck = true): per pair, forward keys (every query row's first
// nearest train row) and backward keys (every train row's first nearest query row) -> the pair's score record.
//   mode 1 (mutual):  query q keeps (d, j) = fkey[q] iff bkey[j].idx == q
//   mode 2 (legacy):  query q keeps min over the train rows t with bkey[t].idx == q of (bkey[t].dist, t)
// then the usual min-of-mins / ratio filter / count over the queries that kept a match.
struct CrossDesc {
    uint32_t f_slot;             // forward keys at keys + f_slot * MAX_FUSED_QUERY_ROWS (nq <= 2048 entries)
    uint32_t b_slot;             // backward keys at keys + b_slot * MAX_FUSED_QUERY_ROWS (nt entries, contiguous)
    uint32_t nq, nt;
    uint32_t out;                // record index in scores
};
struct CrossArgs {
    const uint32_t*  keys;
    const CrossDesc* descs;
    void*            scores;
    uint32_t*        idx_sums;   // optional
    int32_t          mode, ratio, dist_floor;
};
hipError_t launch_cross_score(const CrossArgs& a, uint32_t n_pairs, hipStream_t st);

// Give every frame of a row matrix the padding the TRAIN role needs: rows [n, round_up(n, 4) + 4) of frame f become
// copies of its row n - 1 (n = counts[f]; nothing is written for n == 0).  stride_rows >= round_up(max n, 4) + 4.
hipError_t launch_pad_rows(uint32_t* rows, const int32_t* counts, uint32_t stride_rows, uint32_t n_frames, hipStream_t st);

// ---- OPT-IN matrix-core variant (lcm_set_kernel_variant(4)) ---------------------------------------------------------
// BASELINE.json's north_star rules MFMA out for the product path ("this path is integer bitwise work"), and variants
// 0 / 1 honour that.  Variant 4 exists to MEASURE what the rule costs: the same exact integers from
// v_mfma_i32_32x32x32_i8.  With every descriptor bit mapped to an int8 +1 / -1, <q, t> = 256 - 2 * hamming(q, t), so
// min distance = max dot product, exactly, in int32.
//
// Expanded operand image ("pm1"): per frame, per tile of 32 rows, per k-step of 32 bits: 64 lanes x 16 bytes =
// 1 KiB, lane (r = lane & 31, h = lane >> 5) holding bits [32 ks + 16 h, +16) of row 32 tile + r as 16 int8 — exactly
// one MFMA operand fragment per lane, so a wave's fragment load is one coalesced, conflict-free 1 KiB block.  Rows
// past a frame's end repeat its last row (cannot beat it).  8 KiB per tile = 8x the packed bytes.
constexpr int PM1_TILE_BYTES = 8192;
hipError_t launch_expand_pm1(const uint32_t* rows, const int32_t* counts, uint32_t stride_words, uint32_t n_frames,
                             uint32_t tiles_per_frame, uint8_t* pm1, hipStream_t st);

// One workgroup (4 waves) = 256 query rows (chunk q_chunk of query frame q_frame) against stored slots
// [slot_begin, slot_begin + n_slots): wave w keeps query tiles 2w, 2w+1 of the chunk in registers as B operands (64
// VGPRs), the stored frame's tiles stream through LDS as A operands (double-buffered, shared by the 4 waves), 16 MFMAs
// per tile per wave, running max per query in registers.  Output: best DISTANCE per (pair, query row) -> dist; the
// per-pair records come from launch_finalize_bulk (min-of-mins, ratio filter, count).
struct MfmaItem { uint32_t q_frame, q_chunk, slot_begin, n_slots, out_offset; };
struct MfmaArgs {
    const uint8_t*  q_pm1;  uint32_t q_tiles_per_frame;  const int32_t* q_counts;    // query frames (expanded)
    const uint8_t*  db_pm1; uint32_t db_tiles_per_frame; const int32_t* db_counts;   // stored frames (expanded)
    const MfmaItem* items;
    uint32_t*       dist;          // dist[(out_offset + s - pair_base) * 2048 + query row]
    uint32_t        pair_base;
    // ARGMIN form (argmin != 0): dist receives packed keys `distance << 22 | first train row attaining it`.  The matrix
    // instruction finds, per query row, the best dot product and the FIRST 32-row tile that reaches it; the lane then
    // re-scans that one tile on the vector ALU over the ORIGINAL packed rows (exact XOR + popcount) for the first row.
    int32_t         argmin;
    const uint32_t* q_rows;  uint32_t q_stride_words;    // packed query frames (frame f at q_rows + f * q_stride_words)
    const uint32_t* db_rows; uint32_t db_stride_words;   // packed stored frames
};
hipError_t launch_score_mfma(const MfmaArgs& a, uint32_t n_items, hipStream_t st);

// Variant 5: the same on v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands (+1 = 0x2, -1 = 0xA, scales 1.0): 4 KiB
// per tile of 32 rows (4 k-steps of 64 bits), one workgroup = 512 query rows (q_chunk counts 512-row chunks).
constexpr int FP4_TILE_BYTES = 4096;
hipError_t launch_expand_fp4(const uint32_t* rows, const int32_t* counts, uint32_t stride_words, uint32_t n_frames,
                             uint32_t tiles_per_frame, uint8_t* img, hipStream_t st);
hipError_t launch_score_mfma_fp4(const MfmaArgs& a, uint32_t n_items, hipStream_t st);

// Fold of per-row words into score records, one wave per pair.  Which pair a scratch slot l (0 <= l < n_pairs) is:
//   * pk_pairs == NULL (matrix-core variants): global pair p = pair_base + l belongs to query c = last c with
//     offsets[c] <= p and is stored slot p - offsets[c];
//   * pk_pairs != NULL (packed route, a chunk = group of query frames x range of stored slots): l belongs to the group's
//     frame k = last k with pk_pairs[k] <= l, is stored slot slot0 + l - pk_pairs[k], query c = pk_cidx[k], and its
//     record is scores[offsets[c] + slot].
// Folds dist[l * stride + r], r < nq[c].
struct FinalizeBulkArgs {
    uint32_t        stride;        // words of dist per pair (0 = MAX_FUSED_QUERY_ROWS); a multiple of 8
    uint32_t        word_bytes;    // 4 (0 means 4): dist holds uint32 words; 2: uint16 distances, 0xFFFF = none (packed
                                   // route, distance-only mode: halves the scratch traffic); keys are always 4 bytes
    int32_t         key_shift;     // 0: dist holds distances; KEY_SHIFT: packed keys dist << 22 | train row
    uint32_t*       idx_sums;      // optional (keys): per pair, sum of the good matches' train rows mod 2^32
    const uint32_t* dist;
    const uint32_t* offsets;       // n_q + 1
    const int32_t*  nq;            // n_q query row counts
    const int32_t*  db_counts;
    void*           scores;        // records at scores[p]
    uint32_t        n_q, pair_base;
    int32_t         ratio, dist_floor;
    const uint32_t* pk_pairs;      // packed route: pk_n + 1 local pair offsets of the group's frames in this chunk
    const uint32_t* pk_cidx;       // packed route: query index c of the group's frame k
    uint32_t        pk_n, slot0;
};
hipError_t launch_finalize_bulk(const FinalizeBulkArgs& a, uint32_t n_pairs, hipStream_t st);

// Multi-GPU merge (lcm_group_*): the W per-shard score arrays, gathered back to back on one device, are un-permuted
// into the single-device (query ascending, stored ascending) order.  Shard r's records are in (query ascending, owned
// stored ascending) order; query c's k-th record of shard r is stored position r + k * W, so it lands at
// offsets[c] + r + k * W.  HBM-bound: 8 bytes read + 8 bytes written per pair.
struct MergeArgs {
    const void*     gathered;      // all shards' lcm_score records, shard r at record index shard_base[r]
    void*           merged;        // output, n_total records
    const uint32_t* shard_offsets; // W arrays of n_q + 1 per-query offsets inside shard r (shard-local record index)
    const uint32_t* offsets;       // n_q + 1 per-query offsets of the merged array
    uint32_t        shard_base[9]; // W + 1 entries (W <= 8)
    uint32_t        world, n_q, n_total;
    uint32_t        elem_words;    // 2: lcm_score records (0 means 2); 1: the argmin search's per-pair index checksums
};
hipError_t launch_merge_shards(const MergeArgs& a, hipStream_t st);

}  // namespace lcm
