// lcm_internal.h — shared between the translation units of liblcm_hip.so's host side:
//   lcm_api.cpp        handle lifetime, parameters, the device database (append, snapshot), launch info, scratch helpers
//   lcm_pair.cpp       pair mode / match lists        lcm_online.cpp   online queries (single, micro-batch), detectLoops
//   lcm_bulk.cpp       bulk all-vs-all, fused loops   lcm_cross.cpp    cross_check scoring
//   lcm_mfma_host.cpp  opt-in matrix-core variants    lcm_group.cpp    multi-GPU group (RCCL)
// Not installed; the public surface is include/lcm.h.
#pragma once
#include "../../include/lcm.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lcm_kernels.h"

namespace lcm {
// last error message of the calling thread (lcm_last_error()); `fail` formats it and returns `code`
std::string& last_error();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const char* msg);   // for the host-class shim (lcs_host.cpp)
}  // namespace lcm

namespace {
using lcm::fail;

// Nothing may throw across the C boundary (include/lcm.h): every entry point that touches a std container runs inside
// this guard, which turns std::bad_alloc into LCM_ERR_OOM and anything else into LCM_ERR_HIP + message.
template <typename F>
int guarded(F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(LCM_ERR_OOM, "host allocation failed (std::bad_alloc)"); }
    catch (const std::exception& e) { return fail(LCM_ERR_HIP, "unexpected C++ exception: %s", e.what()); }
    catch (...) { return fail(LCM_ERR_HIP, "unexpected C++ exception"); }
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? LCM_ERR_OOM : LCM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

constexpr int ROW_PAD = 4;            // stored rows are padded to a multiple of 4 with copies of the last row
constexpr size_t ARENA_SLACK = 512;   // the kernel prefetches up to 4 rows past a frame's padded end
constexpr int DEFAULT_MAX_DESC = 2000;  // ORB nfeatures of the reference (README.md:114)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int padded_rows(int n) { return round_up(n, ROW_PAD); }   // rows [n, round_up(n,4)) of a stored frame repeat row n-1
inline uint64_t mix(uint64_t h, uint64_t v) { return h ^ (v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2)); }

}  // namespace

namespace lcm {

struct FrameMeta {
    int32_t id;
    int32_t n;       // descriptor rows
    int32_t n_kp;    // keypoints (similarity denominator)
};

constexpr int QUERY_SLOTS = 4;    // online queries that may be in flight at once (lcm_query_submit / _collect)
constexpr int STAGE_BUFS = 2;
// A micro-batch of fewer pairs than this is scored in split mode (1024-row chunks at the top of the range): with the
// slots on separate streams, finer workgroups let the next launch fill in behind the draining one.  Measured with
// bench.py --mode stream (8 frames per batch): 2500 frames 2.81e12 (limit 12288) -> 2.86e12.
constexpr size_t ONLINE_SPLIT_MAX_PAIRS = 65536;

struct QuerySlot {                // everything one in-flight online query owns
    bool busy = false;
    int n_elig = 0, nq = 0, query_id = 0;
    uint64_t db_generation = 0;   // h->db_generation at submit: a clear / load in between invalidates the ticket
    int n_batch = 0;              // > 0: a micro-batch (lcm_query_submit_batch); n_elig = records of all its queries
    int bat_elig[lcm::MAX_QUERY_BATCH] = {0};
    uint8_t* h_query = nullptr;   size_t h_query_bytes = 0;    // pinned staging of the query rows
    lcm_score* h_scores = nullptr; size_t h_scores_n = 0;      // pinned landing zone of the score records
    uint8_t* d_query = nullptr;   size_t d_query_bytes = 0;
    lcm_score* d_scores = nullptr; size_t d_scores_n = 0;
    uint32_t* d_dist = nullptr;   size_t d_dist_n = 0;         // split mode: best distance per (pair, row)
    uint8_t* h_meta = nullptr;    size_t h_meta_bytes = 0;     // matrix-core variants: pinned [row counts | work items]
    uint8_t* d_meta = nullptr;    size_t d_meta_bytes = 0;
    hipEvent_t done = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;      // around this query's kernel(s): summed into the handle's online stats
    // The slot's own stream: consecutive online queries run on different streams, so the upload and the first
    // workgroups of query k + 1 overlap the draining tail of query k's launch (a launch of a few thousand workgroups
    // leaves the chip partly idle while its last ones finish).  `fence` orders the slot stream after everything
    // enqueued on the handle's stream before the submit.
    hipStream_t stream = nullptr;
    hipEvent_t fence = nullptr;
    uint64_t acc_pairs = 0, acc_distances = 0, acc_bytes = 0;
    uint32_t acc_launches = 0, acc_queries = 0;
};

struct PlanSig {         // everything a bulk plan is built from: a cached plan is reused only when ALL of it is unchanged
    uint64_t db_generation = 0;
    size_t n_db = 0;     // stored frames (appends only add slots, a clear / truncate / load bumps db_generation)
    int n_q = 0, gap = 0, q_stride = 0, item_slots = 0, pack_mode = 0;
    bool self = false;
    size_t scratch_words = 0;
    std::vector<int32_t> q_ids, q_counts;      // external query set only
    std::vector<uint32_t> q_frame_of;
    bool operator==(const PlanSig& o) const {
        return db_generation == o.db_generation && n_db == o.n_db && n_q == o.n_q && gap == o.gap && q_stride == o.q_stride &&
               item_slots == o.item_slots && pack_mode == o.pack_mode && self == o.self && scratch_words == o.scratch_words &&
               q_ids == o.q_ids && q_counts == o.q_counts && q_frame_of == o.q_frame_of;
    }
};

struct Plan {            // cached work list of one bulk call shape
    uint64_t key = 0;    // 0 = invalid (set by everything that changes the database or the parameters); else sig is compared
    PlanSig sig;
    std::vector<lcm::WorkItem> items;
    std::vector<size_t> offsets;    // per query frame, start of its run of pairs (n_q + 1)
    lcm::WorkItem* d_items = nullptr;
    size_t d_items_cap = 0;
    size_t n_pairs = 0;
    uint64_t distances = 0, algo_bytes = 0;
    int max_q_rows = 0;
    // PACKED form (lcm_kernels.h, ScoreArgs::pk_*): a chunk = one GROUP of query frames (laid end to end in a virtual row
    // space) x one RANGE of stored slots; `items` holds (column, slot run) items chunk by chunk and
    // d_pk_tab = [pair offsets (n_q + 1) | query row counts (n_q) | per group: vstart (n_pos + 1), qframe, elig, cidx |
    //             per chunk: local pair offsets (n_pos + 1)]
    struct PackedChunk { uint32_t item0, n_items, tab0, ptab0, n_pos, slot0, n_pairs; };
    bool packed = false;
    std::vector<PackedChunk> pk_chunks;
    uint32_t* d_pk_tab = nullptr;
    size_t d_pk_tab_n = 0;
    size_t pk_max_pairs = 0;        // largest chunk: sizes the per-row scratch (8 KB per pair)
    uint32_t pk_n_q = 0;
    uint32_t pk_col_rows = 2048;    // rows per column = rows per workgroup of the packed score kernel
    uint32_t pk_stride = 2048;      // scratch words per pair
};

}  // namespace lcm

using lcm::FrameMeta;
using lcm::Plan;
using lcm::QuerySlot;

struct lcm_handle {
    lcm_params params;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;
    // packed bulk search of more than one chunk: the chunks alternate between `stream` and `stream2` (and between the two
    // halves of the per-row scratch), so that the draining tail of one chunk's launch is filled by the next chunk's workgroups
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t db_ready = nullptr;     // last append landed (recorded on copy_stream)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_aux_start = nullptr, ev_aux_stop = nullptr;   // the follow-up kernel of a call (k_loop_test, ...)
    bool aux_pending = false;
    // packed bulk search: one (start, stop) event pair around EVERY chunk's fold kernel; aux_kernel_ms is their sum
    std::vector<hipEvent_t> fold_ev;
    size_t fold_ev_used = 0;
    int variant = 0;
    int tune_item_slots = 0;           // 0 = automatic (pick_chunk)
    int tune_online_split = -1;        // -1 = automatic (enqueue_query)
    int tune_pair_upload_kernel = 1;   // pair mode, latency shape: staging block uploaded by a kernel (1) or by hipMemcpyAsync (0)
    int tune_pair_host_fold = 1;       // ... and the fold kernel writes into pinned host memory (1) or into device memory + a copy (0)
    int tune_online_streams = 1;       // 1 = every query slot runs on its own stream, 0 = all on the handle's stream
    // packed route: 4-byte words of per-row scratch per chunk.  _cfg is what LCM_TUNE_PACKED_SCRATCH_MB asked for (default
    // 1 GiB); the effective size is halved when the allocation fails and goes back to _cfg at the next plan rebuild.
    size_t pk_scratch_cfg_words = (size_t)1 << 28;
    size_t pk_scratch_words = (size_t)1 << 28;
    bool pk_oom_retry = false;
    int tune_packed = -1;              // -1 = automatic (bulk plan: when it saves lane slots), 0 = never, 1 = always

    // database arena
    uint8_t* d_rows = nullptr;
    int32_t* d_counts = nullptr;
    int cap_frames = 0;
    int stride_rows = 0;               // rows per frame slot (multiple of ROW_PAD)
    std::vector<FrameMeta> frames;
    bool pending_copy = false;
    bool db_ready_recorded = false;    // db_ready has been recorded at least once (slot streams wait on it)

    // pinned staging ring for streaming appends
    uint8_t* h_stage[lcm::STAGE_BUFS] = {nullptr, nullptr};
    size_t h_stage_bytes = 0;
    hipEvent_t stage_done[lcm::STAGE_BUFS] = {nullptr, nullptr};
    int32_t* h_counts = nullptr;       // pinned mirror of d_counts (source of the 4-byte async copies)
    int h_counts_cap = 0;
    int stage_next = 0;

    // scratch for query uploads / pair mode / results
    uint32_t* d_keys = nullptr; size_t d_keys_n = 0;
    uint8_t* h_pair_stage = nullptr; size_t h_pair_stage_bytes = 0;   // pair mode: pinned [rows | items | descriptors]
    uint8_t* d_pair_stage = nullptr; size_t d_pair_stage_bytes = 0;
    uint32_t* h_final_keys = nullptr; size_t h_final_keys_n = 0;      // pair mode: pinned landing zone of the folded keys
    uint8_t* d_xq = nullptr; size_t d_xq_bytes = 0;                   // cross_check: padded copy of an external query set
    // opt-in MFMA variant (4): +1 / -1 int8 operand images of the database and of an external query set, scratch
    uint8_t* d_pm1 = nullptr; size_t d_pm1_bytes = 0; uint64_t pm1_stamp = 0; size_t pm1_frames = 0;   // frames expanded so far
    uint8_t* d_qpm1 = nullptr; size_t d_qpm1_bytes = 0;
    uint32_t* d_mdist = nullptr; size_t d_mdist_n = 0;
    uint8_t* d_mitems = nullptr; size_t d_mitems_bytes = 0;
    uint32_t* d_mmeta = nullptr; size_t d_mmeta_n = 0;
    lcm_score* d_bulk_scores = nullptr; size_t d_bulk_scores_n = 0;   // lcm_all_vs_all_loops: scores stay on the device
    size_t bulk_scores_valid = 0;                                     // records of the last fused call still in there
    int32_t* d_meta = nullptr; size_t d_meta_n = 0;
    lcm_loop_candidate* d_cands = nullptr; size_t d_cands_n = 0;

    QuerySlot qslots[lcm::QUERY_SLOTS];
    lcm_online_stats online{};         // totals over collected online queries (lcm_online_stats_read)
    uint64_t db_generation = 1;        // bumped whenever stored frames are dropped (lcm_db_clear / lcm_db_load)
    Plan plan;
    lcm_launch_info info{};
    bool info_pending = false;
};


namespace {

template <typename T>
int ensure_dev(T*& p, size_t& have, size_t need, size_t slack_bytes = 0) {
    if (need <= have && p) return LCM_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr; have = 0;
    size_t n = std::max<size_t>(need, 16);
    HIP_TRY(hipMalloc((void**)&p, n * sizeof(T) + slack_bytes));
    have = n;
    return LCM_OK;
}

template <typename T>
int ensure_pinned(T*& p, size_t& have, size_t need) {
    if (need <= have && p) return LCM_OK;
    if (p) HIP_TRY(hipHostFree(p));
    p = nullptr; have = 0;
    const size_t n = std::max<size_t>(need, 16);
    HIP_TRY(hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault));
    have = n;
    return LCM_OK;
}

}  // namespace

namespace lcm {
// ---- lcm_api.cpp
int set_device(const lcm_handle* h);
int wait_db(lcm_handle* h);            // make the match stream see every append issued so far
int sync_online_streams(lcm_handle* h);   // host-wait for every query slot's own stream
int eligible_prefix(const lcm_handle* h, int query_id, int gap);   // eligible stored slots are a prefix: its length
int pick_chunk(const lcm_handle* h, size_t total_pairs);
int launch_and_time(lcm_handle* h, const ScoreArgs& a, uint32_t n_items, int max_q_rows, bool write_keys);
// snapshot file helpers (lcm_db_save / _load and their group forms share ONE format)
int snapshot_open(const char* path, std::vector<FrameMeta>& metas, uint32_t* real_max_rows, FILE** f_out);
bool snapshot_write_header(FILE* f, const std::vector<FrameMeta>& metas);
// ---- lcm_cross.cpp: cross_check scoring of "query c against stored slots [0, elig[c])", records in (query, slot) order
int cross_score_prefixes(lcm_handle* h, const uint8_t* d_qbase, const uint32_t* q_row0, const int* nq, const int* elig,
                         int n_q, lcm_score* d_scores, uint32_t* d_idx_sums);
// ---- lcm_mfma_host.cpp: opt-in matrix-core variants 4 / 5
int mfma_online(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int pitch_rows, int B, const int* nq, const int* elig);
int mfma_bulk(lcm_handle* h, bool self, const uint8_t* q_rows, const int32_t* d_q_counts, uint32_t q_pitch_rows,
              const uint32_t* q_frame_of, const int* nqv, int n_q, const std::vector<size_t>& offsets, lcm_score* d_scores,
              uint32_t* d_idx_sums);     // d_idx_sums non-NULL: the argmin form (keys + index checksums)
// ---- lcm_bulk.cpp
// Bulk search behind lcm_all_vs_all / lcm_all_vs_all_argmin.  q_frame_of (optional, n_q_frames entries): query frame c
// lives at index q_frame_of[c] of d_query_rows / d_query_counts instead of index c (the group's rank-major gathered
// query buffer).  h_query_counts (optional, host, indexed by c) spares the device read of the row counts.
// d_idx_sums non-NULL selects the argmin kernel.
int all_vs_all(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts, const int32_t* q_ids,
               int n_q_frames, int q_stride_rows, void* d_scores, size_t scores_cap, size_t* n_pairs,
               size_t* pair_offsets, uint32_t* d_idx_sums, const uint32_t* q_frame_of, const int32_t* h_query_counts);
// Loop test + ordered compaction over a score array on h's device (lcm_all_vs_all_loops; a group's shard over its own records)
int loop_test_device(lcm_handle* h, const void* d_scores, size_t n_pairs, const uint32_t* offsets, int n_q,
                     const int32_t* q_ids, const int32_t* q_kp, int n_db, const int32_t* db_ids, const int32_t* db_kp,
                     size_t cap, size_t* n_found);
}  // namespace lcm

namespace {
using lcm::cross_score_prefixes;
using lcm::eligible_prefix;
using lcm::launch_and_time;
using lcm::mfma_bulk;
using lcm::mfma_online;
using lcm::pick_chunk;
using lcm::QUERY_SLOTS;
using lcm::set_device;
using lcm::sync_online_streams;
using lcm::STAGE_BUFS;
using lcm::wait_db;
}  // namespace
