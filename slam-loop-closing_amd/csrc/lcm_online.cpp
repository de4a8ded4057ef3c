// lcm_online.cpp — the online path: one query frame or a micro-batch against the stored database (async tickets, split mode), detectLoops.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

// Split-mode codes (LCM_TUNE_ONLINE_SPLIT): 1 / 2 / 4 = 256-thread workgroups holding that many query rows per lane
// (256- / 512- / 1024-row chunks); 16 / 32 = 64- / 128-thread workgroups of 8 rows per lane (512- / 1024-row chunks: the
// bulk kernel's per-wave shape at a finer workgroup grain).
static inline bool is_split(int qpt) { return qpt == 1 || qpt == 2 || qpt == 4 || qpt == 16 || qpt == 32; }
static inline int split_chunk_rows(int qpt) { return qpt == 16 ? 512 : qpt == 32 ? 1024 : 256 * qpt; }

// The ONE place that decides whether an online launch is cut into row chunks (split mode), for single queries and
// micro-batches, at submit time (staging pitch) and at enqueue time alike: 0 = one workgroup per (query, run of stored
// frames), else the split code.  Only kernel variant 0 has a split form (the A/B variants 1-3 run unsplit, so that their
// online numbers are their own); frames of <= 512 rows fill too few lanes to be worth cutting.
// Measured (bench.py --mode stream): finer pieces balance 256 CUs better whenever a launch holds only a few thousand
// pairs — single query, 1000 frames: 1.71e12 unsplit -> 2.18e12; micro-batches of 8: split below 65536 pairs.
static int pick_online_split(const lcm_handle* h, int max_nq, size_t pairs, bool batch) {
    if (h->variant != 0 || max_nq <= 512) return 0;
    if (h->tune_online_split >= 0) return h->tune_online_split;     // lcm_set_tuning(LCM_TUNE_ONLINE_SPLIT)
    // Round 3, after the inner loop got 49 % faster (tools/online_split_sweep.py, profiles/r03_online_split_sweep.txt):
    // 2 query rows per lane (512-row chunks, 8 waves per SIMD) is the best or within 1 % of it at every size from 48
    // pairs up — an unsplit launch cannot finish before ONE workgroup has scanned a whole stored frame at 8 rows per
    // lane (400 us), the 2-row pieces take 140 us — and 1 row per lane only wins below ~48 pairs (100 us).
    if (batch) return pairs < lcm::ONLINE_SPLIT_MAX_PAIRS ? 2 : 0;
    return pairs < 48 ? 1 : pairs < 4096 ? 2 : 0;
}

extern "C" {

// scores of (query frame at q_rows[q_frame]) against stored slots [0, n_elig)
static void account_prefix(lcm_handle* h, int nq, int n_elig) {
    uint64_t dist = 0, bytes = (uint64_t)nq * 32;
    for (int s = 0; s < n_elig; ++s) { dist += (uint64_t)nq * h->frames[s].n; bytes += (uint64_t)h->frames[s].n * 32 + 8; }
    h->info.pairs = (uint64_t)n_elig; h->info.distances = dist; h->info.algo_bytes = bytes;
}

// Enqueue (no host synchronisation) the scoring of ONE query frame — `nq` rows at device address d_q — against stored
// slots [0, n_elig), and the download of the n_elig score records into the slot's pinned buffer.  Work items are
// implicit (derived from blockIdx), so nothing but the query itself crosses PCIe.  Short databases use the split
// mode (lcm_kernels.hip): 2 / 4 / 8 workgroups per pair + the on-device fold.
static int enqueue_query(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int nq, int n_elig, hipStream_t S) {
    q.n_elig = n_elig; q.nq = nq; q.n_batch = 0;
    q.acc_pairs = q.acc_distances = q.acc_bytes = 0; q.acc_launches = 0; q.acc_queries = 1;
    if (n_elig <= 0) { HIP_TRY(hipEventRecord(q.done, S)); return LCM_OK; }
    int rc = wait_db(h); if (rc) return rc;
    if (h->params.cross_check) {
        // both directions + the on-device mutual test; the caller has padded the query rows for the train role
        rc = ensure_dev(q.d_scores, q.d_scores_n, (size_t)n_elig); if (rc) return rc;
        rc = ensure_pinned(q.h_scores, q.h_scores_n, (size_t)n_elig); if (rc) return rc;
        const uint32_t row0 = 0;
        HIP_TRY(hipEventRecord(q.k0, h->stream));
        rc = cross_score_prefixes(h, (const uint8_t*)d_q, &row0, &nq, &n_elig, 1, q.d_scores, nullptr); if (rc) return rc;
        HIP_TRY(hipEventRecord(q.k1, h->stream));
        q.acc_pairs = h->info.pairs; q.acc_distances = h->info.distances; q.acc_bytes = h->info.algo_bytes; q.acc_launches = h->info.launches;
        HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * (size_t)n_elig, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(q.done, h->stream));
        return LCM_OK;
    }
    if (h->variant >= 4) {                                   // opt-in: the same records from the matrix cores
        // (the query is padded to 2048 rows in its own image; d_q may be a stored frame's rows or the staged query)
        return mfma_online(h, q, d_q, std::max(nq, 1), 1, &nq, &n_elig);
    }
    const int qpt = pick_online_split(h, nq, (size_t)n_elig, false);
    rc = ensure_dev(q.d_scores, q.d_scores_n, (size_t)n_elig); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, (size_t)n_elig); if (rc) return rc;
    lcm::ScoreArgs a{};
    a.q_rows = d_q; a.q_counts = nullptr; a.items = nullptr;
    a.db_rows = (const uint32_t*)h->d_rows; a.db_counts = h->d_counts; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    a.imp_nq = nq; a.imp_total = (uint32_t)n_elig;
    HIP_TRY(hipEventRecord(h->ev_start, S));
    HIP_TRY(hipEventRecord(q.k0, S));
    if (is_split(qpt)) {
        const int chunk_rows = split_chunk_rows(qpt);
        const int n_chunks = (nq + chunk_rows - 1) / chunk_rows;
        const size_t n_items = (size_t)n_elig * n_chunks;
        rc = ensure_dev(q.d_dist, q.d_dist_n, n_items * chunk_rows); if (rc) return rc;
        a.q_stride_words = (uint32_t)chunk_rows * LCM_DESC_WORDS;
        a.imp_chunks = (uint32_t)n_chunks; a.imp_chunk_rows = (uint32_t)chunk_rows; a.imp_spi = 1;
        a.scores = nullptr /* split mode writes no per-chunk records */; a.keys = q.d_dist; a.keys_stride = (uint32_t)chunk_rows;
        hipError_t e = lcm::launch_score_split(a, (uint32_t)n_items, qpt, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeArgs f{};
        f.dist = q.d_dist; f.padded_rows = (uint32_t)(n_chunks * chunk_rows); f.nq = nq;
        f.db_counts = h->d_counts; f.slot_begin = 0; f.scores = q.d_scores;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        e = lcm::launch_finalize(f, (uint32_t)n_elig, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
        h->info.launches = 2; h->info.workgroups = (uint32_t)n_items; h->info.route = LCM_ROUTE_SPLIT;
    } else {
        const int spi = n_elig >= 8192 ? 4 : (n_elig >= 4096 ? 2 : 1);
        const uint32_t n_items = (uint32_t)((n_elig + spi - 1) / spi);
        a.q_stride_words = 0;
        a.imp_chunks = 1; a.imp_chunk_rows = (uint32_t)std::max(nq, 1); a.imp_spi = (uint32_t)spi;
        a.scores = q.d_scores; a.keys = nullptr; a.keys_stride = 0;
        hipError_t e = lcm::launch_score(a, n_items, nq, false, h->variant, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        h->info.launches = 1; h->info.workgroups = n_items; h->info.route = LCM_ROUTE_PLAIN;
    }
    HIP_TRY(hipEventRecord(h->ev_stop, S));
    HIP_TRY(hipEventRecord(q.k1, S));
    h->info_pending = true;
    account_prefix(h, nq, n_elig);
    q.acc_pairs = h->info.pairs; q.acc_distances = h->info.distances; q.acc_bytes = h->info.algo_bytes;
    q.acc_launches = h->info.launches; q.acc_queries = 1;
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * (size_t)n_elig, hipMemcpyDeviceToHost, S));
    HIP_TRY(hipEventRecord(q.done, S));
    return LCM_OK;
}

// Micro-batch of online queries: B query frames (already in device memory at d_q, query b at row b * rows_per_query)
// against stored slots [0, elig[b]) each, ONE score launch (+ one finalize launch in split mode), one download.
// A launch of B x n_elig pairs fills the chip where a single query's few hundred pairs leave its tail idle, and the
// host pays one submit / collect round trip per B frames.
static int enqueue_batch(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int rows_per_query, int B, const int* nq, const int* elig, hipStream_t S) {
    size_t total = 0;
    int max_nq = 0;
    for (int b = 0; b < B; ++b) { total += (size_t)elig[b]; max_nq = std::max(max_nq, nq[b]); q.bat_elig[b] = elig[b]; }
    q.n_batch = B; q.n_elig = (int)total; q.nq = max_nq;
    q.acc_pairs = q.acc_distances = q.acc_bytes = 0; q.acc_launches = 0; q.acc_queries = (uint32_t)B;
    if (total == 0) { HIP_TRY(hipEventRecord(q.done, S)); return LCM_OK; }
    if (total > 0x7FFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^31 pairs in one batch");
    int rc = wait_db(h); if (rc) return rc;
    if (h->params.cross_check) {
        rc = ensure_dev(q.d_scores, q.d_scores_n, total); if (rc) return rc;
        rc = ensure_pinned(q.h_scores, q.h_scores_n, total); if (rc) return rc;
        uint32_t row0[lcm::MAX_QUERY_BATCH];
        for (int b = 0; b < B; ++b) row0[b] = (uint32_t)(b * rows_per_query);
        HIP_TRY(hipEventRecord(q.k0, h->stream));
        rc = cross_score_prefixes(h, (const uint8_t*)d_q, row0, nq, elig, B, q.d_scores, nullptr); if (rc) return rc;
        HIP_TRY(hipEventRecord(q.k1, h->stream));
        q.acc_pairs = h->info.pairs; q.acc_distances = h->info.distances; q.acc_bytes = h->info.algo_bytes; q.acc_launches = h->info.launches;
        HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * total, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(q.done, h->stream));
        return LCM_OK;
    }
    if (h->variant >= 4) return mfma_online(h, q, d_q, rows_per_query, B, nq, elig);
    const int qpt = pick_online_split(h, max_nq, total, true);
    rc = ensure_dev(q.d_scores, q.d_scores_n, total); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, total); if (rc) return rc;
    lcm::ScoreArgs a{};
    a.q_rows = d_q; a.q_counts = nullptr; a.items = nullptr;
    a.db_rows = (const uint32_t*)h->d_rows; a.db_counts = h->d_counts; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    a.imp_nbatch = (uint32_t)B;
    const bool split = is_split(qpt);
    const int chunk_rows = split ? split_chunk_rows(qpt) : rows_per_query;
    const int n_chunks = split ? (max_nq + chunk_rows - 1) / chunk_rows : 1;
    if (rows_per_query % chunk_rows != 0 || n_chunks * chunk_rows > rows_per_query)
        return fail(LCM_ERR_INVALID_ARG, "batch staging pitch %d does not fit %d chunks of %d rows", rows_per_query, n_chunks, chunk_rows);
    const int spi = split ? 1 : (total >= 16384 ? 4 : (total >= 8192 ? 2 : 1));
    // every query occupies rows_per_query rows of the staging buffer = rows_per_query / chunk_rows chunk slots, of
    // which the first n_chunks are scored
    const uint32_t chunk_slots = (uint32_t)(rows_per_query / chunk_rows);
    a.q_stride_words = (uint32_t)chunk_rows * LCM_DESC_WORDS;
    a.imp_chunks = (uint32_t)n_chunks; a.imp_chunk_rows = (uint32_t)chunk_rows; a.imp_spi = (uint32_t)spi;
    uint32_t wg = 0, pair = 0;
    for (int b = 0; b < B; ++b) {
        a.bat_wg[b] = wg; a.bat_pair[b] = pair;
        a.bat_nq[b] = nq[b]; a.bat_elig[b] = (uint32_t)elig[b];
        wg += (uint32_t)((elig[b] + spi - 1) / spi) * (uint32_t)n_chunks;
        pair += (uint32_t)elig[b];
    }
    a.bat_wg[B] = wg; a.bat_pair[B] = pair;
    // chunk index of query b's chunk c is b * imp_chunks + c in the kernel; with a pitch of chunk_slots chunks per query
    // that only holds when imp_chunks == chunk_slots: the staging copy below packs the queries at that pitch
    if ((uint32_t)n_chunks != chunk_slots) return fail(LCM_ERR_HIP, "internal: batch pitch %u != %d chunks", chunk_slots, n_chunks);
    HIP_TRY(hipEventRecord(h->ev_start, S));
    HIP_TRY(hipEventRecord(q.k0, S));
    if (split) {
        rc = ensure_dev(q.d_dist, q.d_dist_n, total * (size_t)n_chunks * chunk_rows); if (rc) return rc;
        a.scores = nullptr; a.keys = q.d_dist; a.keys_stride = (uint32_t)chunk_rows;
        hipError_t e = lcm::launch_score_split(a, wg, qpt, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeArgs f{};
        f.dist = q.d_dist; f.padded_rows = (uint32_t)(n_chunks * chunk_rows); f.nq = 0;
        f.db_counts = h->d_counts; f.slot_begin = 0; f.scores = q.d_scores;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        f.n_batch = (uint32_t)B;
        for (int b = 0; b <= B; ++b) f.bat_pair[b] = a.bat_pair[b];
        for (int b = 0; b < B; ++b) f.bat_nq[b] = nq[b];
        e = lcm::launch_finalize(f, pair, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
        h->info.launches = 2; h->info.route = LCM_ROUTE_SPLIT;
    } else {
        a.scores = q.d_scores; a.keys = nullptr; a.keys_stride = 0;
        // (the train-row-per-lane A/B kernels, variants 2 / 3, have no micro-batch form: a batch under them runs variant 0)
        hipError_t e = lcm::launch_score(a, wg, max_nq, false, h->variant >= 2 ? 0 : h->variant, S);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        h->info.launches = 1; h->info.route = LCM_ROUTE_PLAIN;
    }
    h->info.workgroups = wg;
    HIP_TRY(hipEventRecord(h->ev_stop, S));
    HIP_TRY(hipEventRecord(q.k1, S));
    h->info_pending = true;
    {
        uint64_t dist = 0, bytes = 0, prows = 0;
        int s = 0;
        std::vector<std::pair<int, int>> order((size_t)B);      // queries by eligibility, to walk the prefix sums once
        for (int b = 0; b < B; ++b) order[(size_t)b] = {elig[b], b};
        std::sort(order.begin(), order.end());
        for (auto [e, b] : order) {
            for (; s < e; ++s) prows += (uint64_t)h->frames[(size_t)s].n;
            dist += (uint64_t)nq[b] * prows;
            bytes += prows * 32 + (uint64_t)nq[b] * 32 + 8ull * (uint64_t)e;
        }
        h->info.pairs = total; h->info.distances = dist; h->info.algo_bytes = bytes;
        q.acc_pairs = total; q.acc_distances = dist; q.acc_bytes = bytes; q.acc_launches = h->info.launches;
    }
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * total, hipMemcpyDeviceToHost, S));
    HIP_TRY(hipEventRecord(q.done, S));
    return LCM_OK;
}

// Query frames above 2048 rows (ORB with nfeatures > 2048) do not fit one workgroup's registers: they go through the
// bulk search's packed route (lcm_bulk.cpp) as an external query set of B frames — same records, same (query, slot)
// order — on the handle's stream, with the bulk plan and scratch.  Throughput path of a rare shape, not a fast one.
static int enqueue_big(lcm_handle* h, QuerySlot& q, const void* d_q, const int32_t* d_counts, int stride_rows, int B,
                       const int* nq, const int* ids, const int* elig) {
    size_t total = 0;
    int max_nq = 0;
    for (int b = 0; b < B; ++b) { total += (size_t)elig[b]; max_nq = std::max(max_nq, nq[b]); q.bat_elig[b] = elig[b]; }
    q.n_elig = (int)total; q.nq = max_nq;
    q.acc_pairs = q.acc_distances = q.acc_bytes = 0; q.acc_launches = 0; q.acc_queries = (uint32_t)B;
    if (total == 0) { HIP_TRY(hipEventRecord(q.done, h->stream)); return LCM_OK; }
    if (total > 0x7FFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^31 pairs in one batch");
    int rc = ensure_dev(q.d_scores, q.d_scores_n, total); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, total); if (rc) return rc;
    HIP_TRY(hipEventRecord(q.k0, h->stream));
    size_t n = 0;
    rc = lcm::all_vs_all(h, d_q, d_counts, ids, B, stride_rows, q.d_scores, total, &n, nullptr, nullptr, nullptr, nq);
    if (rc) return rc;
    if (n != total) return fail(LCM_ERR_HIP, "internal: bulk route scored %zu pairs, the online path expected %zu", n, total);
    HIP_TRY(hipEventRecord(q.k1, h->stream));
    q.acc_pairs = h->info.pairs; q.acc_distances = h->info.distances; q.acc_bytes = h->info.algo_bytes; q.acc_launches = h->info.launches;
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(q.done, h->stream));
    return LCM_OK;
}

// The per-query row counts of a big-frame submit, on the device (the bulk search reads them there)
static int upload_counts(lcm_handle* h, QuerySlot& q, const int* nq, int B) {
    int rc = ensure_pinned(q.h_meta, q.h_meta_bytes, sizeof(int32_t) * (size_t)lcm::MAX_QUERY_BATCH); if (rc) return rc;
    rc = ensure_dev(q.d_meta, q.d_meta_bytes, sizeof(int32_t) * (size_t)lcm::MAX_QUERY_BATCH); if (rc) return rc;
    memcpy(q.h_meta, nq, sizeof(int32_t) * (size_t)B);
    HIP_TRY(hipMemcpyAsync(q.d_meta, q.h_meta, sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    return LCM_OK;
}

// A finished query (its `done` event has been waited for) joins the handle's online totals.
static void fold_online_stats(lcm_handle* h, QuerySlot& q) {
    if (q.acc_launches) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, q.k0, q.k1) == hipSuccess) h->online.kernel_ms += ms; else (void)hipGetLastError();
    }
    h->online.launches += q.acc_launches; h->online.queries += q.acc_queries;
    h->online.pairs += q.acc_pairs; h->online.distances += q.acc_distances; h->online.algo_bytes += q.acc_bytes;
    q.acc_launches = 0; q.acc_queries = 0; q.acc_pairs = q.acc_distances = q.acc_bytes = 0;
}

static int find_slot(const lcm_handle* h, int frame_id) {
    int lo = 0, hi = (int)h->frames.size();
    while (lo < hi) { int mid = (lo + hi) / 2; if (h->frames[mid].id < frame_id) lo = mid + 1; else hi = mid; }
    return (lo < (int)h->frames.size() && h->frames[lo].id == frame_id) ? lo : -1;
}

static int acquire_query_slot(lcm_handle* h, int* ticket) {
    for (int i = 0; i < QUERY_SLOTS; ++i)
        if (!h->qslots[i].busy) {
            if (!h->qslots[i].done) HIP_TRY(hipEventCreateWithFlags(&h->qslots[i].done, hipEventDisableTiming));
            if (!h->qslots[i].k0) HIP_TRY(hipEventCreate(&h->qslots[i].k0));
            if (!h->qslots[i].k1) HIP_TRY(hipEventCreate(&h->qslots[i].k1));
            if (!h->qslots[i].fence) HIP_TRY(hipEventCreateWithFlags(&h->qslots[i].fence, hipEventDisableTiming));
            if (!h->qslots[i].stream) HIP_TRY(hipStreamCreateWithFlags(&h->qslots[i].stream, hipStreamNonBlocking));
            *ticket = i;
            return LCM_OK;
        }
    return fail(LCM_ERR_CAPACITY, "%d queries already in flight: collect one first", QUERY_SLOTS);
}

// The stream an online query runs on: the slot's own (QuerySlot::stream) for the vector-ALU kernels, ordered after
// everything already enqueued on the handle's stream and after every append issued so far; the handle's stream for
// the routes that use handle-wide scratch (cross_check, matrix-core variants) or when LCM_TUNE_ONLINE_STREAMS is 0.
static int pick_stream(lcm_handle* h, QuerySlot& q, hipStream_t* S) {
    *S = h->stream;
    if (h->params.cross_check || h->variant >= 4 || !h->tune_online_streams) return LCM_OK;
    HIP_TRY(hipEventRecord(q.fence, h->stream));
    HIP_TRY(hipStreamWaitEvent(q.stream, q.fence, 0));
    if (h->db_ready_recorded) HIP_TRY(hipStreamWaitEvent(q.stream, h->db_ready, 0));
    *S = q.stream;
    return LCM_OK;
}

static int query_submit_impl(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, int* ticket) {
    if (!h || nq < 0 || !ticket || (nq > 0 && !query)) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *ticket = -1;
    if (nq > 65535) return fail(LCM_ERR_CAPACITY, "a frame may hold at most 65535 rows (got %d)", nq);
    const bool big = nq > lcm::MAX_FUSED_QUERY_ROWS;              // served by the bulk search's packed route (enqueue_big)
    int rc = set_device(h); if (rc) return rc;
    int t = -1;
    rc = acquire_query_slot(h, &t); if (rc) return rc;
    QuerySlot& q = h->qslots[t];
    q.query_id = query_frame_id;
    const int n_elig = eligible_prefix(h, query_frame_id, h->params.min_gap);
    // cross_check scores the pair in both directions: the query rows then also serve in the TRAIN role and need its
    // padding rows (copies of the last row)
    const int rows_up = h->params.cross_check ? padded_rows(nq) + ROW_PAD : nq;
    const size_t bytes = (size_t)std::max(rows_up, 1) * LCM_DESC_BYTES;
    rc = ensure_pinned(q.h_query, q.h_query_bytes, bytes); if (rc) return rc;
    rc = ensure_dev(q.d_query, q.d_query_bytes, bytes, ARENA_SLACK); if (rc) return rc;
    hipStream_t S = h->stream;
    if (!big) { rc = pick_stream(h, q, &S); if (rc) return rc; }
    if (nq > 0 && n_elig > 0) {
        memcpy(q.h_query, query, (size_t)nq * LCM_DESC_BYTES);       // the caller's buffer is free when we return
        for (int r = nq; r < rows_up; ++r) memcpy(q.h_query + (size_t)r * LCM_DESC_BYTES, query + (size_t)(nq - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
        HIP_TRY(hipMemcpyAsync(q.d_query, q.h_query, (size_t)rows_up * LCM_DESC_BYTES, hipMemcpyHostToDevice, S));
    }
    if (big) {
        q.n_batch = 0;
        rc = upload_counts(h, q, &nq, 1); if (rc) return rc;
        rc = enqueue_big(h, q, q.d_query, reinterpret_cast<const int32_t*>(q.d_meta), rows_up, 1, &nq, &query_frame_id, &n_elig); if (rc) return rc;
    } else {
        rc = enqueue_query(h, q, (const uint32_t*)q.d_query, nq, n_elig, S); if (rc) return rc;
    }
    q.busy = true;
    q.db_generation = h->db_generation;
    *ticket = t;
    return LCM_OK;
}

static int query_collect_impl(lcm_handle* h, int ticket, lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out) {
    if (!h || !n_out || ticket < 0 || ticket >= QUERY_SLOTS || !h->qslots[ticket].busy) return fail(LCM_ERR_INVALID_ARG, "bad ticket");
    if (h->qslots[ticket].n_batch > 0) return fail(LCM_ERR_INVALID_ARG, "ticket %d is a batch: use lcm_query_collect_batch", ticket);
    *n_out = 0;
    int rc = set_device(h); if (rc) return rc;
    QuerySlot& q = h->qslots[ticket];
    if (q.db_generation != h->db_generation) {
        // the database was cleared / reloaded after the submit: the records describe slots that are gone
        (void)hipEventSynchronize(q.done);
        q.busy = false;
        return fail(LCM_ERR_NOT_FOUND, "ticket %d was submitted before lcm_db_clear / lcm_db_load: its result is void", ticket);
    }
    HIP_TRY(hipEventSynchronize(q.done));
    // Recoverable argument errors keep the ticket: the finished result can be collected again with enough room.
    if (q.n_elig > cap) return fail(LCM_ERR_CAPACITY, "%d score records but room for %d (the ticket stays valid)", q.n_elig, cap);
    if (q.n_elig > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL (the ticket stays valid)");
    if (q.n_elig > 0) {
        memcpy(out_scores, q.h_scores, sizeof(lcm_score) * (size_t)q.n_elig);
        // slots [0, n_elig) existed at submit time and appends only add slots behind them
        if (out_frame_ids) for (int s = 0; s < q.n_elig; ++s) out_frame_ids[s] = h->frames[s].id;
    }
    *n_out = q.n_elig;
    q.busy = false;
    fold_online_stats(h, q);
    return LCM_OK;
}

static int query_submit_batch_impl(lcm_handle* h, const uint8_t* const* queries, const int* nq, const int* query_frame_ids,
                                   int n_queries, int* ticket) {
    if (!h || !ticket || !queries || !nq || !query_frame_ids) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *ticket = -1;
    if (n_queries < 1 || n_queries > lcm::MAX_QUERY_BATCH) return fail(LCM_ERR_INVALID_ARG, "a batch holds 1..%d queries", lcm::MAX_QUERY_BATCH);
    int max_nq = 0;
    for (int b = 0; b < n_queries; ++b) {
        if (nq[b] < 0 || (nq[b] > 0 && !queries[b])) return fail(LCM_ERR_INVALID_ARG, "query %d: bad rows", b);
        if (nq[b] > 65535) return fail(LCM_ERR_CAPACITY, "a frame may hold at most 65535 rows (query %d has %d)", b, nq[b]);
        max_nq = std::max(max_nq, nq[b]);
    }
    int rc = set_device(h); if (rc) return rc;
    int t = -1;
    rc = acquire_query_slot(h, &t); if (rc) return rc;
    QuerySlot& q = h->qslots[t];
    q.query_id = query_frame_ids[0];
    int elig[lcm::MAX_QUERY_BATCH];
    size_t total = 0;
    for (int b = 0; b < n_queries; ++b) { elig[b] = eligible_prefix(h, query_frame_ids[b], h->params.min_gap); total += (size_t)elig[b]; }
    // staging pitch: every query gets the same number of rows, a whole number of the chunks enqueue_batch will cut
    int pitch = std::max(max_nq, 1);
    {
        const int qpt = pick_online_split(h, max_nq, total, true);       // the same call enqueue_batch makes
        if (is_split(qpt)) pitch = round_up(max_nq, split_chunk_rows(qpt));
    }
    if (h->params.cross_check) pitch = padded_rows(std::max(max_nq, 1)) + 2 * ROW_PAD;   // room for every query's padding rows
    const size_t bytes = (size_t)pitch * (size_t)n_queries * LCM_DESC_BYTES;
    rc = ensure_pinned(q.h_query, q.h_query_bytes, bytes); if (rc) return rc;
    rc = ensure_dev(q.d_query, q.d_query_bytes, bytes, ARENA_SLACK); if (rc) return rc;
    const bool big = max_nq > lcm::MAX_FUSED_QUERY_ROWS;          // served by the bulk search's packed route (enqueue_big)
    hipStream_t S = h->stream;
    if (!big) { rc = pick_stream(h, q, &S); if (rc) return rc; }
    if (total > 0) {
        for (int b = 0; b < n_queries; ++b)          // the callers' buffers are free when we return
            if (nq[b] > 0) {
                uint8_t* dst = q.h_query + (size_t)b * pitch * LCM_DESC_BYTES;
                memcpy(dst, queries[b], (size_t)nq[b] * LCM_DESC_BYTES);
                if (h->params.cross_check)
                    for (int r = nq[b]; r < padded_rows(nq[b]) + ROW_PAD; ++r) memcpy(dst + (size_t)r * LCM_DESC_BYTES, queries[b] + (size_t)(nq[b] - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
            }
        HIP_TRY(hipMemcpyAsync(q.d_query, q.h_query, bytes, hipMemcpyHostToDevice, S));
    }
    if (big) {
        q.n_batch = n_queries;
        rc = upload_counts(h, q, nq, n_queries); if (rc) return rc;
        rc = enqueue_big(h, q, q.d_query, reinterpret_cast<const int32_t*>(q.d_meta), pitch, n_queries, nq, query_frame_ids, elig); if (rc) return rc;
    } else {
        rc = enqueue_batch(h, q, (const uint32_t*)q.d_query, pitch, n_queries, nq, elig, S); if (rc) return rc;
    }
    q.busy = true;
    q.db_generation = h->db_generation;
    *ticket = t;
    return LCM_OK;
}

static int query_collect_batch_impl(lcm_handle* h, int ticket, lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets) {
    if (!h || !n_out || ticket < 0 || ticket >= QUERY_SLOTS || !h->qslots[ticket].busy || h->qslots[ticket].n_batch <= 0)
        return fail(LCM_ERR_INVALID_ARG, "bad batch ticket");
    *n_out = 0;
    int rc = set_device(h); if (rc) return rc;
    QuerySlot& q = h->qslots[ticket];
    if (q.db_generation != h->db_generation) {
        (void)hipEventSynchronize(q.done);
        q.busy = false; q.n_batch = 0;
        return fail(LCM_ERR_NOT_FOUND, "ticket %d was submitted before lcm_db_clear / lcm_db_load: its result is void", ticket);
    }
    HIP_TRY(hipEventSynchronize(q.done));
    if ((size_t)q.n_elig > cap) return fail(LCM_ERR_CAPACITY, "%d score records but room for %zu (the ticket stays valid)", q.n_elig, cap);
    if (q.n_elig > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL (the ticket stays valid)");
    if (q.n_elig > 0) memcpy(out_scores, q.h_scores, sizeof(lcm_score) * (size_t)q.n_elig);
    if (offsets) {
        size_t o = 0;
        for (int b = 0; b < q.n_batch; ++b) { offsets[b] = o; o += (size_t)q.bat_elig[b]; }
        offsets[q.n_batch] = o;
    }
    *n_out = (size_t)q.n_elig;
    q.busy = false; q.n_batch = 0;
    fold_online_stats(h, q);
    return LCM_OK;
}

static int query_scores_impl(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id,
                     lcm_score* out_scores, int32_t* out_frame_ids, int* n_out) {
    if (!h || !n_out) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    int t = -1;
    int rc = lcm_query_submit(h, query, nq, query_frame_id, &t); if (rc) return rc;
    return lcm_query_collect(h, t, out_scores, out_frame_ids, lcm_db_size(h), n_out);
}

static int detect_loops_impl(lcm_handle* h, int current_frame_id, const uint8_t* query, int nq, int n_keypoints,
                     lcm_loop_candidate* out, int cap, int* n_out) {
    if (!h || !n_out || cap < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    int rc = set_device(h); if (rc) return rc;
    int q_kp = n_keypoints;
    int t = -1;
    if (query) {
        if (nq < 0) return fail(LCM_ERR_INVALID_ARG, "negative row count");
        if (q_kp < 0) q_kp = nq;
        rc = lcm_query_submit(h, query, nq, current_frame_id, &t); if (rc) return rc;
    } else {
        // the current frame is already stored: its device rows are the query, nothing is uploaded
        const int slot = find_slot(h, current_frame_id);
        if (slot < 0) return fail(LCM_ERR_NOT_FOUND, "frame id %d is not stored", current_frame_id);
        nq = h->frames[slot].n;
        q_kp = h->frames[slot].n_kp;
        rc = acquire_query_slot(h, &t); if (rc) return rc;
        QuerySlot& q = h->qslots[t];
        q.query_id = current_frame_id;
        const uint8_t* d_frame = h->d_rows + (size_t)slot * h->stride_rows * LCM_DESC_BYTES;
        const int n_elig = eligible_prefix(h, current_frame_id, h->params.min_gap);
        if (nq > lcm::MAX_FUSED_QUERY_ROWS) {       // a stored frame above 2048 rows as query: the bulk search's packed route
            q.n_batch = 0;
            rc = enqueue_big(h, q, d_frame, h->d_counts + slot, h->stride_rows, 1, &nq, &current_frame_id, &n_elig);
        } else {
            hipStream_t S;
            rc = pick_stream(h, q, &S); if (rc) return rc;
            rc = enqueue_query(h, q, (const uint32_t*)d_frame, nq, n_elig, S);
        }
        if (rc) return rc;
        q.busy = true;
        q.db_generation = h->db_generation;
    }
    QuerySlot& q = h->qslots[t];
    q.busy = false;                                  // the ticket never leaves this function, whatever happens below
    HIP_TRY(hipEventSynchronize(q.done));
    fold_online_stats(h, q);
    int k = 0, total = 0;
    for (int s = 0; s < q.n_elig; ++s) {
        double sim;
        if (lcm_loop_test(&h->params, &q.h_scores[s], q_kp, h->frames[s].n_kp, &sim)) {
            if (k < cap && out) {
                out[k].current_frame_id = current_frame_id;
                out[k].matched_frame_id = h->frames[s].id;
                out[k].num_matches = (int32_t)q.h_scores[s].good_count;
                out[k].similarity_score = sim;
                ++k;
            }
            ++total;
        }
    }
    *n_out = k;
    if (total > k) return fail(LCM_ERR_CAPACITY, "%d loop candidates but room for %d", total, cap);
    return LCM_OK;
}

/* ---- exported entry points, behind the exception guard ---- */

int lcm_query_submit(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, int* ticket) {
    return guarded([&] { return query_submit_impl(h, query, nq, query_frame_id, ticket); });
}
int lcm_query_collect(lcm_handle* h, int ticket, lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out) {
    return guarded([&] { return query_collect_impl(h, ticket, out_scores, out_frame_ids, cap, n_out); });
}
int lcm_query_submit_batch(lcm_handle* h, const uint8_t* const* queries, const int* nq, const int* query_frame_ids, int n_queries, int* ticket) {
    return guarded([&] { return query_submit_batch_impl(h, queries, nq, query_frame_ids, n_queries, ticket); });
}
int lcm_query_collect_batch(lcm_handle* h, int ticket, lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets) {
    return guarded([&] { return query_collect_batch_impl(h, ticket, out_scores, cap, n_out, offsets); });
}
int lcm_query_scores(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, lcm_score* out_scores, int32_t* out_frame_ids, int* n_out) {
    return guarded([&] { return query_scores_impl(h, query, nq, query_frame_id, out_scores, out_frame_ids, n_out); });
}
int lcm_detect_loops(lcm_handle* h, int current_frame_id, const uint8_t* query, int nq, int n_keypoints, lcm_loop_candidate* out, int cap, int* n_out) {
    return guarded([&] { return detect_loops_impl(h, current_frame_id, query, nq, n_keypoints, out, cap, n_out); });
}

}  // extern "C"
