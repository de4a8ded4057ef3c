// lcm_pair.cpp — pair mode: matchFeatures for one pair or many (PairItem work items, device segment fold, host filter), cross-check combine.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

extern "C" {

/* ---- pair mode --------------------------------------------------------------------------------------- */

// Row source of the pair mode: host rows (uploaded to scratch) or rows already on the device (a stored frame).
struct RowSrc {
    const uint8_t* host;
    const uint8_t* dev;
    int n;
};

// One matchFeatures job of a batch: query rows x train rows, both given as ROW INDICES into one query matrix and one
// train matrix on the device (the database arena, or this call's staging block).
struct PairJob { uint32_t q_row; int nq; uint32_t t_row; int nt; };

// Best packed key (dist << 22 | GLOBAL train index) of every query row of every job -> keys (pinned host memory owned
// by the handle; job p's rows start at row0[p]).
//
// Every pair is cut into (query chunk of <= 2048 rows) x (train segment of SEG rows) work items (PairItem) so that
// a single 2000 x 2000 match occupies ~64 workgroups instead of one and a batch of loop candidates fills the chip:
// ONE launch of the key kernel over all items of all pairs, ONE launch of k_fold_pair_keys (per-segment keys carry
// segment-local train indices; the fold adds the segment base and takes the min, so the FIRST minimum wins across
// segments), ONE download.  `stage_bytes` bytes at h->h_pair_stage (already filled by the caller with any host rows)
// precede the items / descriptors this function appends, and the whole block goes up in ONE hipMemcpyAsync.
static int run_pair_jobs(lcm_handle* h, const uint8_t* d_q_base, const uint8_t* d_t_base, bool q_in_stage, bool t_in_stage,
                         size_t stage_bytes, const std::vector<PairJob>& jobs, const uint32_t** keys_out, std::vector<size_t>& row0) {
    const size_t P = jobs.size();
    // Work-item shape.  Throughput shape: query chunks of 2048 rows (8 per lane).  LATENCY shape, for calls of up to 64 M
    // distances (one matchFeatures of 2000 x 2000 is 4 M): chunks of 512 rows (2 per lane) — four times the waves with a
    // quarter of the work each (lcm_kernels.h, launch_score_pairs_small).
    uint64_t call_distances = 0;
    for (const PairJob& jb : jobs) call_distances += (uint64_t)jb.nq * (uint64_t)jb.nt;
    const bool small = call_distances <= (64ull << 20);
    const int CH = small ? 512 : lcm::MAX_FUSED_QUERY_ROWS;
    row0.assign(P + 1, 0);
    size_t n_items = 0, total_rows = 0;
    int max_nq = 0;
    size_t chunks_total = 0;
    for (const PairJob& jb : jobs) chunks_total += (size_t)((jb.nq + CH - 1) / CH);
    // aim at ~1536 workgroups over the whole batch (6 per CU); a segment is at least 32 rows, a multiple of 16
    const int seg_target = (int)std::max<size_t>(1, 1536 / std::max<size_t>(chunks_total, 1));
    std::vector<lcm::PairItem> items;
    std::vector<lcm::PairDesc> descs(P);
    for (size_t p = 0; p < P; ++p) {
        const PairJob& jb = jobs[p];
        if (jb.nt > LCM_MAX_TRAIN_ROWS) return fail(LCM_ERR_CAPACITY, "at most %d train rows per matrix", LCM_MAX_TRAIN_ROWS);
        const int n_chunks = (jb.nq + CH - 1) / CH;
        int n_seg = std::max(1, std::min((jb.nt + 31) / 32, seg_target));
        const int SEG = round_up((jb.nt + n_seg - 1) / n_seg, 16);
        n_seg = std::max(1, (jb.nt + SEG - 1) / SEG);
        if (SEG >= (1 << 20)) return fail(LCM_ERR_CAPACITY, "train segment of %d rows", SEG);
        descs[p] = {(uint32_t)items.size(), (uint32_t)n_seg, (uint32_t)SEG, (uint32_t)jb.nq, (uint32_t)total_rows};
        for (int c = 0; c < n_chunks; ++c)
            for (int g = 0; g < n_seg; ++g) {
                const uint32_t nqc = (uint32_t)std::min(CH, jb.nq - c * CH), ntg = (uint32_t)std::min(SEG, jb.nt - g * SEG);
                items.push_back({jb.q_row + (uint32_t)(c * CH), jb.t_row + (uint32_t)(g * SEG), nqc | (ntg << 12), (uint32_t)items.size()});
            }
        row0[p] = total_rows;
        total_rows += (size_t)jb.nq;
        max_nq = std::max(max_nq, jb.nq);
    }
    row0[P] = total_rows;
    n_items = items.size();
    *keys_out = nullptr;
    if (n_items == 0 || total_rows == 0) return LCM_OK;

    // ---- one staging block up: [caller's rows | items | descriptors]
    const size_t off_items = (stage_bytes + 255) & ~(size_t)255;
    const size_t off_descs = off_items + sizeof(lcm::PairItem) * n_items;
    const size_t up_bytes = off_descs + sizeof(lcm::PairDesc) * P;
    int rc = LCM_OK;
    if (up_bytes > h->h_pair_stage_bytes) {           // grow, keeping the rows the caller has already staged
        uint8_t* bigger = nullptr;
        const size_t want = up_bytes + up_bytes / 2;
        HIP_TRY(hipHostMalloc((void**)&bigger, want, hipHostMallocDefault));
        if (h->h_pair_stage) { memcpy(bigger, h->h_pair_stage, std::min(stage_bytes, h->h_pair_stage_bytes)); HIP_TRY(hipHostFree(h->h_pair_stage)); }
        h->h_pair_stage = bigger; h->h_pair_stage_bytes = want;
    }
    rc = ensure_dev(h->d_pair_stage, h->d_pair_stage_bytes, up_bytes, ARENA_SLACK); if (rc) return rc;
    memcpy(h->h_pair_stage + off_items, items.data(), sizeof(lcm::PairItem) * n_items);
    memcpy(h->h_pair_stage + off_descs, descs.data(), sizeof(lcm::PairDesc) * P);
    rc = ensure_dev(h->d_keys, h->d_keys_n, n_items * (size_t)CH + total_rows); if (rc) return rc;
    rc = ensure_pinned(h->h_final_keys, h->h_final_keys_n, total_rows); if (rc) return rc;
    // only the part the caller did not fill needs the copy when the rows are device-resident already
    const size_t up_from = (q_in_stage || t_in_stage) ? 0 : off_items;
    if (small && h->tune_pair_upload_kernel) {
        // latency shape: the staging block goes up by a KERNEL on the compute queue (the score kernel follows it after a
        // normal kernel-to-kernel gap; a DMA copy hands over across engines first: ~6 us more per call)
        const size_t from16 = up_from & ~(size_t)15;
        const hipError_t eu = lcm::launch_upload(h->d_pair_stage + from16, h->h_pair_stage + from16, up_bytes - from16, h->stream);
        if (eu != hipSuccess) return fail(LCM_ERR_HIP, "upload kernel launch failed: %s", hipGetErrorString(eu));
    } else {
        HIP_TRY(hipMemcpyAsync(h->d_pair_stage + up_from, h->h_pair_stage + up_from, up_bytes - up_from, hipMemcpyHostToDevice, h->stream));
    }

    lcm::ScoreArgs a{};
    a.q_rows = (const uint32_t*)(q_in_stage ? h->d_pair_stage : d_q_base);
    a.db_rows = (const uint32_t*)(t_in_stage ? h->d_pair_stage : d_t_base);
    a.pair_items = reinterpret_cast<const lcm::PairItem*>(h->d_pair_stage + off_items);
    a.scores = nullptr; a.keys = h->d_keys; a.keys_stride = CH;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    if (small) {
        HIP_TRY(hipEventRecord(h->ev_start, h->stream));
        const hipError_t e1 = lcm::launch_score_pairs_small(a, (uint32_t)n_items, h->stream);
        if (e1 != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e1));
        HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
        h->info_pending = true;
        h->info.workgroups = (uint32_t)n_items; h->info.route = LCM_ROUTE_PLAIN;
    } else {
        rc = launch_and_time(h, a, (uint32_t)n_items, max_nq > CH ? CH : max_nq, true); if (rc) return rc;
    }
    lcm::FoldArgs f{};
    f.seg_keys = h->d_keys;
    f.chunk_rows = (uint32_t)CH;
    f.pairs = reinterpret_cast<const lcm::PairDesc*>(h->d_pair_stage + off_descs);
    // Latency shape: the fold kernel writes the folded keys STRAIGHT into the pinned host buffer (mapped into the device's
    // address space; 4 bytes per query row over PCIe), so no device-to-host copy packet follows it: the stream's
    // synchronisation below is also the hand-over.  Throughput shape: device buffer + one copy.
    const bool host_fold = small && h->tune_pair_host_fold;
    f.final_keys = host_fold ? h->h_final_keys : h->d_keys + n_items * (size_t)CH;
    f.n_pairs = (uint32_t)P;
    hipError_t e = lcm::launch_fold_pair_keys(f, (uint32_t)max_nq, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "fold kernel launch failed: %s", hipGetErrorString(e));
    h->info.launches = 2;
    h->info.pairs = P; h->info.distances = 0; h->info.algo_bytes = 0;
    for (const PairJob& jb : jobs) {
        h->info.distances += (uint64_t)jb.nq * (uint64_t)jb.nt;
        h->info.algo_bytes += (uint64_t)jb.nt * 32 + (uint64_t)jb.nq * 32 + 8;
    }
    if (!host_fold) HIP_TRY(hipMemcpyAsync(h->h_final_keys, f.final_keys, sizeof(uint32_t) * total_rows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *keys_out = h->h_final_keys;
    return LCM_OK;
}

// Cross-check on shipped keys (integer bookkeeping, O(nq + nt)): fkeys[q] = (d, first nearest train row of q),
// bkeys[t] = (d, first nearest query row of t) -> keys[q] = the match q keeps, 0xFFFFFFFF = none.  Same rule as
// k_cross_score / oracle orc_bf_match_cross.
static void cross_combine(int mode, const uint32_t* fkeys, int nq, const uint32_t* bkeys, int nt, std::vector<uint32_t>& keys) {
    keys.assign((size_t)nq, 0xFFFFFFFFu);
    if (mode == 1) {
        for (int q = 0; q < nq; ++q)
            if ((int)(bkeys[fkeys[q] & lcm::KEY_IDX_MASK] & lcm::KEY_IDX_MASK) == q) keys[(size_t)q] = fkeys[q];
    } else {
        for (int t = 0; t < nt; ++t) {
            const uint32_t i = bkeys[t] & lcm::KEY_IDX_MASK;
            const uint32_t cand = (bkeys[t] & ~lcm::KEY_IDX_MASK) | (uint32_t)t;
            if (cand < keys[i]) keys[i] = cand;       // (dist, t) lexicographic: strict '<' on dist, first t on ties
        }
    }
}

// One pair with rows from the host and / or the device: host rows travel inside the staging block (a matrix in the TRAIN
// role with its padding rows — copies of the last row — written straight into pinned memory: no extra copies).
// With cross_check the pair runs twice, roles swapped the second time, and the two key arrays are combined.
// `cross` = the cross-check mode to apply (the handle's, for the outer call; 0 for the two one-directional passes the
// cross-check itself is made of).  It is an ARGUMENT: the handle's parameters are never touched, so no failure in the
// nested passes — an exception included — can leave the handle with its cross-check switched off.
static int pair_keys(lcm_handle* h, RowSrc q, RowSrc t, std::vector<uint32_t>& keys_out, int cross) {
    int rc = set_device(h); if (rc) return rc;
    if (cross) {
        if (q.n > LCM_MAX_TRAIN_ROWS) return fail(LCM_ERR_CAPACITY, "cross_check: at most %d query rows", LCM_MAX_TRAIN_ROWS);
        std::vector<uint32_t> fk, bk;
        rc = pair_keys(h, q, t, fk, 0); if (rc) return rc;
        rc = pair_keys(h, t, q, bk, 0); if (rc) return rc;
        cross_combine(cross, fk.data(), q.n, bk.data(), t.n, keys_out);
        return LCM_OK;
    }
    const size_t q_bytes = q.dev ? 0 : (size_t)q.n * LCM_DESC_BYTES;
    const size_t t_off = (q_bytes + 255) & ~(size_t)255;
    const size_t t_bytes = t.dev ? 0 : (size_t)(padded_rows(t.n) + ROW_PAD) * LCM_DESC_BYTES;
    const size_t stage_bytes = t_off + t_bytes;
    if (stage_bytes) {
        rc = ensure_pinned(h->h_pair_stage, h->h_pair_stage_bytes, stage_bytes + 65536); if (rc) return rc;
        if (!q.dev && q.n > 0) memcpy(h->h_pair_stage, q.host, q_bytes);
        if (!t.dev && t.n > 0) {
            uint8_t* dst = h->h_pair_stage + t_off;
            memcpy(dst, t.host, (size_t)t.n * LCM_DESC_BYTES);
            for (int r = t.n; r < padded_rows(t.n) + ROW_PAD; ++r) memcpy(dst + (size_t)r * LCM_DESC_BYTES, t.host + (size_t)(t.n - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
        }
    }
    if (q.dev || t.dev) {
        rc = wait_db(h); if (rc) return rc;
        if ((size_t)h->cap_frames * (size_t)h->stride_rows >= 0xFFFFFFFFull)
            return fail(LCM_ERR_CAPACITY, "the database arena exceeds 2^32 rows: pair items address rows with 32 bits");
    }
    // device-resident sides are addressed from the arena base (row index = byte offset / 32)
    std::vector<PairJob> jobs(1);
    jobs[0].nq = q.n; jobs[0].nt = t.n;
    jobs[0].q_row = q.dev ? (uint32_t)((size_t)(q.dev - h->d_rows) / LCM_DESC_BYTES) : 0u;
    jobs[0].t_row = t.dev ? (uint32_t)((size_t)(t.dev - h->d_rows) / LCM_DESC_BYTES) : (uint32_t)(t_off / LCM_DESC_BYTES);
    const uint32_t* keys = nullptr;
    std::vector<size_t> row0;
    rc = run_pair_jobs(h, h->d_rows, h->d_rows, !q.dev, !t.dev, stage_bytes, jobs, &keys, row0); if (rc) return rc;
    keys_out.assign(keys, keys + q.n);
    return LCM_OK;
}

// Device rows + row counts of a stored frame.
static int stored_src(lcm_handle* h, int frame_id, RowSrc* out, int* n_kp) {
    int lo = 0, hi = (int)h->frames.size();
    while (lo < hi) { int mid = (lo + hi) / 2; if (h->frames[mid].id < frame_id) lo = mid + 1; else hi = mid; }
    if (lo >= (int)h->frames.size() || h->frames[lo].id != frame_id) return fail(LCM_ERR_NOT_FOUND, "frame id %d is not stored", frame_id);
    out->host = nullptr;
    out->dev = h->d_rows + (size_t)lo * h->stride_rows * LCM_DESC_BYTES;
    out->n = h->frames[lo].n;
    if (n_kp) *n_kp = h->frames[lo].n_kp;
    return LCM_OK;
}

static int filter_keys(const lcm_handle* h, const std::vector<uint32_t>& keys, int nq, lcm_dmatch* out, int* n_out, int* min_dist) {
    // README.md:117 filter on the shipped integers (O(nq) bookkeeping)
    uint32_t m = 0xFFFFFFFFu;
    for (int i = 0; i < nq; ++i) if (keys[i] != 0xFFFFFFFFu) m = std::min(m, keys[i] >> lcm::KEY_SHIFT);
    const uint32_t thr = std::max((uint32_t)h->params.ratio * m, (uint32_t)h->params.dist_floor);
    int k = 0;
    for (int i = 0; i < nq; ++i) {
        const uint32_t d = keys[i] >> lcm::KEY_SHIFT;
        if (keys[i] != 0xFFFFFFFFu && d <= thr) {      // 0xFFFFFFFF: the cross-check left this query unmatched
            out[k].query_idx = i;
            out[k].train_idx = (int32_t)(keys[i] & lcm::KEY_IDX_MASK);
            out[k].img_idx = 0;
            out[k].distance = (float)d;
            ++k;
        }
    }
    *n_out = k;
    if (min_dist) *min_dist = m == 0xFFFFFFFFu ? -1 : (int)m;
    return LCM_OK;
}

static int match_pair_impl(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                   int32_t* train_idx, uint16_t* dist, int* n_matches) {
    if (!h || nq < 0 || nt < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    if (n_matches) *n_matches = 0;
    if (nq == 0 || nt == 0) return LCM_OK;            // BFMatcher: no train rows => no matches
    if (!query || !train || !train_idx || !dist) return fail(LCM_ERR_INVALID_ARG, "NULL buffer");
    std::vector<uint32_t> keys;
    int rc = pair_keys(h, RowSrc{query, nullptr, nq}, RowSrc{train, nullptr, nt}, keys, h->params.cross_check); if (rc) return rc;
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (keys[i] == 0xFFFFFFFFu) { train_idx[i] = -1; dist[i] = 0xFFFF; continue; }   // cross_check: no match for row i
        train_idx[i] = (int32_t)(keys[i] & lcm::KEY_IDX_MASK);
        dist[i] = (uint16_t)(keys[i] >> lcm::KEY_SHIFT);
        ++n;
    }
    if (n_matches) *n_matches = h->params.cross_check ? n : nq;
    return LCM_OK;
}

static int match_features_impl(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                       lcm_dmatch* out, int* n_out, int* min_dist) {
    if (!h || nq < 0 || nt < 0 || !n_out) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    if (min_dist) *min_dist = -1;
    if (nq == 0 || nt == 0) return LCM_OK;
    if (!query || !train || !out) return fail(LCM_ERR_INVALID_ARG, "NULL buffer");
    std::vector<uint32_t> keys;
    int rc = pair_keys(h, RowSrc{query, nullptr, nq}, RowSrc{train, nullptr, nt}, keys, h->params.cross_check); if (rc) return rc;
    return filter_keys(h, keys, nq, out, n_out, min_dist);
}

static int match_stored_impl(lcm_handle* h, int query_frame_id, int train_frame_id, lcm_dmatch* out, int cap, int* n_out, int* min_dist) {
    if (!h || !n_out) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    if (min_dist) *min_dist = -1;
    int rc = set_device(h); if (rc) return rc;
    RowSrc q{}, t{};
    rc = stored_src(h, query_frame_id, &q, nullptr); if (rc) return rc;
    rc = stored_src(h, train_frame_id, &t, nullptr); if (rc) return rc;
    if (q.n == 0 || t.n == 0) return LCM_OK;
    if (!out || cap < q.n) return fail(LCM_ERR_CAPACITY, "need room for %d matches", q.n);
    std::vector<uint32_t> keys;
    rc = pair_keys(h, q, t, keys, h->params.cross_check); if (rc) return rc;
    return filter_keys(h, keys, q.n, out, n_out, min_dist);
}

// Keys of one job -> its DMatch list appended at out[*n_total ...] (README.md:117 filter, query order kept).
static int emit_matches(const lcm_handle* h, const uint32_t* keys, int nq, lcm_dmatch* out, size_t cap, size_t* n_total, int32_t* min_dist) {
    uint32_t m = 0xFFFFFFFFu;
    for (int i = 0; i < nq; ++i) if (keys[i] != 0xFFFFFFFFu) m = std::min(m, keys[i] >> lcm::KEY_SHIFT);
    const uint32_t thr = std::max((uint32_t)h->params.ratio * m, (uint32_t)h->params.dist_floor);
    size_t k = *n_total;
    for (int i = 0; i < nq; ++i) {
        const uint32_t d = keys[i] >> lcm::KEY_SHIFT;
        if (keys[i] != 0xFFFFFFFFu && d <= thr) {
            if (k >= cap) return fail(LCM_ERR_CAPACITY, "match buffer holds %zu records: too small", cap);
            out[k].query_idx = i;
            out[k].train_idx = (int32_t)(keys[i] & lcm::KEY_IDX_MASK);
            out[k].img_idx = 0;
            out[k].distance = (float)d;
            ++k;
        }
    }
    *n_total = k;
    if (min_dist) *min_dist = m == 0xFFFFFFFFu ? -1 : (int32_t)m;
    return LCM_OK;
}

// matchFeatures for MANY pairs in one launch (N1: the match lists of all loop candidates of a frame, README.md:101).
// q_host != NULL: one query frame from the host against stored train frames; else both sides stored.
static int match_batch_impl(lcm_handle* h, const uint8_t* q_host, int nq_host, const lcm_pair_ref* pairs, const int32_t* train_ids,
                            int n_pairs, lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists) {
    if (!h || n_pairs < 0 || !offsets || (n_pairs > 0 && !pairs && !train_ids)) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    offsets[0] = 0;
    int rc = set_device(h); if (rc) return rc;
    if (q_host && nq_host > lcm::MAX_FUSED_QUERY_ROWS * 64) return fail(LCM_ERR_CAPACITY, "query frame too large");
    std::vector<PairJob> jobs;
    std::vector<int> job_of((size_t)n_pairs, -1);
    size_t stage_bytes = 0;
    if (q_host && nq_host > 0) {
        stage_bytes = (size_t)nq_host * LCM_DESC_BYTES;
        rc = ensure_pinned(h->h_pair_stage, h->h_pair_stage_bytes, stage_bytes + 65536 + (size_t)n_pairs * 2048); if (rc) return rc;
        memcpy(h->h_pair_stage, q_host, stage_bytes);
    }
    for (int p = 0; p < n_pairs; ++p) {
        RowSrc q{}, t{};
        if (q_host) { q.host = q_host; q.n = nq_host; }
        else { rc = stored_src(h, pairs[p].query_frame_id, &q, nullptr); if (rc) return rc; }
        rc = stored_src(h, q_host ? train_ids[p] : pairs[p].train_frame_id, &t, nullptr); if (rc) return rc;
        if (q.n == 0 || t.n == 0) continue;                     // BFMatcher: an empty side => no matches
        job_of[(size_t)p] = (int)jobs.size();
        jobs.push_back({q.dev ? (uint32_t)((size_t)(q.dev - h->d_rows) / LCM_DESC_BYTES) : 0u, q.n,
                        (uint32_t)((size_t)(t.dev - h->d_rows) / LCM_DESC_BYTES), t.n});
    }
    rc = wait_db(h); if (rc) return rc;
    if ((size_t)h->cap_frames * (size_t)h->stride_rows >= 0xFFFFFFFFull)
        return fail(LCM_ERR_CAPACITY, "the database arena exceeds 2^32 rows: pair items address rows with 32 bits");
    const uint32_t* keys = nullptr;
    std::vector<size_t> row0;
    rc = run_pair_jobs(h, h->d_rows, h->d_rows, q_host != nullptr, false, stage_bytes, jobs, &keys, row0); if (rc) return rc;
    std::vector<uint32_t> fwd, bwd_all, ck;
    std::vector<size_t> brow0;
    if (h->params.cross_check && !jobs.empty()) {
        // second pass, roles swapped: every train row's first nearest QUERY row.  A host query travels again, this time
        // with the padding rows the train role needs.
        fwd.assign(keys, keys + row0.back());
        std::vector<PairJob> back(jobs.size());
        size_t bstage = 0;
        if (q_host) {
            const int np = padded_rows(nq_host) + ROW_PAD;
            bstage = (size_t)np * LCM_DESC_BYTES;
            rc = ensure_pinned(h->h_pair_stage, h->h_pair_stage_bytes, bstage + 65536 + (size_t)n_pairs * 2048); if (rc) return rc;
            memcpy(h->h_pair_stage, q_host, (size_t)nq_host * LCM_DESC_BYTES);
            for (int r = nq_host; r < np; ++r) memcpy(h->h_pair_stage + (size_t)r * LCM_DESC_BYTES, q_host + (size_t)(nq_host - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
        }
        for (size_t j = 0; j < jobs.size(); ++j) back[j] = {jobs[j].t_row, jobs[j].nt, jobs[j].q_row, jobs[j].nq};
        rc = run_pair_jobs(h, h->d_rows, h->d_rows, false, q_host != nullptr, bstage, back, &keys, brow0); if (rc) return rc;
        bwd_all.assign(keys, keys + brow0.back());
    }
    size_t total = 0;
    for (int p = 0; p < n_pairs; ++p) {
        offsets[p] = total;
        if (min_dists) min_dists[p] = -1;
        const int j = job_of[(size_t)p];
        if (j < 0) continue;
        const uint32_t* kp = keys + row0[(size_t)j];
        if (h->params.cross_check) {
            cross_combine(h->params.cross_check, fwd.data() + row0[(size_t)j], jobs[(size_t)j].nq, bwd_all.data() + brow0[(size_t)j], jobs[(size_t)j].nt, ck);
            kp = ck.data();
        }
        rc = emit_matches(h, kp, jobs[(size_t)j].nq, out, out ? cap : 0, &total, min_dists ? &min_dists[p] : nullptr);
        if (rc) return rc;
    }
    offsets[n_pairs] = total;
    return LCM_OK;
}

/* ---- exported entry points, behind the exception guard ---- */

int lcm_match_pair(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* train_idx, uint16_t* dist, int* n_matches) {
    return guarded([&] { return match_pair_impl(h, query, nq, train, nt, train_idx, dist, n_matches); });
}
int lcm_match_features(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt, lcm_dmatch* out, int* n_out, int* min_dist) {
    return guarded([&] { return match_features_impl(h, query, nq, train, nt, out, n_out, min_dist); });
}
int lcm_match_stored(lcm_handle* h, int query_frame_id, int train_frame_id, lcm_dmatch* out, int cap, int* n_out, int* min_dist) {
    return guarded([&] { return match_stored_impl(h, query_frame_id, train_frame_id, out, cap, n_out, min_dist); });
}
int lcm_match_stored_batch(lcm_handle* h, const lcm_pair_ref* pairs, int n_pairs, lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists) {
    return guarded([&] { return match_batch_impl(h, nullptr, 0, pairs, nullptr, n_pairs, out, cap, offsets, min_dists); });
}
int lcm_match_query_batch(lcm_handle* h, const uint8_t* query, int nq, const int32_t* train_frame_ids, int n_trains, lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists) {
    if (nq < 0 || (nq > 0 && !query)) return fail(LCM_ERR_INVALID_ARG, "bad query rows");
    static const uint8_t none[LCM_DESC_BYTES] = {0};
    return guarded([&] { return match_batch_impl(h, query ? query : none, nq, nullptr, train_frame_ids, n_trains, out, cap, offsets, min_dists); });
}

}  // extern "C"
