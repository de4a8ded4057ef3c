// lcm_bulk.cpp — the bulk all-vs-all search: plan (cached work list), launches, argmin / cross-check / matrix-core routing, fused loop test.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

namespace {

// The packed form of a bulk search (lcm_kernels.h, ScoreArgs::pk_*).  Per chunk of <= PK_CHUNK_PAIRS pairs (8 KB of
// per-row scratch each) the query frames that have work are laid end to end in order of descending eligibility and cut
// into 2048-row columns; a column is scored against slots [0, eligibility of its first frame) in runs of `chunk` slots.
struct PackedPlan {
    std::vector<uint32_t> tab;                  // [pair offsets (n_q + 1) | row counts (n_q) | per chunk: vstart, qframe, elig, pairs]
    std::vector<lcm::WorkItem> items;           // (column, slot run) items, chunk by chunk; out_offset = the column's first position
    std::vector<Plan::PackedChunk> chunks;
    size_t max_pairs = 0;                       // largest chunk
    uint64_t lane_slots = 0;                    // sum over columns of 2048 x slots scored: what the launch occupies
};

int build_packed_plan(const std::vector<size_t>& offsets, const std::vector<int32_t>& qn, const uint32_t* q_frame_of, int chunk,
                      uint32_t COL, uint32_t stride, size_t scratch_words, PackedPlan& pk) {
    const size_t PK_CHUNK_PAIRS = std::max<size_t>(1, scratch_words / stride);      // per-row scratch of one chunk (default 8 GiB)
    const int n_q = (int)qn.size();
    auto elig_of = [&](int c) { return (uint32_t)(offsets[(size_t)c + 1] - offsets[(size_t)c]); };
    pk.tab.resize((size_t)n_q * 2 + 1);
    for (int c = 0; c <= n_q; ++c) pk.tab[(size_t)c] = (uint32_t)offsets[(size_t)c];
    for (int c = 0; c < n_q; ++c) pk.tab[(size_t)n_q + 1 + (size_t)c] = (uint32_t)qn[(size_t)c];
    std::vector<int> pos;
    int c0 = 0;
    while (c0 < n_q) {
        int c1 = c0;
        size_t pairs = 0;
        while (c1 < n_q && (pairs == 0 || pairs + elig_of(c1) <= PK_CHUNK_PAIRS)) { pairs += elig_of(c1); ++c1; }
        if (pairs > 0) {
            pos.clear();
            for (int c = c0; c < c1; ++c) if (elig_of(c) > 0 && qn[(size_t)c] > 0) pos.push_back(c);
            std::stable_sort(pos.begin(), pos.end(), [&](int x, int y) { return elig_of(x) > elig_of(y); });
            Plan::PackedChunk ch{};
            ch.item0 = (uint32_t)pk.items.size(); ch.tab0 = (uint32_t)pk.tab.size(); ch.n_pos = (uint32_t)pos.size();
            ch.pair_base = (uint32_t)offsets[(size_t)c0]; ch.n_pairs = (uint32_t)pairs;
            const size_t np = pos.size();
            pk.tab.resize(pk.tab.size() + 4 * np + 1);
            uint32_t* vstart = pk.tab.data() + ch.tab0;
            uint32_t* qframe = vstart + np + 1;
            uint32_t* eligp = qframe + np;
            uint32_t* pairsp = eligp + np;
            uint64_t v = 0;
            for (size_t k = 0; k < np; ++k) {
                const int c = pos[k];
                vstart[k] = (uint32_t)v; v += (uint64_t)qn[(size_t)c];
                qframe[k] = q_frame_of ? q_frame_of[c] : (uint32_t)c;
                eligp[k] = elig_of(c);
                pairsp[k] = (uint32_t)(offsets[(size_t)c] - offsets[(size_t)c0]);
            }
            if (v > 0xFFFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^32 query rows in one chunk");
            vstart[np] = (uint32_t)v;
            size_t k = 0;
            for (uint64_t w = 0; w * COL < v; ++w) {
                while (vstart[k + 1] <= w * COL) ++k;          // position holding the column's first row
                const uint32_t me = eligp[k];                  // the largest eligibility in the column (descending order)
                pk.lane_slots += (uint64_t)COL * me;
                for (uint32_t b = 0; b < me; b += (uint32_t)chunk)
                    pk.items.push_back({(uint32_t)w, b, std::min((uint32_t)chunk, me - b), (uint32_t)k});
            }
            ch.n_items = (uint32_t)pk.items.size() - ch.item0;
            pk.chunks.push_back(ch);
            pk.max_pairs = std::max(pk.max_pairs, pairs);
        }
        c0 = c1;
    }
    return LCM_OK;
}

// Packed route: score kernel (per-row best distance / key of every eligible pair) + fold kernel, chunk by chunk on one
// stream: the fold of chunk k is done with the scratch before the scores of chunk k + 1 are written.
int launch_packed(lcm_handle* h, const Plan& P, lcm::ScoreArgs a, bool argmin, void* d_scores, uint32_t* d_idx_sums) {
    int rc = ensure_dev(h->d_mdist, h->d_mdist_n, P.pk_max_pairs * (size_t)P.pk_stride); if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    uint32_t launches = 0, biggest = 0;
    for (const Plan::PackedChunk& ch : P.pk_chunks) {
        a.items = P.d_items + ch.item0;
        a.pk_vstart = P.d_pk_tab + ch.tab0;
        a.pk_qframe = a.pk_vstart + ch.n_pos + 1;
        a.pk_elig = a.pk_qframe + ch.n_pos;
        a.pk_pairs = a.pk_elig + ch.n_pos;
        a.pk_dist = h->d_mdist; a.pk_n = ch.n_pos; a.pk_col_rows = P.pk_col_rows; a.pk_stride = P.pk_stride;
        hipError_t e = lcm::launch_score_packed(a, ch.n_items, argmin, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeBulkArgs f{};
        f.stride = P.pk_stride; f.key_shift = argmin ? lcm::KEY_SHIFT : 0; f.idx_sums = d_idx_sums;
        f.dist = h->d_mdist; f.offsets = P.d_pk_tab; f.nq = reinterpret_cast<const int32_t*>(P.d_pk_tab + P.pk_n_q + 1);
        f.db_counts = h->d_counts; f.scores = d_scores; f.n_q = P.pk_n_q; f.pair_base = ch.pair_base;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        const bool last = (&ch == &P.pk_chunks.back());      // the last chunk's fold is timed by itself (aux_kernel_ms)
        if (last) HIP_TRY(hipEventRecord(h->ev_aux_start, h->stream));
        e = lcm::launch_finalize_bulk(f, ch.n_pairs, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "fold kernel launch failed: %s", hipGetErrorString(e));
        if (last) { HIP_TRY(hipEventRecord(h->ev_aux_stop, h->stream)); h->aux_pending = true; }
        launches += 2; biggest = std::max(biggest, ch.n_items);
    }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = launches; h->info.workgroups = biggest; h->info.route = LCM_ROUTE_PACKED;
    h->info.pairs = P.n_pairs; h->info.distances = P.distances; h->info.algo_bytes = P.algo_bytes;
    return LCM_OK;
}

}  // namespace

extern "C" {

/* ---- bulk all-vs-all --------------------------------------------------------------------------------- */

static int all_vs_all_impl(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts,
                   const int32_t* q_ids, int n_q_frames, int q_stride_rows,
                   void* d_scores, size_t scores_cap, size_t* n_pairs, size_t* pair_offsets,
                   uint32_t* d_idx_sums = nullptr, const uint32_t* q_frame_of = nullptr,
                   const int32_t* h_query_counts = nullptr) {
    if (!h || !n_pairs) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    int rc = set_device(h); if (rc) return rc;
    const bool self = (d_query_rows == nullptr);
    std::vector<int32_t> self_ids;
    if (self) {
        n_q_frames = (int)h->frames.size();
        self_ids.resize(n_q_frames);
        for (int i = 0; i < n_q_frames; ++i) self_ids[i] = h->frames[i].id;
        q_ids = self_ids.data();
        q_stride_rows = h->stride_rows;
    } else if (!d_query_counts || !q_ids || n_q_frames < 0 || q_stride_rows <= 0) {
        return fail(LCM_ERR_INVALID_ARG, "external query set needs counts, ids and a stride");
    }

    // ---- plan (cached while the database, the query-id list AND the query frames' row counts are unchanged)
    // The row counts of an external query set live on the device and may change between calls with the same ids, and
    // they pick the workgroup shape (a stale, smaller maximum would silently skip rows): they are fetched on every
    // call (n_q_frames * 4 bytes) and are part of the key, as is the stride.
    std::vector<int32_t> qc;
    if (!self && n_q_frames > 0) {
        qc.resize((size_t)n_q_frames);
        if (h_query_counts) {            // the caller (lcm_group_*) already knows them on the host
            memcpy(qc.data(), h_query_counts, sizeof(int32_t) * (size_t)n_q_frames);
        } else {
            HIP_TRY(hipMemcpyAsync(qc.data(), d_query_counts, sizeof(int32_t) * (size_t)n_q_frames, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        for (int c = 0; c < n_q_frames; ++c)
            if (qc[c] < 0 || qc[c] > q_stride_rows) return fail(LCM_ERR_INVALID_ARG, "query frame %d has %d rows, stride %d", c, qc[c], q_stride_rows);
    }
    // the packed form (full 2048-row workgroups across query frames) serves the row-per-lane kernels only
    const bool pack_ok = h->tune_packed != 0 && !h->params.cross_check && (h->variant < 2 || h->variant >= 4);
    uint64_t key = mix(mix(mix(0x1234, (uint64_t)h->frames.size()), (uint64_t)n_q_frames), (uint64_t)h->params.min_gap);
    key = mix(mix(key, self ? 1 : 2), (uint64_t)q_stride_rows);
    key = mix(key, h->db_generation);
    key = mix(key, pack_ok ? 0x9Bull + (uint64_t)(h->tune_packed + 1) : 0x9Aull);
    key = mix(key, (uint64_t)h->pk_scratch_words);
    for (int i = 0; i < n_q_frames; ++i) key = mix(key, (uint64_t)(uint32_t)q_ids[i]);
    if (q_frame_of) for (int i = 0; i < n_q_frames; ++i) key = mix(key, 0x51ull + q_frame_of[i]);
    for (int32_t c : qc) key = mix(key, (uint64_t)(uint32_t)c);
    if (!h->frames.empty()) key = mix(mix(key, (uint64_t)h->frames.front().id), (uint64_t)h->frames.back().id);
    if (key == 0) key = 1;
    Plan& P = h->plan;
    if (P.key != key) {
        P.key = 0;                        // a failed rebuild must not leave a half-built plan behind the old key
        P.items.clear();
        P.packed = false; P.pk_chunks.clear(); P.pk_max_pairs = 0;
        P.offsets.assign((size_t)n_q_frames + 1, 0);
        size_t total = 0;
        for (int c = 0; c < n_q_frames; ++c) { P.offsets[c] = total; total += (size_t)eligible_prefix(h, q_ids[c], h->params.min_gap); }
        P.offsets[n_q_frames] = total;
        if (total > 0xFFFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^32 pairs in one call");
        const int chunk = pick_chunk(h, total);
        P.distances = 0; P.algo_bytes = 0; P.max_q_rows = 0;
        // prefix sums of stored row counts for the distance / byte accounting
        std::vector<uint64_t> pre(h->frames.size() + 1, 0);
        for (size_t s = 0; s < h->frames.size(); ++s) pre[s + 1] = pre[s] + (uint64_t)h->frames[s].n;
        std::vector<int32_t> qn((size_t)n_q_frames);          // rows per query frame
        for (int c = 0; c < n_q_frames; ++c) qn[(size_t)c] = self ? h->frames[(size_t)c].n : qc[(size_t)c];
        auto elig_of = [&](int c) { return (uint32_t)(P.offsets[(size_t)c + 1] - P.offsets[(size_t)c]); };
        for (int c = 0; c < n_q_frames; ++c) {
            const uint32_t e = elig_of(c);
            if (e > 0) {
                P.distances += (uint64_t)qn[(size_t)c] * pre[e];
                P.algo_bytes += pre[e] * 32 + (uint64_t)qn[(size_t)c] * 32 + 8ull * e;
                P.max_q_rows = std::max(P.max_q_rows, (int)qn[(size_t)c]);
            }
        }
        // Query frames above 2048 rows (ORB with nfeatures > 2048) do not fit one workgroup's registers: only the packed
        // route serves them (a frame then spans several 2048-row columns), so it is taken whatever its size.
        const bool big_rows = P.max_q_rows > lcm::MAX_FUSED_QUERY_ROWS;
        if (big_rows && !pack_ok)
            return fail(LCM_ERR_CAPACITY, "query frames above %d rows need the packed bulk route (not with cross_check, kernel variants 2 / 3 or LCM_TUNE_PACKED = 0)", lcm::MAX_FUSED_QUERY_ROWS);

        // ---- packed form (ScoreArgs::pk_*): built beside the accounting, adopted when it saves lane slots
        PackedPlan pk;
        if (pack_ok && total > 0) {
            P.pk_col_rows = h->tune_packed == 2 ? 1536u : (uint32_t)lcm::MAX_FUSED_QUERY_ROWS;
            // scratch words per pair: 2048, or the largest query frame (rounded up) when frames exceed that
            P.pk_stride = big_rows ? (uint32_t)round_up(P.max_q_rows, 256) : (uint32_t)lcm::MAX_FUSED_QUERY_ROWS;
            rc = build_packed_plan(P.offsets, qn, q_frame_of, chunk, P.pk_col_rows, P.pk_stride, h->pk_scratch_words, pk); if (rc) return rc;
            const uint64_t shape_rows = P.max_q_rows <= 512 ? 512 : P.max_q_rows <= 1024 ? 1024 : P.max_q_rows <= 1536 ? 1536 : 2048;
            const uint64_t lanes_plain = (uint64_t)total * shape_rows;
            // automatic: worth it when it saves >= 1 % of the lane slots of a search big enough to be throughput-bound
            P.packed = big_rows || h->tune_packed >= 1 || (total >= 8192 && pk.lane_slots * 100 <= lanes_plain * 99);
        }
        if (P.packed) {
            P.items.swap(pk.items);
            P.pk_chunks.swap(pk.chunks);
            P.pk_max_pairs = pk.max_pairs;
            P.pk_n_q = (uint32_t)n_q_frames;
            rc = ensure_dev(P.d_pk_tab, P.d_pk_tab_n, pk.tab.size()); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(P.d_pk_tab, pk.tab.data(), sizeof(uint32_t) * pk.tab.size(), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));       // `tab` is a local: consumed before any early return below
        } else {
            // heaviest query frames first so the tail of the launch is made of short items
            for (int c = n_q_frames - 1; c >= 0; --c) {
                const int e = (int)elig_of(c);
                for (int b = 0; b < e; b += chunk)
                    P.items.push_back({q_frame_of ? q_frame_of[c] : (uint32_t)c, (uint32_t)b, (uint32_t)std::min(chunk, e - b), (uint32_t)(P.offsets[c] + b)});
            }
        }
        P.n_pairs = total;
        if (!P.items.empty()) {
            rc = ensure_dev(P.d_items, P.d_items_cap, P.items.size()); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(P.d_items, P.items.data(), sizeof(lcm::WorkItem) * P.items.size(), hipMemcpyHostToDevice, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
        P.key = key;
    }
    *n_pairs = P.n_pairs;
    if (pair_offsets) memcpy(pair_offsets, P.offsets.data(), sizeof(size_t) * ((size_t)n_q_frames + 1));
    if (!d_scores) return LCM_OK;       // sizing call
    if (scores_cap < P.n_pairs) return fail(LCM_ERR_CAPACITY, "scores buffer holds %zu records, need %zu", scores_cap, P.n_pairs);
    if (P.n_pairs == 0) return LCM_OK;

    rc = wait_db(h); if (rc) return rc;
    if (h->params.cross_check) {
        // BFMatcher crossCheck: every pair is matched in both directions and folded on the device
        std::vector<uint32_t> row0((size_t)n_q_frames);
        std::vector<int> nqv((size_t)n_q_frames), ev((size_t)n_q_frames);
        const uint8_t* qbase = h->d_rows;
        uint32_t pitch = (uint32_t)h->stride_rows;
        if (!self) {
            // the caller's rows carry no train-role padding: work on a padded copy
            pitch = (uint32_t)(padded_rows(q_stride_rows) + 2 * ROW_PAD);
            uint32_t n_slots = 0;
            for (int c = 0; c < n_q_frames; ++c) n_slots = std::max(n_slots, (q_frame_of ? q_frame_of[c] : (uint32_t)c) + 1);
            rc = ensure_dev(h->d_xq, h->d_xq_bytes, (size_t)n_slots * pitch * LCM_DESC_BYTES, ARENA_SLACK); if (rc) return rc;
            HIP_TRY(hipMemcpy2DAsync(h->d_xq, (size_t)pitch * LCM_DESC_BYTES, d_query_rows, (size_t)q_stride_rows * LCM_DESC_BYTES,
                                     (size_t)q_stride_rows * LCM_DESC_BYTES, n_slots, hipMemcpyDeviceToDevice, h->stream));
            hipError_t e = lcm::launch_pad_rows((uint32_t*)h->d_xq, d_query_counts, pitch, n_slots, h->stream);
            if (e != hipSuccess) return fail(LCM_ERR_HIP, "pad kernel launch failed: %s", hipGetErrorString(e));
            qbase = h->d_xq;
        }
        for (int c = 0; c < n_q_frames; ++c) {
            row0[(size_t)c] = (q_frame_of ? q_frame_of[c] : (uint32_t)c) * pitch;
            nqv[(size_t)c] = self ? h->frames[(size_t)c].n : qc[(size_t)c];
            ev[(size_t)c] = (int)(P.offsets[(size_t)c + 1] - P.offsets[(size_t)c]);
        }
        return cross_score_prefixes(h, qbase, row0.data(), nqv.data(), ev.data(), n_q_frames, (lcm_score*)d_scores, d_idx_sums);
    }
    if ((h->variant == 4 || h->variant == 5) && P.max_q_rows <= lcm::MAX_FUSED_QUERY_ROWS) {      // (bigger query frames: packed vector-ALU route)
        std::vector<int> nqv((size_t)n_q_frames);
        for (int c = 0; c < n_q_frames; ++c) nqv[(size_t)c] = self ? h->frames[(size_t)c].n : qc[(size_t)c];
        return mfma_bulk(h, self, self ? h->d_rows : (const uint8_t*)d_query_rows, d_query_counts,
                         (uint32_t)(self ? h->stride_rows : q_stride_rows), q_frame_of, nqv.data(), n_q_frames, P.offsets, (lcm_score*)d_scores, d_idx_sums);
    }
    lcm::ScoreArgs a{};
    a.q_rows = self ? (const uint32_t*)h->d_rows : (const uint32_t*)d_query_rows;
    a.q_counts = self ? h->d_counts : d_query_counts;
    a.q_stride_words = (uint32_t)q_stride_rows * LCM_DESC_WORDS;
    a.db_rows = (const uint32_t*)h->d_rows; a.db_counts = h->d_counts; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
    a.items = P.d_items; a.scores = d_scores; a.keys = nullptr; a.keys_stride = 0;
    a.idx_sums = d_idx_sums;             // non-NULL: the argmin kernel (variant 1) runs whatever the handle's variant
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    const int variant = d_idx_sums ? 1 : h->variant;
    if (P.packed) {
        rc = launch_packed(h, P, a, variant == 1, d_scores, d_idx_sums);
        // The per-row scratch (8 GiB per chunk by default) did not fit next to whatever else lives on this device: halve the
        // chunk and plan again, down to 64 MiB, rather than fail a search whose inputs and outputs do fit.
        if (rc == LCM_ERR_OOM && h->pk_scratch_words > ((size_t)1 << 24)) {
            (void)hipGetLastError();                 // the failed allocation's sticky error must not be read as a launch failure
            h->pk_scratch_words >>= 1;
            P.key = 0;
            return all_vs_all_impl(h, d_query_rows, d_query_counts, self ? nullptr : q_ids, n_q_frames, q_stride_rows, d_scores, scores_cap,
                                   n_pairs, pair_offsets, d_idx_sums, q_frame_of, h_query_counts);
        }
        return rc;
    }
    // Very large searches go out as several launches (<= 2^20 work items, a few seconds each): no single kernel runs
    // long enough to meet a compute-queue timeout, and the stream stays responsive.
    constexpr size_t MAX_ITEMS_PER_LAUNCH = 1u << 20;
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    uint32_t launches = 0, biggest = 0;
    for (size_t first = 0; first < P.items.size(); first += MAX_ITEMS_PER_LAUNCH) {
        const uint32_t n = (uint32_t)std::min(MAX_ITEMS_PER_LAUNCH, P.items.size() - first);
        a.items = P.d_items + first;
        hipError_t e = lcm::launch_score(a, n, P.max_q_rows, false, variant, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        ++launches; biggest = std::max(biggest, n);
    }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = launches; h->info.workgroups = biggest; h->info.route = LCM_ROUTE_PLAIN;
    h->info.pairs = P.n_pairs; h->info.distances = P.distances; h->info.algo_bytes = P.algo_bytes;
    return LCM_OK;
}

}  // extern "C"

// the same search for the other translation units (lcm_group.cpp): index map + host-side row counts
namespace lcm {
int all_vs_all(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts, const int32_t* q_ids,
               int n_q_frames, int q_stride_rows, void* d_scores, size_t scores_cap, size_t* n_pairs,
               size_t* pair_offsets, uint32_t* d_idx_sums, const uint32_t* q_frame_of, const int32_t* h_query_counts) {
    return guarded([&] { return all_vs_all_impl(h, d_query_rows, d_query_counts, q_ids, n_q_frames, q_stride_rows, d_scores,
                                                scores_cap, n_pairs, pair_offsets, d_idx_sums, q_frame_of, h_query_counts); });
}
}  // namespace lcm

extern "C" {

// Bulk loop search with the loop test fused on the device: all-vs-all scores stay in device memory, a second tiny
// kernel applies README.md:123-126 per pair and compacts the candidates; only those cross PCIe.
static int all_vs_all_loops_impl(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts,
                         const int32_t* q_ids, const int32_t* q_keypoints, int n_q_frames, int q_stride_rows,
                         lcm_loop_candidate* out, size_t cap, size_t* n_out, size_t* n_pairs_out) {
    if (!h || !n_out) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    int rc = set_device(h); if (rc) return rc;
    const bool self = (d_query_rows == nullptr);
    size_t n_pairs = 0;
    rc = lcm_all_vs_all(h, d_query_rows, d_query_counts, q_ids, n_q_frames, q_stride_rows, nullptr, 0, &n_pairs, nullptr);
    if (rc) return rc;
    if (n_pairs_out) *n_pairs_out = n_pairs;
    h->bulk_scores_valid = 0;
    if (n_pairs == 0) return LCM_OK;
    rc = ensure_dev(h->d_bulk_scores, h->d_bulk_scores_n, n_pairs); if (rc) return rc;
    rc = lcm_all_vs_all(h, d_query_rows, d_query_counts, q_ids, n_q_frames, q_stride_rows, h->d_bulk_scores, n_pairs, &n_pairs, nullptr);
    if (rc) return rc;
    const Plan& P = h->plan;
    const int nq = self ? (int)h->frames.size() : n_q_frames;
    const int ns = (int)h->frames.size();
    // metadata the loop test needs, as one upload: offsets | q_ids | q_kp | db_ids | db_kp
    std::vector<int32_t> meta((size_t)(nq + 1) + 2 * (size_t)nq + 2 * (size_t)ns);
    int32_t* m_off = meta.data();
    int32_t* m_qid = m_off + (nq + 1);
    int32_t* m_qkp = m_qid + nq;
    int32_t* m_did = m_qkp + nq;
    int32_t* m_dkp = m_did + ns;
    for (int c = 0; c <= nq; ++c) m_off[c] = (int32_t)(uint32_t)P.offsets[c];
    std::vector<int32_t> qc;
    if (!self && !q_keypoints) {           // external query set without keypoint counts: rows == keypoints (ORB)
        qc.resize((size_t)nq);
        HIP_TRY(hipMemcpy(qc.data(), d_query_counts, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost));
    }
    for (int c = 0; c < nq; ++c) {
        m_qid[c] = self ? h->frames[c].id : q_ids[c];
        m_qkp[c] = self ? h->frames[c].n_kp : (q_keypoints ? q_keypoints[c] : qc[c]);
    }
    for (int s = 0; s < ns; ++s) { m_did[s] = h->frames[s].id; m_dkp[s] = h->frames[s].n_kp; }
    const size_t n_blocks = (n_pairs + 255) / 256;
    rc = ensure_dev(h->d_meta, h->d_meta_n, meta.size() + 4 + n_blocks); if (rc) return rc;
    const size_t dev_cap = std::max<size_t>(std::min<size_t>(cap, n_pairs), 1);
    rc = ensure_dev(h->d_cands, h->d_cands_n, dev_cap); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice, h->stream));
    uint32_t* d_counter = reinterpret_cast<uint32_t*>(h->d_meta + meta.size());
    HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(uint32_t), h->stream));
    lcm::LoopTestArgs a{};
    a.scores = h->d_bulk_scores;
    a.offsets = reinterpret_cast<const uint32_t*>(h->d_meta);
    a.q_ids = h->d_meta + (nq + 1); a.q_kp = a.q_ids + nq; a.db_ids = a.q_kp + nq; a.db_kp = a.db_ids + ns;
    a.out = h->d_cands; a.counter = d_counter;
    a.block_counts = d_counter + 4;
    a.n_q = (uint32_t)nq; a.n_pairs = (uint32_t)n_pairs; a.cap = (uint32_t)dev_cap;
    a.min_matches = h->params.min_matches; a.sim_threshold = h->params.sim_threshold;
    HIP_TRY(hipEventRecord(h->ev_aux_start, h->stream));
    hipError_t e = lcm::launch_loop_test(a, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "loop-test kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(h->ev_aux_stop, h->stream));
    h->aux_pending = true;
    h->bulk_scores_valid = n_pairs;
    uint32_t found = 0;
    HIP_TRY(hipMemcpyAsync(&found, d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n_out = found;
    if (found > cap || !out) return found ? fail(LCM_ERR_CAPACITY, "%u loop candidates but room for %zu", found, cap) : LCM_OK;
    // the device compacted them in pair order = (current id, matched id) ascending: nothing to sort
    if (found) HIP_TRY(hipMemcpy(out, h->d_cands, sizeof(lcm_loop_candidate) * found, hipMemcpyDeviceToHost));
    return LCM_OK;
}

/* ---- exported entry points, behind the exception guard ---- */

int lcm_all_vs_all(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts, const int32_t* q_ids, int n_q_frames, int q_stride_rows, void* d_scores, size_t scores_cap, size_t* n_pairs, size_t* pair_offsets) {
    return guarded([&] { return all_vs_all_impl(h, d_query_rows, d_query_counts, q_ids, n_q_frames, q_stride_rows, d_scores, scores_cap, n_pairs, pair_offsets); });
}
int lcm_all_vs_all_argmin(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts, const int32_t* q_ids, int n_q_frames, int q_stride_rows, void* d_scores, size_t scores_cap, void* d_index_sums, size_t* n_pairs, size_t* pair_offsets) {
    if (d_scores && !d_index_sums) return fail(LCM_ERR_INVALID_ARG, "d_index_sums is NULL");
    return guarded([&] { return all_vs_all_impl(h, d_query_rows, d_query_counts, q_ids, n_q_frames, q_stride_rows, d_scores, scores_cap, n_pairs, pair_offsets, (uint32_t*)d_index_sums); });
}
int lcm_all_vs_all_loops(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts, const int32_t* q_ids, const int32_t* q_keypoints, int n_q_frames, int q_stride_rows, lcm_loop_candidate* out, size_t cap, size_t* n_out, size_t* n_pairs_out) {
    return guarded([&] { return all_vs_all_loops_impl(h, d_query_rows, d_query_counts, q_ids, q_keypoints, n_q_frames, q_stride_rows, out, cap, n_out, n_pairs_out); });
}

}  // extern "C"
