// lcm_bulk.cpp — the bulk all-vs-all search: plan (cached work list), launches, argmin / cross-check / matrix-core routing, fused loop test.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

namespace {

// The packed form of a bulk search (lcm_kernels.h, ScoreArgs::pk_*).  The query frames that have work are taken in GROUPS
// of consecutive frames; a group's frames are laid end to end in order of descending eligibility and cut into 2048-row
// columns; a chunk (one score launch + one fold launch) is a group against a RANGE of stored slots whose pairs fit the
// per-row scratch.  Cutting over slots as well as over frames is what keeps the columns full when the scratch is small:
// a 64 K-pair chunk of frames with 5000 eligible slots each would hold 13 frames — 13 columns, 2000 of 2048 lanes busy,
// the packing gone (measured: cfg4 2.3 % slower at 1 GiB than at 8 GiB); as 256 frames x 256 slots it fills 250 columns.
struct PackedPlan {
    std::vector<uint32_t> tab;                  // [pair offsets (n_q + 1) | row counts (n_q) | per group / per chunk tables]
    std::vector<lcm::WorkItem> items;           // (column, slot run) items, chunk by chunk; out_offset = the column's first position
    std::vector<Plan::PackedChunk> chunks;
    size_t max_pairs = 0;                       // largest chunk
    uint64_t lane_slots = 0;                    // sum over columns of 2048 x slots scored: what the launches occupy
};

constexpr int PK_GROUP_FRAMES = 256;            // frames per group once a group no longer fits one chunk whole (0.2 % idle in its last column)

int build_packed_plan(const std::vector<size_t>& offsets, const std::vector<int32_t>& qn, const uint32_t* q_frame_of, int chunk,
                      uint32_t COL, uint32_t stride, size_t scratch_words, PackedPlan& pk) {
    const size_t budget = std::max<size_t>(1, scratch_words / stride);      // pairs whose per-row words fit one chunk's scratch
    const int n_q = (int)qn.size();
    auto elig_of = [&](int c) { return (uint32_t)(offsets[(size_t)c + 1] - offsets[(size_t)c]); };
    pk.tab.resize((size_t)n_q * 2 + 1);
    for (int c = 0; c <= n_q; ++c) pk.tab[(size_t)c] = (uint32_t)offsets[(size_t)c];
    for (int c = 0; c < n_q; ++c) pk.tab[(size_t)n_q + 1 + (size_t)c] = (uint32_t)qn[(size_t)c];
    std::vector<int> pos;
    struct Col { uint32_t w, k, me; };
    std::vector<Col> cols;
    bool first_chunk = offsets[(size_t)n_q] > budget;      // a search of one chunk stays one chunk
    int c0 = 0;
    while (c0 < n_q) {
        // ---- the group: as many consecutive frames as fit one chunk WHOLE, but at least PK_GROUP_FRAMES (those then
        //      meet the stored slots range by range); never so many that a range would be shorter than one work item
        const int min_frames = (int)std::max<size_t>(1, std::min<size_t>((size_t)PK_GROUP_FRAMES, budget / (size_t)std::max(chunk, 1)));
        int c1 = c0;
        size_t pairs = 0;
        while (c1 < n_q && (c1 - c0 < min_frames || pairs + elig_of(c1) <= budget)) { pairs += elig_of(c1); ++c1; }
        pos.clear();
        // (a query frame without rows occupies no virtual rows but keeps its scratch slots: its pairs' records — empty
        // ones — are written by the fold like everybody else's)
        for (int c = c0; c < c1; ++c) if (elig_of(c) > 0) pos.push_back(c);
        c0 = c1;
        if (pos.empty()) continue;
        std::stable_sort(pos.begin(), pos.end(), [&](int x, int y) { return elig_of(x) > elig_of(y); });
        const size_t np = pos.size();
        const uint32_t tab0 = (uint32_t)pk.tab.size();
        pk.tab.resize(pk.tab.size() + 4 * np + 1);
        uint32_t* vstart = pk.tab.data() + tab0;
        uint32_t* qframe = vstart + np + 1;
        uint32_t* eligp = qframe + np;
        uint32_t* cidx = eligp + np;
        uint64_t v = 0;
        for (size_t k = 0; k < np; ++k) {
            const int c = pos[k];
            vstart[k] = (uint32_t)v; v += (uint64_t)qn[(size_t)c];
            qframe[k] = q_frame_of ? q_frame_of[c] : (uint32_t)c;
            eligp[k] = elig_of(c);
            cidx[k] = (uint32_t)c;
        }
        if (v > 0xFFFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^32 query rows in one group");
        vstart[np] = (uint32_t)v;
        // Columns in packed order have DESCENDING eligibility (me), so the columns that still have work at slot run b
        // are a prefix.
        cols.clear();
        size_t k = 0;
        for (uint64_t w = 0; w * COL < v; ++w) {
            while (vstart[k + 1] <= w * COL) ++k;          // position holding the column's first row
            const uint32_t me = pk.tab[tab0 + 2 * np + 1 + k];     // = eligp[k]: the largest eligibility in the column
            pk.lane_slots += (uint64_t)COL * me;
            cols.push_back({(uint32_t)w, (uint32_t)k, me});
        }
        // ---- the group's chunks: ranges of stored slots [s0, s1), each holding <= budget pairs
        const uint32_t e_max = pk.tab[tab0 + 2 * np + 1];                  // eligp[0]
        uint32_t s0 = 0;
        while (s0 < e_max) {
            // The first chunk of a search is half-size: consecutive chunks alternate between two streams (launch_packed),
            // and equal chunks started together would also drain together; offset by half a chunk, one is always in full
            // flow while the other's launch drains.
            const size_t bud = first_chunk ? std::max<size_t>(budget / 2, 1) : budget;
            // longest range from s0 whose pairs fit: sum_k max(0, min(s1, elig_k) - s0) <= bud, a multiple of `chunk` slots
            uint32_t len = (uint32_t)std::max<size_t>((size_t)chunk, bud / np / (size_t)chunk * (size_t)chunk);
            auto pairs_in = [&](uint32_t s1) { size_t t = 0; for (size_t q = 0; q < np; ++q) { const uint32_t e = pk.tab[tab0 + 2 * np + 1 + q]; if (e <= s0) break; t += std::min(s1, e) - s0; } return t; };
            // frames run out of eligible slots as s0 grows: widen the range while it still fits
            while (s0 + len < e_max && pairs_in(s0 + len + (uint32_t)chunk) <= bud) len += (uint32_t)chunk;
            const uint32_t s1 = std::min<uint64_t>((uint64_t)s0 + len, e_max);
            Plan::PackedChunk ch{};
            ch.item0 = (uint32_t)pk.items.size(); ch.tab0 = tab0; ch.ptab0 = (uint32_t)pk.tab.size(); ch.n_pos = (uint32_t)np; ch.slot0 = s0;
            pk.tab.resize(pk.tab.size() + np + 1);
            {
                uint32_t* pairsp = pk.tab.data() + ch.ptab0;
                size_t t = 0;
                for (size_t q = 0; q < np; ++q) {
                    pairsp[q] = (uint32_t)t;
                    const uint32_t e = pk.tab[tab0 + 2 * np + 1 + q];
                    if (e > s0) t += std::min(s1, e) - s0;
                }
                pairsp[np] = (uint32_t)t;
                ch.n_pairs = (uint32_t)t;
            }
            // Items go out SLOT-RUN MAJOR: all columns against slots [b, b + chunk), then the next run.  Workgroups are
            // dispatched in index order, round-robin over the 8 XCDs, so the ~1500 workgroups in flight at any moment
            // stream the same few stored frames: each XCD's L2 fetches a stored frame once per run instead of once per
            // column (column-major order: every resident workgroup streamed a different frame and the database crossed
            // the fabric once per column — 0.6 % of HBM peak, but 28 GB per cfg2 pass for nothing).  Every item is the
            // same size, so the order does not change the tail of the launch.
            size_t live = cols.size();                         // columns [0, live) have me > b
            for (uint32_t bb = s0; bb < s1 && live > 0; bb += (uint32_t)chunk) {
                while (live > 0 && cols[live - 1].me <= bb) --live;
                for (size_t cc = 0; cc < live; ++cc)
                    pk.items.push_back({cols[cc].w, bb, std::min((uint32_t)chunk, std::min(s1, cols[cc].me) - bb), cols[cc].k});
            }
            ch.n_items = (uint32_t)pk.items.size() - ch.item0;
            if (ch.n_pairs > 0) {                              // (no items: a group of empty query frames — the fold still runs)
                pk.chunks.push_back(ch);
                pk.max_pairs = std::max(pk.max_pairs, (size_t)ch.n_pairs);
                first_chunk = false;
            }
            s0 = s1;
        }
    }
    return LCM_OK;
}

// Packed route: score kernel (per-row best distance / key of every eligible pair) + fold kernel, chunk by chunk.
// A search of several chunks alternates them between TWO streams and the two halves of the scratch: chunk k + 1's
// workgroups fill the chip while chunk k's launch drains and its fold runs (one stream: every chunk boundary cost one
// workgroup's run time, ~4 ms — 0.6 % of a cfg2 pass at 4 chunks, 2.3 % of a cfg4 pass at 96).  Chunks k and k + 2 share
// a stream, hence a scratch half: the fold of chunk k is done with it before the scores of chunk k + 2 are written.
int launch_packed(lcm_handle* h, const Plan& P, lcm::ScoreArgs a, bool argmin, void* d_scores, uint32_t* d_idx_sums) {
    const bool two = P.pk_chunks.size() > 1;
    // per-row scratch of the largest chunk (x 2 when the chunks alternate); a buffer left behind by a much larger search
    // (or by a one-off big-frame online query) is given back first instead of being carried for the rest of the handle's life
    const size_t half_words = P.pk_max_pairs * (size_t)P.pk_stride;
    const size_t need_words = half_words * (two ? 2 : 1);
    if (h->d_mdist && h->d_mdist_n > 4 * std::max<size_t>(need_words, (size_t)1 << 24)) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(h->d_mdist));
        h->d_mdist = nullptr; h->d_mdist_n = 0;
    }
    int rc = ensure_dev(h->d_mdist, h->d_mdist_n, need_words); if (rc) return rc;
    if (two && !h->stream2) {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    if (two) {      // the second stream starts behind everything the handle's stream holds (appends, the plan upload)
        HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    }
    h->fold_ev_used = 0;
    h->aux_pending = false;
    uint32_t launches = 0, biggest = 0;
    size_t k = 0;
    for (const Plan::PackedChunk& ch : P.pk_chunks) {
        hipStream_t st = (two && (k & 1)) ? h->stream2 : h->stream;
        uint32_t* scratch = h->d_mdist + ((two && (k & 1)) ? half_words : 0);
        ++k;
        a.items = P.d_items + ch.item0;
        a.pk_vstart = P.d_pk_tab + ch.tab0;
        a.pk_qframe = a.pk_vstart + ch.n_pos + 1;
        a.pk_elig = a.pk_qframe + ch.n_pos;
        a.pk_pairs = P.d_pk_tab + ch.ptab0; a.pk_slot0 = ch.slot0;
        a.pk_dist = scratch; a.pk_n = ch.n_pos; a.pk_col_rows = P.pk_col_rows; a.pk_stride = P.pk_stride;
        // three events per chunk on its stream: before the score kernel, between score and fold, after the fold
        while (h->fold_ev.size() < h->fold_ev_used + 3) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            h->fold_ev.push_back(ev);
        }
        HIP_TRY(hipEventRecord(h->fold_ev[h->fold_ev_used], st));
        hipError_t e = lcm::launch_score_packed(a, ch.n_items, argmin, st);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeBulkArgs f{};
        f.stride = P.pk_stride; f.key_shift = argmin ? lcm::KEY_SHIFT : 0; f.idx_sums = d_idx_sums;
        f.word_bytes = argmin ? 4 : 2;                        // distance-only: uint16 distances (k_score_rowlane's packed epilogue)
        f.dist = scratch; f.offsets = P.d_pk_tab; f.nq = reinterpret_cast<const int32_t*>(P.d_pk_tab + P.pk_n_q + 1);
        f.db_counts = h->d_counts; f.scores = d_scores; f.n_q = P.pk_n_q; f.pair_base = 0;
        f.pk_pairs = a.pk_pairs; f.pk_cidx = a.pk_elig + ch.n_pos; f.pk_n = ch.n_pos; f.slot0 = ch.slot0;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        HIP_TRY(hipEventRecord(h->fold_ev[h->fold_ev_used + 1], st));
        e = lcm::launch_finalize_bulk(f, ch.n_pairs, st);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "fold kernel launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipEventRecord(h->fold_ev[h->fold_ev_used + 2], st));
        h->fold_ev_used += 3;
        launches += 2; biggest = std::max(biggest, ch.n_items);
    }
    if (two) {      // the handle's stream continues only when both have finished
        HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = launches; h->info.workgroups = biggest; h->info.route = LCM_ROUTE_PACKED;
    h->info.launches_in_flight = two ? 2 : 1;
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
    lcm::PlanSig sig;
    sig.db_generation = h->db_generation; sig.n_db = h->frames.size();
    sig.n_q = n_q_frames; sig.gap = h->params.min_gap; sig.q_stride = q_stride_rows; sig.item_slots = h->tune_item_slots;
    sig.pack_mode = pack_ok ? 1 + h->tune_packed : -2; sig.self = self;
    if (!self) {
        sig.q_ids.assign(q_ids, q_ids + n_q_frames);
        sig.q_counts = qc;
        if (q_frame_of) sig.q_frame_of.assign(q_frame_of, q_frame_of + n_q_frames);
    }
    Plan& P = h->plan;
    // A valid plan is reused only if every input it was built from is the same (compared, not hashed).  A rebuild starts
    // from the CONFIGURED scratch size again — an earlier out-of-memory fallback does not shrink the chunks for good —
    // unless it IS that fallback's retry (pk_oom_retry), which plans with the size it has just halved.
    const bool oom_retry = h->pk_oom_retry;
    h->pk_oom_retry = false;
    sig.scratch_words = P.key != 0 ? P.sig.scratch_words : (oom_retry ? h->pk_scratch_words : h->pk_scratch_cfg_words);
    if (P.key == 0 || !(P.sig == sig)) {
        if (P.key != 0) sig.scratch_words = h->pk_scratch_cfg_words;      // the inputs changed under a valid plan: fresh start
        h->pk_scratch_words = sig.scratch_words;
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
            // a chunk takes HALF of the scratch: consecutive chunks alternate between its two halves (launch_packed)
            rc = build_packed_plan(P.offsets, qn, q_frame_of, chunk, P.pk_col_rows, P.pk_stride, h->pk_scratch_words / 2, pk); if (rc) return rc;
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
        P.sig = std::move(sig);
        P.key = 1;
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
        // The per-row scratch (1 GiB per chunk by default) did not fit next to whatever else lives on this device: halve the
        // chunk and plan again, down to 64 MiB, rather than fail a search whose inputs and outputs do fit.
        if (rc == LCM_ERR_OOM && h->pk_scratch_words > ((size_t)1 << 24)) {
            (void)hipGetLastError();                 // the failed allocation's sticky error must not be read as a launch failure
            h->pk_scratch_words >>= 1;
            h->pk_oom_retry = true;
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

namespace lcm {

// Loop test (README.md:123-126) + ordered compaction over a finished score array that lives on h's device, on h's stream.
// Pair p belongs to query frame c = last c with offsets[c] <= p and is the (p - offsets[c])-th frame of the `db_*` lists
// (a handle's own frames, or — for a group's shard — the frames that shard owns).  Two steps: count the candidates,
// read the count, and only then size the candidate buffer: a caller whose `cap` is too small learns the needed count
// (LCM_ERR_CAPACITY, *n_found set) BEFORE anything proportional to the pair count is allocated.  On LCM_OK the
// *n_found candidates are in h->d_cands, in pair order = (current id, matched id) ascending.
int loop_test_device(lcm_handle* h, const void* d_scores, size_t n_pairs, const uint32_t* offsets, int n_q,
                     const int32_t* q_ids, const int32_t* q_kp, int n_db, const int32_t* db_ids, const int32_t* db_kp,
                     size_t cap, size_t* n_found) {
    *n_found = 0;
    if (n_pairs == 0) return LCM_OK;
    { const int rc0 = set_device(h); if (rc0) return rc0; }      // (called from a group's per-device host threads too)
    // metadata the loop test needs, as one upload: offsets | q_ids | q_kp | db_ids | db_kp
    std::vector<int32_t> meta((size_t)(n_q + 1) + 2 * (size_t)n_q + 2 * (size_t)n_db);
    int32_t* m_off = meta.data();
    int32_t* m_qid = m_off + (n_q + 1);
    int32_t* m_qkp = m_qid + n_q;
    int32_t* m_did = m_qkp + n_q;
    int32_t* m_dkp = m_did + n_db;
    for (int c = 0; c <= n_q; ++c) m_off[c] = (int32_t)offsets[c];
    memcpy(m_qid, q_ids, sizeof(int32_t) * (size_t)n_q);
    memcpy(m_qkp, q_kp, sizeof(int32_t) * (size_t)n_q);
    memcpy(m_did, db_ids, sizeof(int32_t) * (size_t)n_db);
    memcpy(m_dkp, db_kp, sizeof(int32_t) * (size_t)n_db);
    const size_t n_blocks = (n_pairs + 255) / 256;
    int rc = ensure_dev(h->d_meta, h->d_meta_n, meta.size() + 4 + n_blocks); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice, h->stream));
    uint32_t* d_counter = reinterpret_cast<uint32_t*>(h->d_meta + meta.size());
    HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(uint32_t), h->stream));
    lcm::LoopTestArgs a{};
    a.scores = d_scores;
    a.offsets = reinterpret_cast<const uint32_t*>(h->d_meta);
    a.q_ids = h->d_meta + (n_q + 1); a.q_kp = a.q_ids + n_q; a.db_ids = a.q_kp + n_q; a.db_kp = a.db_ids + n_db;
    a.out = nullptr; a.counter = d_counter;
    a.block_counts = d_counter + 4;
    a.n_q = (uint32_t)n_q; a.n_pairs = (uint32_t)n_pairs; a.cap = 0;
    a.min_matches = h->params.min_matches; a.sim_threshold = h->params.sim_threshold;
    HIP_TRY(hipEventRecord(h->ev_aux_start, h->stream));
    hipError_t e = lcm::launch_loop_count(a, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "loop-test kernel launch failed: %s", hipGetErrorString(e));
    uint32_t found = 0;
    HIP_TRY(hipMemcpyAsync(&found, d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // also: `meta` (pageable) has been consumed
    *n_found = found;
    if (found > cap) {
        HIP_TRY(hipEventRecord(h->ev_aux_stop, h->stream));
        h->aux_pending = true;
        return fail(LCM_ERR_CAPACITY, "%u loop candidates but room for %zu", found, cap);
    }
    if (found) {
        rc = ensure_dev(h->d_cands, h->d_cands_n, (size_t)found); if (rc) return rc;
        a.out = h->d_cands; a.cap = found;
        e = lcm::launch_loop_emit(a, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "loop-test kernel launch failed: %s", hipGetErrorString(e));
    }
    HIP_TRY(hipEventRecord(h->ev_aux_stop, h->stream));
    h->aux_pending = true;
    return LCM_OK;
}

}  // namespace lcm

extern "C" {

// Bulk loop search with the loop test fused on the device: all-vs-all scores stay in device memory, the loop-test
// kernels apply README.md:123-126 per pair and compact the candidates; only those cross PCIe.
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
    std::vector<uint32_t> offs((size_t)nq + 1);
    std::vector<int32_t> qid((size_t)nq), qkp((size_t)nq), did((size_t)ns), dkp((size_t)ns);
    for (int c = 0; c <= nq; ++c) offs[(size_t)c] = (uint32_t)P.offsets[(size_t)c];
    std::vector<int32_t> qc;
    if (!self && !q_keypoints) {           // external query set without keypoint counts: rows == keypoints (ORB)
        qc.resize((size_t)nq);
        HIP_TRY(hipMemcpy(qc.data(), d_query_counts, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost));
    }
    for (int c = 0; c < nq; ++c) {
        qid[(size_t)c] = self ? h->frames[(size_t)c].id : q_ids[c];
        qkp[(size_t)c] = self ? h->frames[(size_t)c].n_kp : (q_keypoints ? q_keypoints[c] : qc[(size_t)c]);
    }
    for (int s = 0; s < ns; ++s) { did[(size_t)s] = h->frames[(size_t)s].id; dkp[(size_t)s] = h->frames[(size_t)s].n_kp; }
    h->bulk_scores_valid = n_pairs;
    size_t found = 0;
    rc = lcm::loop_test_device(h, h->d_bulk_scores, n_pairs, offs.data(), nq, qid.data(), qkp.data(), ns, did.data(), dkp.data(),
                               out ? cap : 0, &found);
    *n_out = found;
    if (rc) return rc;                      // LCM_ERR_CAPACITY: *n_out says how many there are
    // the device compacted them in pair order = (current id, matched id) ascending: nothing to sort
    if (found) {       // on the handle's stream, behind k_loop_emit (a plain hipMemcpy would not wait for a non-blocking stream)
        HIP_TRY(hipMemcpyAsync(out, h->d_cands, sizeof(lcm_loop_candidate) * found, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
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
