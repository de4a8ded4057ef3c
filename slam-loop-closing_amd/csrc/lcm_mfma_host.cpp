// lcm_mfma_host.cpp — host side of the OPT-IN matrix-core variants 4 / 5 (kernels: lcm_mfma.hip): operand images, work items, bulk and online launches.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

namespace lcm {

// ---- matrix-core variants: shared pieces --------------------------------------------------------------------------
struct MfmaRun { uint32_t qf, n_chunks, slot_begin, n_slots, out; };

// Runs -> work items; the query chunks of 8 consecutive runs are interleaved so that workgroups b and b + 8 (same XCD
// under round-robin placement: speed only) stream the same stored frames.
static void mfma_items_from_runs(const std::vector<MfmaRun>& runs, std::vector<lcm::MfmaItem>& items) {
    items.clear();
    for (size_t g = 0; g < runs.size(); g += 8) {
        const size_t nr = std::min<size_t>(8, runs.size() - g);
        uint32_t max_ch = 0;
        for (size_t k = 0; k < nr; ++k) max_ch = std::max(max_ch, runs[g + k].n_chunks);
        for (uint32_t qc = 0; qc < max_ch; ++qc)
            for (size_t k = 0; k < 8; ++k) {
                if (k < nr && qc < runs[g + k].n_chunks) items.push_back({runs[g + k].qf, qc, runs[g + k].slot_begin, runs[g + k].n_slots, runs[g + k].out});
                else items.push_back({0, 0, 0, 0, 0});          // keeps workgroup index mod 8 aligned with the run
            }
    }
}

static hipError_t mfma_expand(lcm_handle* h, const uint32_t* rows, const int32_t* counts, uint32_t stride_words, uint32_t n_frames,
                              uint32_t tiles, uint8_t* img) {
    return h->variant == 5 ? lcm::launch_expand_fp4(rows, counts, stride_words, n_frames, tiles, img, h->stream)
                           : lcm::launch_expand_pm1(rows, counts, stride_words, n_frames, tiles, img, h->stream);
}

// The database's operand image follows the arena INCREMENTALLY: only the frames appended since the last call are
// expanded (an online run appends one frame at a time); anything that moves or drops rows rebuilds it.
static int mfma_db_image(lcm_handle* h) {
    const size_t tile_bytes = h->variant == 5 ? lcm::FP4_TILE_BYTES : lcm::PM1_TILE_BYTES;
    const uint32_t db_tiles = (uint32_t)((h->stride_rows + 31) / 32);
    const size_t n_db = h->frames.size();
    const uint64_t stamp = mix(mix(mix(mix(0x77 + (uint64_t)h->variant, h->db_generation), (uint64_t)h->cap_frames), (uint64_t)h->stride_rows), (uint64_t)(uintptr_t)h->d_rows);
    const size_t need = (size_t)std::max(h->cap_frames, 1) * db_tiles * tile_bytes;
    if (need > h->d_pm1_bytes || !h->d_pm1) h->pm1_stamp = 0;          // a new buffer starts empty
    int rc = ensure_dev(h->d_pm1, h->d_pm1_bytes, need); if (rc) return rc;
    if (h->pm1_stamp != stamp) { h->pm1_frames = 0; h->pm1_stamp = stamp; }
    if (h->pm1_frames < n_db) {
        const size_t first = h->pm1_frames;
        const size_t stride_words = (size_t)h->stride_rows * LCM_DESC_WORDS;
        hipError_t e = mfma_expand(h, (const uint32_t*)h->d_rows + first * stride_words, h->d_counts + first, (uint32_t)stride_words,
                                   (uint32_t)(n_db - first), db_tiles, h->d_pm1 + first * db_tiles * tile_bytes);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "expand kernel launch failed: %s", hipGetErrorString(e));
        h->pm1_frames = n_db;
    }
    return LCM_OK;
}

// Online queries on the matrix cores (variants 4 / 5): B query frames at d_q (query b at row b * pitch_rows) against
// stored slots [0, elig[b]).  Query image + items go up in one staged copy; per-row best distances land in the
// slot's split-mode buffer and k_finalize_pairs folds them exactly as in the vector-ALU split mode.
int mfma_online(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int pitch_rows, int B, const int* nq, const int* elig) {
    const bool fp4 = (h->variant == 5);
    const size_t tile_bytes = fp4 ? lcm::FP4_TILE_BYTES : lcm::PM1_TILE_BYTES;
    const int wg_rows = fp4 ? 512 : 256;
    size_t total = 0;
    for (int b = 0; b < B; ++b) total += (size_t)elig[b];
    int rc = mfma_db_image(h); if (rc) return rc;
    const uint32_t db_tiles = (uint32_t)((h->stride_rows + 31) / 32);
    const uint32_t q_tiles = (uint32_t)((pitch_rows + 31) / 32);
    rc = ensure_dev(h->d_qpm1, h->d_qpm1_bytes, (size_t)B * q_tiles * tile_bytes); if (rc) return rc;
    rc = ensure_dev(q.d_scores, q.d_scores_n, total); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, total); if (rc) return rc;
    rc = ensure_dev(q.d_dist, q.d_dist_n, total * (size_t)lcm::MAX_FUSED_QUERY_ROWS); if (rc) return rc;
    constexpr uint32_t SPI = 4;
    std::vector<MfmaRun> runs;
    uint32_t pair = 0, bat_pair[lcm::MAX_QUERY_BATCH + 1];
    for (int b = 0; b < B; ++b) {
        bat_pair[b] = pair;
        const uint32_t nch = (uint32_t)((nq[b] + wg_rows - 1) / wg_rows);
        for (uint32_t s = 0; s < (uint32_t)elig[b] && nch > 0; s += SPI)
            runs.push_back({(uint32_t)b, nch, s, std::min(SPI, (uint32_t)elig[b] - s), pair + s});
        pair += (uint32_t)elig[b];
    }
    bat_pair[B] = pair;
    std::vector<lcm::MfmaItem> items;
    mfma_items_from_runs(runs, items);
    // staged upload: [row counts of the B queries | items]
    const size_t off_items = 256, up = off_items + items.size() * sizeof(lcm::MfmaItem);
    rc = ensure_pinned(q.h_meta, q.h_meta_bytes, up); if (rc) return rc;          // the slot's own block: no other launch reads it
    rc = ensure_dev(q.d_meta, q.d_meta_bytes, up); if (rc) return rc;
    memcpy(q.h_meta, nq, sizeof(int) * (size_t)B);
    memcpy(q.h_meta + off_items, items.data(), items.size() * sizeof(lcm::MfmaItem));
    HIP_TRY(hipMemcpyAsync(q.d_meta, q.h_meta, up, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    HIP_TRY(hipEventRecord(q.k0, h->stream));
    hipError_t e = mfma_expand(h, d_q, reinterpret_cast<const int32_t*>(q.d_meta), (uint32_t)pitch_rows * LCM_DESC_WORDS, (uint32_t)B, q_tiles, h->d_qpm1);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "expand kernel launch failed: %s", hipGetErrorString(e));
    lcm::MfmaArgs a{};
    a.q_pm1 = h->d_qpm1; a.q_tiles_per_frame = q_tiles; a.q_counts = reinterpret_cast<const int32_t*>(q.d_meta);
    a.db_pm1 = h->d_pm1; a.db_tiles_per_frame = db_tiles; a.db_counts = h->d_counts;
    a.items = reinterpret_cast<const lcm::MfmaItem*>(q.d_meta + off_items);
    a.dist = q.d_dist; a.pair_base = 0;
    e = fp4 ? lcm::launch_score_mfma_fp4(a, (uint32_t)items.size(), h->stream) : lcm::launch_score_mfma(a, (uint32_t)items.size(), h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "MFMA kernel launch failed: %s", hipGetErrorString(e));
    lcm::FinalizeArgs f{};
    f.dist = q.d_dist; f.padded_rows = (uint32_t)lcm::MAX_FUSED_QUERY_ROWS; f.nq = 0;
    f.db_counts = h->d_counts; f.slot_begin = 0; f.scores = q.d_scores;
    f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
    f.n_batch = (uint32_t)B;
    for (int b = 0; b <= B; ++b) f.bat_pair[b] = bat_pair[b];
    for (int b = 0; b < B; ++b) f.bat_nq[b] = nq[b];
    e = lcm::launch_finalize(f, pair, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    HIP_TRY(hipEventRecord(q.k1, h->stream));
    h->info_pending = true;
    h->info.launches = 3; h->info.workgroups = (uint32_t)items.size(); h->info.route = LCM_ROUTE_MATRIX;
    {
        uint64_t dist = 0, bytes = 0;
        for (int b = 0; b < B; ++b) {
            uint64_t prows = 0;
            for (int s2 = 0; s2 < elig[b]; ++s2) prows += (uint64_t)h->frames[(size_t)s2].n;
            dist += (uint64_t)nq[b] * prows; bytes += prows * 32 + (uint64_t)nq[b] * 32 + 8ull * (uint64_t)elig[b];
        }
        h->info.pairs = total; h->info.distances = dist; h->info.algo_bytes = bytes;
        q.acc_pairs = total; q.acc_distances = dist; q.acc_bytes = bytes; q.acc_launches = 3;
    }
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(q.done, h->stream));
    return LCM_OK;
}

int mfma_bulk(lcm_handle* h, bool self, const uint8_t* q_rows, const int32_t* d_q_counts, uint32_t q_pitch_rows,
                     const uint32_t* q_frame_of, const int* nqv, int n_q, const std::vector<size_t>& offsets, lcm_score* d_scores,
                     uint32_t* d_idx_sums) {
    const bool fp4 = (h->variant == 5);                  // 4: int8 operands, 256 query rows per workgroup; 5: fp4, 512
    const size_t tile_bytes = fp4 ? lcm::FP4_TILE_BYTES : lcm::PM1_TILE_BYTES;
    const int wg_rows = fp4 ? 512 : 256;
    const uint32_t db_tiles = (uint32_t)((h->stride_rows + 31) / 32);
    const size_t n_db = h->frames.size();
    int rc = mfma_db_image(h); if (rc) return rc;
    const uint8_t* q_pm1 = h->d_pm1;
    uint32_t q_tiles = db_tiles;
    const int32_t* q_counts_dev = h->d_counts;
    if (!self) {
        uint32_t n_slots = 0;
        for (int c = 0; c < n_q; ++c) n_slots = std::max(n_slots, (q_frame_of ? q_frame_of[c] : (uint32_t)c) + 1);
        q_tiles = (q_pitch_rows + 31) / 32;
        rc = ensure_dev(h->d_qpm1, h->d_qpm1_bytes, (size_t)n_slots * q_tiles * tile_bytes); if (rc) return rc;
        hipError_t e = mfma_expand(h, (const uint32_t*)q_rows, d_q_counts, q_pitch_rows * LCM_DESC_WORDS, n_slots, q_tiles, h->d_qpm1);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "expand kernel launch failed: %s", hipGetErrorString(e));
        q_pm1 = h->d_qpm1;
        q_counts_dev = d_q_counts;
    }
    // ---- per-query metadata for the fold: offsets | nq
    std::vector<uint32_t> meta((size_t)n_q * 2 + 1);
    for (int c = 0; c <= n_q; ++c) meta[(size_t)c] = (uint32_t)offsets[(size_t)c];
    for (int c = 0; c < n_q; ++c) meta[(size_t)n_q + 1 + (size_t)c] = (uint32_t)nqv[c];
    rc = ensure_dev(h->d_mmeta, h->d_mmeta_n, meta.size()); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_mmeta, meta.data(), sizeof(uint32_t) * meta.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    constexpr size_t CHUNK_PAIRS = 524288;               // 4 GiB of per-row best distances per chunk
    constexpr uint32_t SPI = 4;                          // stored frames per work item
    uint32_t launches = 0, biggest = 0;
    uint64_t dist = 0, bytes = 0;
    std::vector<lcm::MfmaItem> items;
    int c0 = 0;
    bool first = true;
    while (c0 < n_q) {
        int c1 = c0;
        size_t pairs = 0;
        while (c1 < n_q && (pairs == 0 || pairs + (offsets[(size_t)c1 + 1] - offsets[(size_t)c1]) <= CHUNK_PAIRS)) { pairs += offsets[(size_t)c1 + 1] - offsets[(size_t)c1]; ++c1; }
        if (pairs > 0) {
            // runs of SPI stored frames per query frame; the 8 query chunks of 8 consecutive runs are interleaved so that
            // workgroup b and b + 8 (same XCD under round-robin placement: speed only) stream the same stored frames
            std::vector<MfmaRun> runs;
            for (int c = c1 - 1; c >= c0; --c) {                 // heaviest query frames first; an empty query frame gets no
                const uint32_t e = (uint32_t)(offsets[(size_t)c + 1] - offsets[(size_t)c]);       // work: the fold writes its records
                const uint32_t nch = (uint32_t)((nqv[c] + wg_rows - 1) / wg_rows);
                for (uint32_t b = 0; b < e && nch > 0; b += SPI)
                    runs.push_back({q_frame_of ? q_frame_of[c] : (uint32_t)c, nch, b, std::min(SPI, e - b), (uint32_t)offsets[(size_t)c] + b});
            }
            mfma_items_from_runs(runs, items);
            rc = ensure_dev(h->d_mdist, h->d_mdist_n, pairs * (size_t)lcm::MAX_FUSED_QUERY_ROWS); if (rc) return rc;
            if (!first) HIP_TRY(hipStreamSynchronize(h->stream));                 // the previous chunk still reads its item list
            rc = ensure_dev(h->d_mitems, h->d_mitems_bytes, items.size() * sizeof(lcm::MfmaItem)); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(h->d_mitems, items.data(), items.size() * sizeof(lcm::MfmaItem), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));                             // `items` is reused by the next chunk
            lcm::MfmaArgs a{};
            a.q_pm1 = q_pm1; a.q_tiles_per_frame = q_tiles; a.q_counts = q_counts_dev;
            a.db_pm1 = h->d_pm1; a.db_tiles_per_frame = db_tiles; a.db_counts = h->d_counts;
            a.items = reinterpret_cast<const lcm::MfmaItem*>(h->d_mitems);
            a.dist = h->d_mdist; a.pair_base = (uint32_t)offsets[(size_t)c0];
            if (d_idx_sums) {           // argmin form: keys out of the kernel (tile of the first best + exact re-scan of that tile)
                a.argmin = 1;
                a.q_rows = (const uint32_t*)q_rows; a.q_stride_words = q_pitch_rows * LCM_DESC_WORDS;
                a.db_rows = (const uint32_t*)h->d_rows; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
            }
            hipError_t e = fp4 ? lcm::launch_score_mfma_fp4(a, (uint32_t)items.size(), h->stream)
                               : lcm::launch_score_mfma(a, (uint32_t)items.size(), h->stream);
            if (e != hipSuccess) return fail(LCM_ERR_HIP, "MFMA kernel launch failed: %s", hipGetErrorString(e));
            lcm::FinalizeBulkArgs f{};
            f.dist = h->d_mdist; f.offsets = h->d_mmeta; f.nq = reinterpret_cast<const int32_t*>(h->d_mmeta + n_q + 1);
            f.db_counts = h->d_counts; f.scores = d_scores; f.n_q = (uint32_t)n_q; f.pair_base = (uint32_t)offsets[(size_t)c0];
            f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
            if (d_idx_sums) { f.key_shift = lcm::KEY_SHIFT; f.idx_sums = d_idx_sums; }
            e = lcm::launch_finalize_bulk(f, (uint32_t)pairs, h->stream);
            if (e != hipSuccess) return fail(LCM_ERR_HIP, "fold kernel launch failed: %s", hipGetErrorString(e));
            launches += 2; biggest = std::max(biggest, (uint32_t)items.size());
            first = false;
        }
        c0 = c1;
    }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    // accounting: same algorithmic definition as the VALU path (packed rows: what the search has to read)
    {
        std::vector<uint64_t> pre(n_db + 1, 0);
        for (size_t s2 = 0; s2 < n_db; ++s2) pre[s2 + 1] = pre[s2] + (uint64_t)h->frames[s2].n;
        for (int c = 0; c < n_q; ++c) {
            const size_t e = offsets[(size_t)c + 1] - offsets[(size_t)c];
            if (e) { dist += (uint64_t)nqv[c] * pre[e]; bytes += pre[e] * 32 + (uint64_t)nqv[c] * 32 + 8ull * e; }
        }
    }
    h->info_pending = true;
    h->info.launches = launches; h->info.workgroups = biggest; h->info.route = LCM_ROUTE_MATRIX;
    h->info.pairs = offsets[(size_t)n_q]; h->info.distances = dist; h->info.algo_bytes = bytes;
    return LCM_OK;
}

}  // namespace lcm
