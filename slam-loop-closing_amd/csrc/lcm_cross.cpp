// lcm_cross.cpp — cross_check scoring of (query, stored prefix) pairs: forward + role-swapped backward keys, k_cross_score on the device.
// Part of liblcm_hip.so's host side (C ABI in include/lcm.h); shared state and helpers: lcm_internal.h.
#include "lcm_internal.h"

namespace lcm {

// cross_check scoring of "query c against stored slots [0, elig[c])" for n_q queries, records written to d_scores in
// (query, slot) order starting at index 0.  Query c is nq[c] rows starting at row q_row0[c] of the matrix at d_qbase,
// padded for the train role.  Per chunk of pairs (bounded key scratch): forward keys (query rows -> stored frame),
// backward keys (stored rows -> query frame: roles swapped), k_cross_score folds both into the score record on the device.
int cross_score_prefixes(lcm_handle* h, const uint8_t* d_qbase, const uint32_t* q_row0, const int* nq, const int* elig,
                                int n_q, lcm_score* d_scores, uint32_t* d_idx_sums) {
    const int CH = lcm::MAX_FUSED_QUERY_ROWS;
    if ((size_t)h->cap_frames * (size_t)h->stride_rows >= 0xFFFFFFFFull)
        return fail(LCM_ERR_CAPACITY, "the database arena exceeds 2^32 rows: pair items address rows with 32 bits");
    constexpr size_t SLOT_BUDGET = 49152;                 // key slots of 8 KB per chunk: 384 MB of scratch
    std::vector<lcm::PairItem> fitems, bitems;
    std::vector<lcm::CrossDesc> descs;
    size_t slots = 0;
    int max_fq = 0, max_bq = 0;
    uint64_t dist = 0, bytes = 0, pairs = 0;
    uint32_t out = 0, launches = 0;
    auto flush = [&]() -> int {
        if (descs.empty()) return LCM_OK;
        const size_t off_b = sizeof(lcm::PairItem) * fitems.size();
        const size_t off_d = off_b + sizeof(lcm::PairItem) * bitems.size();
        const size_t up = off_d + sizeof(lcm::CrossDesc) * descs.size();
        int rc = ensure_pinned(h->h_pair_stage, h->h_pair_stage_bytes, up); if (rc) return rc;
        rc = ensure_dev(h->d_pair_stage, h->d_pair_stage_bytes, up, ARENA_SLACK); if (rc) return rc;
        rc = ensure_dev(h->d_keys, h->d_keys_n, slots * (size_t)CH); if (rc) return rc;
        memcpy(h->h_pair_stage, fitems.data(), off_b);
        memcpy(h->h_pair_stage + off_b, bitems.data(), off_d - off_b);
        memcpy(h->h_pair_stage + off_d, descs.data(), up - off_d);
        HIP_TRY(hipMemcpyAsync(h->d_pair_stage, h->h_pair_stage, up, hipMemcpyHostToDevice, h->stream));
        lcm::ScoreArgs a{};
        a.scores = nullptr; a.keys = h->d_keys; a.keys_stride = CH;
        a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
        a.q_rows = (const uint32_t*)d_qbase; a.db_rows = (const uint32_t*)h->d_rows;
        a.pair_items = reinterpret_cast<const lcm::PairItem*>(h->d_pair_stage);
        hipError_t e = lcm::launch_score(a, (uint32_t)fitems.size(), max_fq, true, 0, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        a.q_rows = (const uint32_t*)h->d_rows; a.db_rows = (const uint32_t*)d_qbase;
        a.pair_items = reinterpret_cast<const lcm::PairItem*>(h->d_pair_stage + off_b);
        e = lcm::launch_score(a, (uint32_t)bitems.size(), max_bq, true, 0, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::CrossArgs c{};
        c.keys = h->d_keys; c.descs = reinterpret_cast<const lcm::CrossDesc*>(h->d_pair_stage + off_d);
        c.scores = d_scores; c.idx_sums = d_idx_sums;
        c.mode = h->params.cross_check; c.ratio = h->params.ratio; c.dist_floor = h->params.dist_floor;
        e = lcm::launch_cross_score(c, (uint32_t)descs.size(), h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "cross-check kernel launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(h->stream));        // the staging block is rewritten by the next chunk
        launches += 3;
        fitems.clear(); bitems.clear(); descs.clear(); slots = 0; max_fq = max_bq = 0;
        return LCM_OK;
    };
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    for (int c = 0; c < n_q; ++c) {
        if (nq[c] > CH) return fail(LCM_ERR_CAPACITY, "a query frame may hold at most %d rows", CH);
        bytes += (uint64_t)nq[c] * 32;
        for (int s = 0; s < elig[c]; ++s) {
            const int nt = h->frames[(size_t)s].n;
            const size_t need = 1 + (size_t)((nt + CH - 1) / CH);
            if (slots + need > SLOT_BUDGET) { const int rc = flush(); if (rc) return rc; }
            const uint32_t t_row0 = (uint32_t)((size_t)s * (size_t)h->stride_rows);
            lcm::CrossDesc d{(uint32_t)slots, (uint32_t)slots + 1, (uint32_t)nq[c], (uint32_t)nt, out++};
            fitems.push_back({q_row0[c], t_row0, (uint32_t)nq[c] | ((uint32_t)nt << 12), (uint32_t)slots});
            for (int k = 0; k * CH < nt; ++k)
                bitems.push_back({t_row0 + (uint32_t)(k * CH), q_row0[c], (uint32_t)std::min(CH, nt - k * CH) | ((uint32_t)nq[c] << 12), (uint32_t)(slots + 1 + (size_t)k)});
            descs.push_back(d);
            slots += need;
            max_fq = std::max(max_fq, nq[c]); max_bq = std::max(max_bq, std::min(CH, nt));
            dist += 2ull * (uint64_t)nq[c] * (uint64_t)nt; bytes += 2ull * (uint64_t)nt * 32 + 8; ++pairs;
        }
    }
    { const int rc = flush(); if (rc) return rc; }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = launches; h->info.workgroups = 0; h->info.route = LCM_ROUTE_CROSS;
    h->info.pairs = pairs; h->info.distances = dist; h->info.algo_bytes = bytes;
    return LCM_OK;
}

}  // namespace lcm
