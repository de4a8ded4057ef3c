// lcm_group.cpp — the frame-sharded multi-GPU loop search behind the C ABI (include/lcm.h, lcm_group_*).
//
// One PROCESS, one lcm_handle per device, one host thread per device for the per-shard work, RCCL (single-process
// ncclCommInitAll communicator) for the exchange steps over xGMI:
//
//   * stored frame with arrival position p is owned by device p mod W (cyclic: balances the triangular all-vs-all and
//     the growing streaming database);
//   * every device needs every frame as a QUERY: the shard arenas are all-gathered (ncclAllGather, the only bulk
//     transfer: N x D x 32 bytes once per search) into a rank-major query buffer per device; work items address
//     query frame p at index (p mod W) * shard_cap + p / W, so no re-packing pass is needed;
//   * each device scores all N query frames against the frames it owns (lcm::all_vs_all on its own stream, planned and
//     launched from its own host thread);
//   * the per-shard 8-byte score records are gathered to device 0 (grouped ncclSend / ncclRecv: a gatherv, shards
//     differ by up to one frame per query), un-permuted there by k_merge_shards into the single-device
//     (query ascending, stored ascending) order, and copied to the host once.
//
// What include/loop_closing.hpp:29-31 would call: a LoopClosingSystem constructed with n_devices > 1 holds an
// lcm_group instead of an lcm_handle.
#include <rccl/rccl.h>

#include <thread>

#include "lcm_internal.h"

namespace {

#define NCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess)                                                                               \
            return fail(LCM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

constexpr int MAX_WORLD = 8;

struct GFrame { int32_t id, n, n_kp; };

}  // namespace

struct lcm_group {
    int world = 0;
    // Rehearsal form (lcm_group_create_loopback): the W shards are W matchers on ONE device and the two exchange steps
    // are device-local copies instead of RCCL calls — every index computation of the multi-device path (cyclic
    // ownership, rank-major query buffer, gatherv offsets, merge) runs for W > 1 on a box with a single GPU.
    bool loopback = false;
    lcm_params params{};
    std::vector<int> devices;
    std::vector<lcm_handle*> h;
    std::vector<ncclComm_t> comms;
    std::vector<GFrame> frames;                 // every frame of every shard, arrival order == ascending id
    // per device: rank-major gathered query rows / counts, this shard's score records
    std::vector<uint8_t*> d_qrows;   std::vector<size_t> d_qrows_bytes;
    std::vector<int32_t*> d_qcounts; std::vector<size_t> d_qcounts_n;
    std::vector<lcm_score*> d_scores; std::vector<size_t> d_scores_n;
    // device 0: gathered shards, merged result, merge metadata
    lcm_score* d_gather = nullptr; size_t d_gather_n = 0;
    lcm_score* d_merged = nullptr; size_t d_merged_n = 0;
    uint32_t* d_meta = nullptr;    size_t d_meta_n = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;   // device 0: search done / gather+merge done / download done
    lcm_group_info info{};
};

namespace {

int set_dev(const lcm_group* g, int r) {
    HIP_TRY(hipSetDevice(g->devices[(size_t)r]));
    return LCM_OK;
}

// eligible stored positions for a query id: positions [0, e) with ids[c] - id >= max(gap, 1)
int eligible_count(const std::vector<GFrame>& f, int query_id, int gap) {
    const long long lim = (long long)query_id - std::max(gap, 1);
    int lo = 0, hi = (int)f.size();
    while (lo < hi) { const int mid = (lo + hi) / 2; if ((long long)f[(size_t)mid].id <= lim) lo = mid + 1; else hi = mid; }
    return lo;
}

// Per-query record offsets of the merged array (offs) and of every shard (offr[r]) for W cyclic shards.
int shard_layout(const std::vector<GFrame>& frames, int W, int gap, std::vector<uint32_t>& offs,
                 std::vector<std::vector<uint32_t>>& offr, uint64_t& total) {
    const int N = (int)frames.size();
    offs.assign((size_t)N + 1, 0);
    offr.assign((size_t)W, std::vector<uint32_t>((size_t)N + 1, 0));
    total = 0;
    for (int c = 0; c < N; ++c) {
        const uint32_t e = (uint32_t)eligible_count(frames, frames[(size_t)c].id, gap);
        total += e;
        if (total > 0xFFFFFFFFull) return fail(LCM_ERR_CAPACITY, "more than 2^32 pairs in one call");
        offs[(size_t)c + 1] = (uint32_t)total;
        for (int r = 0; r < W; ++r)
            offr[(size_t)r][(size_t)c + 1] = offr[(size_t)r][(size_t)c] + (e > (uint32_t)r ? (e - (uint32_t)r + (uint32_t)W - 1) / (uint32_t)W : 0u);
    }
    return LCM_OK;
}

// Enqueue, on `st` (current device), the un-permutation of W back-to-back shard arrays at d_gathered into d_merged.
// d_meta: device scratch of at least (W + 1) * (N + 1) words.
int merge_on_device(const void* d_gathered, void* d_merged, uint32_t* d_meta, const std::vector<uint32_t>& offs,
                    const std::vector<std::vector<uint32_t>>& offr, hipStream_t st) {
    const size_t W = offr.size(), N1 = offs.size();
    lcm::MergeArgs m{};
    m.world = (uint32_t)W; m.n_q = (uint32_t)(N1 - 1); m.n_total = offs.back();
    m.shard_base[0] = 0;
    for (size_t r = 0; r < W; ++r) m.shard_base[r + 1] = m.shard_base[r] + offr[r].back();
    for (size_t r = 0; r < W; ++r)
        HIP_TRY(hipMemcpyAsync(d_meta + r * N1, offr[r].data(), sizeof(uint32_t) * N1, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_meta + W * N1, offs.data(), sizeof(uint32_t) * N1, hipMemcpyHostToDevice, st));
    // the offset vectors are pageable host memory: the copies above have consumed them when the calls return
    m.gathered = d_gathered; m.merged = d_merged;
    m.shard_offsets = d_meta; m.offsets = d_meta + W * N1;
    const hipError_t e = lcm::launch_merge_shards(m, st);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "merge kernel launch failed: %s", hipGetErrorString(e));
    return LCM_OK;
}

}  // namespace

extern "C" {

/* The group's merge step on its own: d_gathered (device) holds the W shard arrays back to back, shard r's
 * shard_counts[r] records in (query ascending, owned stored ascending) order; d_merged (device) receives the
 * single-device order.  Runs k_merge_shards on the handle's stream; lcm_sync(h) before reading the result. */
int lcm_merge_shard_scores_device(lcm_handle* h, const void* d_gathered, const size_t* shard_counts, int world,
                                  const int32_t* ids, int n_frames, int min_gap, void* d_merged, size_t cap, size_t* n_out) {
    if (!h || world < 1 || world > MAX_WORLD || n_frames < 0 || !n_out || !shard_counts || (n_frames > 0 && !ids))
        return fail(LCM_ERR_INVALID_ARG, "bad argument");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(h->device));
        std::vector<GFrame> f((size_t)n_frames);
        for (int i = 0; i < n_frames; ++i) {
            if (i > 0 && ids[i] <= ids[i - 1]) return fail(LCM_ERR_ORDER, "frame ids must be strictly increasing");
            f[(size_t)i] = {ids[i], 0, 0};
        }
        std::vector<uint32_t> offs;
        std::vector<std::vector<uint32_t>> offr;
        uint64_t total = 0;
        int rc = shard_layout(f, world, min_gap, offs, offr, total); if (rc) return rc;
        *n_out = (size_t)total;
        for (int r = 0; r < world; ++r)
            if (shard_counts[r] != offr[(size_t)r].back())
                return fail(LCM_ERR_INVALID_ARG, "shard %d holds %zu records, expected %u", r, shard_counts[r], offr[(size_t)r].back());
        if (!d_merged || total == 0) return LCM_OK;
        if (!d_gathered) return fail(LCM_ERR_INVALID_ARG, "d_gathered is NULL");
        if (cap < total) return fail(LCM_ERR_CAPACITY, "merged array needs %llu records, room for %zu", (unsigned long long)total, cap);
        size_t have = h->d_meta_n;
        int32_t* p = h->d_meta;
        rc = ensure_dev(p, have, (size_t)(world + 1) * ((size_t)n_frames + 1));
        h->d_meta = p; h->d_meta_n = have;
        if (rc) return rc;
        rc = merge_on_device(d_gathered, d_merged, reinterpret_cast<uint32_t*>(h->d_meta), offs, offr, h->stream); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));       // the offset uploads came from this call's stack
        return LCM_OK;
    });
}

/* Host-only: the un-permutation lcm_group_all_vs_all performs on the device (and sharding.merge_shard_scores performs
 * in numpy), for callers that gathered the shards themselves and for the CPU unit tests of the index arithmetic. */
int lcm_merge_shard_scores(const lcm_score* const* shard_scores, const size_t* shard_counts, int world,
                           const int32_t* ids, int n_frames, int min_gap,
                           lcm_score* out, size_t cap, size_t* n_out, size_t* offsets) {
    if (world < 1 || n_frames < 0 || !n_out || (n_frames > 0 && !ids) || !shard_scores || !shard_counts)
        return fail(LCM_ERR_INVALID_ARG, "bad argument");
    return guarded([&]() -> int {
        for (int i = 1; i < n_frames; ++i)
            if (ids[i] <= ids[i - 1]) return fail(LCM_ERR_ORDER, "frame ids must be strictly increasing");
        std::vector<GFrame> f((size_t)n_frames);
        for (int i = 0; i < n_frames; ++i) f[(size_t)i] = {ids[i], 0, 0};
        std::vector<size_t> offs((size_t)n_frames + 1, 0), used((size_t)world, 0);
        for (int c = 0; c < n_frames; ++c) offs[(size_t)c + 1] = offs[(size_t)c] + (size_t)eligible_count(f, ids[c], min_gap);
        const size_t total = offs[(size_t)n_frames];
        *n_out = total;
        if (offsets) memcpy(offsets, offs.data(), sizeof(size_t) * ((size_t)n_frames + 1));
        for (int r = 0; r < world; ++r) {               // every shard must hold exactly its share
            size_t want = 0;
            for (int c = 0; c < n_frames; ++c) {
                const size_t e = offs[(size_t)c + 1] - offs[(size_t)c];
                want += e > (size_t)r ? (e - (size_t)r + (size_t)world - 1) / (size_t)world : 0;
            }
            if (shard_counts[r] != want) return fail(LCM_ERR_INVALID_ARG, "shard %d holds %zu records, expected %zu", r, shard_counts[r], want);
        }
        if (!out) return LCM_OK;                        // sizing call
        if (cap < total) return fail(LCM_ERR_CAPACITY, "merged array needs %zu records, room for %zu", total, cap);
        for (int c = 0; c < n_frames; ++c) {
            const size_t e = offs[(size_t)c + 1] - offs[(size_t)c];
            for (size_t s = 0; s < e; ++s) {
                const size_t r = s % (size_t)world;
                out[offs[(size_t)c] + s] = shard_scores[r][used[r]++];
            }
        }
        return LCM_OK;
    });
}

static int group_create(const lcm_params* params, int n_devices, const int* device_ids, bool loopback, int loop_device, lcm_group** out) {
    if (!out) return fail(LCM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices < 1 || n_devices > MAX_WORLD) return fail(LCM_ERR_INVALID_ARG, "n_devices must be 1..%d", MAX_WORLD);
    const int have = lcm_device_count();
    if (have <= 0) {
        lcm_handle* probe = nullptr;
        return lcm_create(params, 0, nullptr, &probe);        // "no HIP device ... no CPU fallback"
    }
    return guarded([&]() -> int {
        lcm_group* g = new lcm_group();
        auto bail = [&](int rc) { const std::string why = lcm::last_error(); lcm_group_destroy(g); lcm::last_error() = why; return rc; };
        g->world = n_devices;
        g->loopback = loopback;
        lcm_params_default(&g->params);
        if (params) g->params = *params;
        for (int r = 0; r < n_devices; ++r) {
            const int dev = loopback ? loop_device : (device_ids ? device_ids[r] : r);
            if (dev < 0 || dev >= have) return bail(fail(LCM_ERR_INVALID_ARG, "device id %d out of range [0,%d)", dev, have));
            if (!loopback) for (int d : g->devices) if (d == dev) return bail(fail(LCM_ERR_INVALID_ARG, "device id %d listed twice", dev));
            g->devices.push_back(dev);
        }
        const size_t W = (size_t)n_devices;
        g->h.assign(W, nullptr);
        g->d_qrows.assign(W, nullptr); g->d_qrows_bytes.assign(W, 0);
        g->d_qcounts.assign(W, nullptr); g->d_qcounts_n.assign(W, 0);
        g->d_scores.assign(W, nullptr); g->d_scores_n.assign(W, 0);
        for (int r = 0; r < n_devices; ++r) {
            const int rc = lcm_create(&g->params, g->devices[(size_t)r], nullptr, &g->h[(size_t)r]);
            if (rc) return bail(rc);
        }
        if (!loopback) {
            g->comms.assign(W, nullptr);
            ncclResult_t nr = ncclCommInitAll(g->comms.data(), n_devices, g->devices.data());
            if (nr != ncclSuccess) { g->comms.clear(); return bail(fail(LCM_ERR_HIP, "ncclCommInitAll failed: %s", ncclGetErrorString(nr))); }
        }
        if (hipSetDevice(g->devices[0]) != hipSuccess || hipEventCreate(&g->ev0) != hipSuccess ||
            hipEventCreate(&g->ev1) != hipSuccess || hipEventCreate(&g->ev2) != hipSuccess)
            return bail(fail(LCM_ERR_HIP, "hipEventCreate failed"));
        *out = g;
        return LCM_OK;
    });
}

int lcm_group_create(const lcm_params* params, int n_devices, const int* device_ids, lcm_group** out) {
    return group_create(params, n_devices, device_ids, false, 0, out);
}

int lcm_group_create_loopback(const lcm_params* params, int n_shards, int device_id, lcm_group** out) {
    return group_create(params, n_shards, nullptr, true, device_id, out);
}

void lcm_group_destroy(lcm_group* g) {
    if (!g) return;
    for (size_t r = 0; r < g->h.size(); ++r) {
        if (!g->h[r]) continue;
        (void)hipSetDevice(g->devices[r]);
        (void)hipDeviceSynchronize();
        if (r < g->d_qrows.size()) { (void)hipFree(g->d_qrows[r]); (void)hipFree(g->d_qcounts[r]); (void)hipFree(g->d_scores[r]); }
    }
    if (!g->devices.empty()) {
        (void)hipSetDevice(g->devices[0]);
        (void)hipFree(g->d_gather); (void)hipFree(g->d_merged); (void)hipFree(g->d_meta);
        if (g->ev0) (void)hipEventDestroy(g->ev0);
        if (g->ev1) (void)hipEventDestroy(g->ev1);
        if (g->ev2) (void)hipEventDestroy(g->ev2);
    }
    for (ncclComm_t c : g->comms) if (c) (void)ncclCommDestroy(c);
    for (lcm_handle* h : g->h) lcm_destroy(h);
    delete g;
}

int lcm_group_size(const lcm_group* g) { return g ? g->world : 0; }
int lcm_group_db_size(const lcm_group* g) { return g ? (int)g->frames.size() : 0; }

int lcm_group_handle(lcm_group* g, int rank, lcm_handle** out) {
    if (!g || !out || rank < 0 || rank >= g->world) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *out = g->h[(size_t)rank];
    return LCM_OK;
}

int lcm_group_set_params(lcm_group* g, const lcm_params* p) {
    if (!g || !p) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    for (lcm_handle* h : g->h) { const int rc = lcm_set_params(h, p); if (rc) return rc; }
    g->params = *p;
    return LCM_OK;
}

int lcm_group_reserve(lcm_group* g, int n_frames, int max_desc) {
    if (!g || n_frames < 0 || max_desc < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    const int per = (n_frames + g->world - 1) / g->world;
    for (lcm_handle* h : g->h) { const int rc = lcm_db_reserve(h, per, max_desc); if (rc) return rc; }
    return LCM_OK;
}

int lcm_group_append(lcm_group* g, int frame_id, const uint8_t* desc, int n, int n_keypoints) {
    if (!g || n < 0 || (n > 0 && !desc)) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    return guarded([&]() -> int {
        if (!g->frames.empty() && frame_id <= g->frames.back().id)
            return fail(LCM_ERR_ORDER, "frame id %d appended after id %d: ids must be strictly increasing", frame_id, g->frames.back().id);
        const size_t owner = g->frames.size() % (size_t)g->world;      // cyclic ownership by arrival position
        const int rc = lcm_db_append(g->h[owner], frame_id, desc, n, n_keypoints);
        if (rc) return rc;
        g->frames.push_back({frame_id, n, n_keypoints < 0 ? n : n_keypoints});
        return LCM_OK;
    });
}

int lcm_group_clear(lcm_group* g) {
    if (!g) return fail(LCM_ERR_INVALID_ARG, "NULL group");
    for (lcm_handle* h : g->h) { const int rc = lcm_db_clear(h); if (rc) return rc; }
    g->frames.clear();
    return LCM_OK;
}

int lcm_group_all_vs_all(lcm_group* g, lcm_score* out_scores, size_t cap, size_t* n_pairs, size_t* pair_offsets) {
    if (!g || !n_pairs) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    return guarded([&]() -> int {
        const int W = g->world;
        const int N = (int)g->frames.size();
        // ---- bookkeeping: merged offsets, per-shard per-query offsets
        std::vector<uint32_t> offs;
        std::vector<std::vector<uint32_t>> offr;
        uint64_t total = 0;
        { const int rc0 = shard_layout(g->frames, W, g->params.min_gap, offs, offr, total); if (rc0) return rc0; }
        *n_pairs = (size_t)total;
        if (pair_offsets) for (int c = 0; c <= N; ++c) pair_offsets[c] = offs[(size_t)c];
        if (!out_scores) return LCM_OK;                     // sizing call
        if (cap < total) return fail(LCM_ERR_CAPACITY, "scores buffer holds %zu records, need %llu", cap, (unsigned long long)total);
        g->info = lcm_group_info{};
        g->info.n_devices = W;
        g->info.pairs = total;
        if (total == 0) return LCM_OK;

        // ---- 1. equal shard geometry on every device, then the all-gather of the shard arenas into query buffers
        const int shard_cap = (N + W - 1) / W;
        int stride = 4;
        for (lcm_handle* h : g->h) stride = std::max(stride, h->stride_rows);
        for (const GFrame& f : g->frames) stride = std::max(stride, (f.n + 3) / 4 * 4);
        for (int r = 0; r < W; ++r) {
            int rc = lcm_db_reserve(g->h[(size_t)r], shard_cap, stride); if (rc) return rc;
            rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc;               // every append has landed
            if (g->h[(size_t)r]->stride_rows != stride || g->h[(size_t)r]->cap_frames < shard_cap)
                return fail(LCM_ERR_HIP, "shard %d arena geometry mismatch", r);
        }
        const size_t shard_bytes = (size_t)shard_cap * (size_t)stride * LCM_DESC_BYTES;
        for (int r = 0; r < W; ++r) {
            int rc = set_dev(g, r); if (rc) return rc;
            rc = ensure_dev(g->d_qrows[(size_t)r], g->d_qrows_bytes[(size_t)r], shard_bytes * (size_t)W, 512); if (rc) return rc;
            rc = ensure_dev(g->d_qcounts[(size_t)r], g->d_qcounts_n[(size_t)r], (size_t)shard_cap * (size_t)W); if (rc) return rc;
        }
        if (g->loopback) {
            // what ncclAllGather delivers, as device-local copies (every shard was synchronised above)
            for (int r = 0; r < W; ++r)
                for (int s = 0; s < W; ++s) {
                    HIP_TRY(hipMemcpyAsync(g->d_qrows[(size_t)r] + (size_t)s * shard_bytes, g->h[(size_t)s]->d_rows, shard_bytes, hipMemcpyDeviceToDevice, g->h[(size_t)r]->stream));
                    HIP_TRY(hipMemcpyAsync(g->d_qcounts[(size_t)r] + (size_t)s * (size_t)shard_cap, g->h[(size_t)s]->d_counts, sizeof(int32_t) * (size_t)shard_cap, hipMemcpyDeviceToDevice, g->h[(size_t)r]->stream));
                }
        } else {
            NCCL_TRY(ncclGroupStart());
            for (int r = 0; r < W; ++r) {
                lcm_handle* h = g->h[(size_t)r];
                NCCL_TRY(ncclAllGather(h->d_rows, g->d_qrows[(size_t)r], shard_bytes, ncclUint8, g->comms[(size_t)r], h->stream));
                NCCL_TRY(ncclAllGather(h->d_counts, g->d_qcounts[(size_t)r], (size_t)shard_cap, ncclInt32, g->comms[(size_t)r], h->stream));
            }
            NCCL_TRY(ncclGroupEnd());
        }
        g->info.gathered_query_bytes = (uint64_t)shard_bytes * (uint64_t)W;

        // ---- 2. per-shard search: one host thread per device plans + launches on that device's stream
        std::vector<int32_t> ids((size_t)N), counts((size_t)N);
        std::vector<uint32_t> q_frame_of((size_t)N);
        for (int p = 0; p < N; ++p) {
            ids[(size_t)p] = g->frames[(size_t)p].id;
            counts[(size_t)p] = g->frames[(size_t)p].n;
            q_frame_of[(size_t)p] = (uint32_t)((p % W) * shard_cap + p / W);
        }
        // device 0's shard is written straight into the gather buffer (it is the first segment)
        int rc = set_dev(g, 0); if (rc) return rc;
        rc = ensure_dev(g->d_gather, g->d_gather_n, (size_t)total); if (rc) return rc;
        rc = ensure_dev(g->d_merged, g->d_merged_n, (size_t)total); if (rc) return rc;
        std::vector<int> rcs((size_t)W, LCM_OK);
        std::vector<std::string> errs((size_t)W);
        std::vector<size_t> n_shard((size_t)W, 0);
        for (int r = 1; r < W; ++r) {
            rc = set_dev(g, r); if (rc) return rc;
            rc = ensure_dev(g->d_scores[(size_t)r], g->d_scores_n[(size_t)r], (size_t)offr[(size_t)r][(size_t)N]); if (rc) return rc;
        }
        auto shard_job = [&](int r) {
            lcm_handle* h = g->h[(size_t)r];
            void* dst = r == 0 ? (void*)g->d_gather : (void*)g->d_scores[(size_t)r];
            size_t n = 0;
            const int rc2 = lcm::all_vs_all(h, g->d_qrows[(size_t)r], g->d_qcounts[(size_t)r], ids.data(), N, stride, dst,
                                            (size_t)offr[(size_t)r][(size_t)N], &n, nullptr, nullptr, q_frame_of.data(), counts.data());
            rcs[(size_t)r] = rc2;
            n_shard[(size_t)r] = n;
            if (rc2) errs[(size_t)r] = lcm::last_error();        // thread-local in the worker: hand it to the caller
        };
        {
            std::vector<std::thread> workers;
            for (int r = 1; r < W; ++r) workers.emplace_back(shard_job, r);
            shard_job(0);
            for (std::thread& t : workers) t.join();
        }
        for (int r = 0; r < W; ++r) {
            if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "shard %d: %s", r, errs[(size_t)r].c_str());
            if (n_shard[(size_t)r] != offr[(size_t)r][(size_t)N]) return fail(LCM_ERR_HIP, "shard %d scored %zu pairs, expected %u", r, n_shard[(size_t)r], offr[(size_t)r][(size_t)N]);
        }

        // ---- 3. gather the shards' records to device 0 (gatherv by grouped send / recv), merge there, one download
        rc = set_dev(g, 0); if (rc) return rc;
        HIP_TRY(hipEventRecord(g->ev0, g->h[0]->stream));
        std::vector<uint32_t> shard_base((size_t)W + 1, 0);
        for (int r = 0; r < W; ++r) shard_base[(size_t)r + 1] = shard_base[(size_t)r] + offr[(size_t)r][(size_t)N];
        if (W > 1 && g->loopback) {
            // what the grouped ncclSend / ncclRecv delivers: each shard's records behind device 0's
            for (int r = 1; r < W; ++r) {
                const size_t bytes = (size_t)offr[(size_t)r][(size_t)N] * sizeof(lcm_score);
                if (!bytes) continue;
                rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc;
                HIP_TRY(hipMemcpyAsync(g->d_gather + shard_base[(size_t)r], g->d_scores[(size_t)r], bytes, hipMemcpyDeviceToDevice, g->h[0]->stream));
            }
        } else if (W > 1) {
            NCCL_TRY(ncclGroupStart());
            for (int r = 1; r < W; ++r) {
                const size_t bytes = (size_t)offr[(size_t)r][(size_t)N] * sizeof(lcm_score);
                if (!bytes) continue;
                NCCL_TRY(ncclSend(g->d_scores[(size_t)r], bytes, ncclUint8, 0, g->comms[(size_t)r], g->h[(size_t)r]->stream));
                NCCL_TRY(ncclRecv(g->d_gather + shard_base[(size_t)r], bytes, ncclUint8, r, g->comms[0], g->h[0]->stream));
            }
            NCCL_TRY(ncclGroupEnd());
        }
        rc = ensure_dev(g->d_meta, g->d_meta_n, (size_t)(W + 1) * ((size_t)N + 1)); if (rc) return rc;
        rc = merge_on_device(g->d_gather, g->d_merged, g->d_meta, offs, offr, g->h[0]->stream); if (rc) return rc;
        HIP_TRY(hipEventRecord(g->ev1, g->h[0]->stream));
        HIP_TRY(hipMemcpyAsync(out_scores, g->d_merged, sizeof(lcm_score) * (size_t)total, hipMemcpyDeviceToHost, g->h[0]->stream));
        HIP_TRY(hipEventRecord(g->ev2, g->h[0]->stream));
        for (int r = 0; r < W; ++r) { rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc; }

        // ---- timings (device clocks): slowest shard kernel, gather + merge, download
        for (int r = 0; r < W; ++r) {
            lcm_launch_info li{};
            rc = lcm_last_launch_info(g->h[(size_t)r], &li); if (rc) return rc;
            g->info.kernel_ms_max = std::max(g->info.kernel_ms_max, li.kernel_ms);
            g->info.distances += li.distances;
            g->info.algo_bytes += li.algo_bytes;
        }
        rc = set_dev(g, 0); if (rc) return rc;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1)); g->info.gather_merge_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev1, g->ev2)); g->info.download_ms = ms;
        g->info.gathered_score_bytes = (uint64_t)(total - offr[0][(size_t)N]) * sizeof(lcm_score);
        return LCM_OK;
    });
}

int lcm_group_last_info(const lcm_group* g, lcm_group_info* info) {
    if (!g || !info) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *info = g->info;
    return LCM_OK;
}

/* Online query: the frame goes to every device (pinned staging + H2D per device, all enqueued before any is awaited),
 * each scores it against its shard, and the per-shard records — a few KB — are interleaved on the host: for a
 * message this small a collective would only add latency. */
int lcm_group_query_scores(lcm_group* g, const uint8_t* query, int nq, int query_frame_id,
                           lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out) {
    if (!g || !n_out || nq < 0 || (nq > 0 && !query)) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    return guarded([&]() -> int {
        const int W = g->world;
        const int e = eligible_count(g->frames, query_frame_id, g->params.min_gap);
        if (e > cap) return fail(LCM_ERR_CAPACITY, "%d score records but room for %d", e, cap);
        if (e > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL");
        std::vector<int> tickets((size_t)W, -1);
        for (int r = 0; r < W; ++r) {
            const int rc = lcm_query_submit(g->h[(size_t)r], query, nq, query_frame_id, &tickets[(size_t)r]);
            if (rc) {                       // drain what was already submitted, then report
                const std::string why = lcm::last_error();
                std::vector<lcm_score> sink((size_t)std::max(e, 1));
                int n = 0;
                for (int q = 0; q < r; ++q) (void)lcm_query_collect(g->h[(size_t)q], tickets[(size_t)q], sink.data(), nullptr, (int)sink.size(), &n);
                lcm::last_error() = why;
                return rc;
            }
        }
        std::vector<lcm_score> part((size_t)std::max((e + W - 1) / W, 1));
        int rc_all = LCM_OK;
        for (int r = 0; r < W; ++r) {
            int n = 0;
            const int rc = lcm_query_collect(g->h[(size_t)r], tickets[(size_t)r], part.data(), nullptr, (int)part.size(), &n);
            if (rc) { rc_all = rc; continue; }
            const int want = e > r ? (e - r + W - 1) / W : 0;
            if (n != want) { rc_all = fail(LCM_ERR_HIP, "shard %d returned %d records, expected %d", r, n, want); continue; }
            for (int k = 0; k < n; ++k) out_scores[(size_t)r + (size_t)k * (size_t)W] = part[(size_t)k];
        }
        if (rc_all) return rc_all;
        if (out_frame_ids) for (int s = 0; s < e; ++s) out_frame_ids[s] = g->frames[(size_t)s].id;
        *n_out = e;
        return LCM_OK;
    });
}

/* Micro-batched online queries over all shards (cfg5 shape in one process): the batch goes to every device with ONE
 * lcm_query_submit_batch each (all enqueued before any is awaited), the per-shard records are interleaved on the host
 * into the single-device order, query 0's records first.  offsets: n_queries + 1 entries (optional). */
int lcm_group_query_scores_batch(lcm_group* g, const uint8_t* const* queries, const int* nq, const int* query_frame_ids, int n_queries,
                                 lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets) {
    if (!g || !n_out || !queries || !nq || !query_frame_ids || n_queries < 1) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    return guarded([&]() -> int {
        const int W = g->world;
        std::vector<size_t> offs((size_t)n_queries + 1, 0);
        std::vector<int> e((size_t)n_queries);
        for (int b = 0; b < n_queries; ++b) {
            e[(size_t)b] = eligible_count(g->frames, query_frame_ids[b], g->params.min_gap);
            offs[(size_t)b + 1] = offs[(size_t)b] + (size_t)e[(size_t)b];
        }
        const size_t total = offs.back();
        if (offsets) memcpy(offsets, offs.data(), sizeof(size_t) * offs.size());
        if (total > cap) return fail(LCM_ERR_CAPACITY, "%zu score records but room for %zu", total, cap);
        if (total > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL");
        std::vector<int> tickets((size_t)W, -1);
        int rc_all = LCM_OK;
        std::string why;
        for (int r = 0; r < W && !rc_all; ++r) {
            rc_all = lcm_query_submit_batch(g->h[(size_t)r], queries, nq, query_frame_ids, n_queries, &tickets[(size_t)r]);
            if (rc_all) why = lcm::last_error();
        }
        std::vector<lcm_score> part(std::max<size_t>((total + (size_t)W - 1) / (size_t)W + (size_t)n_queries, 1));
        std::vector<size_t> poffs((size_t)n_queries + 1);
        for (int r = 0; r < W; ++r) {
            if (tickets[(size_t)r] < 0) continue;
            size_t n = 0;
            const int rc = lcm_query_collect_batch(g->h[(size_t)r], tickets[(size_t)r], part.data(), part.size(), &n, poffs.data());
            if (rc) { if (!rc_all) { rc_all = rc; why = lcm::last_error(); } continue; }
            if (rc_all) continue;                         // a submit failed: this collect only drained the ticket
            for (int b = 0; b < n_queries; ++b) {
                const int want = e[(size_t)b] > r ? (e[(size_t)b] - r + W - 1) / W : 0;
                if ((int)(poffs[(size_t)b + 1] - poffs[(size_t)b]) != want) { rc_all = fail(LCM_ERR_HIP, "shard %d returned a wrong record count for query %d", r, b); why = lcm::last_error(); break; }
                for (int k = 0; k < want; ++k) out_scores[offs[(size_t)b] + (size_t)r + (size_t)k * (size_t)W] = part[poffs[(size_t)b] + (size_t)k];
            }
        }
        if (rc_all) { lcm::last_error() = why; return rc_all; }
        *n_out = total;
        return LCM_OK;
    });
}

int lcm_group_detect_loops(lcm_group* g, int current_frame_id, const uint8_t* query, int nq, int n_keypoints,
                           lcm_loop_candidate* out, int cap, int* n_out) {
    if (!g || !n_out || cap < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    return guarded([&]() -> int {
        const int e = eligible_count(g->frames, current_frame_id, g->params.min_gap);
        std::vector<lcm_score> sc((size_t)std::max(e, 1));
        int n = 0;
        const int rc = lcm_group_query_scores(g, query, nq, current_frame_id, sc.data(), nullptr, (int)sc.size(), &n);
        if (rc) return rc;
        const int q_kp = n_keypoints < 0 ? nq : n_keypoints;
        int k = 0, total = 0;
        for (int s = 0; s < n; ++s) {
            double sim;
            if (lcm_loop_test(&g->params, &sc[(size_t)s], q_kp, g->frames[(size_t)s].n_kp, &sim)) {
                if (k < cap && out) {
                    out[k].current_frame_id = current_frame_id;
                    out[k].matched_frame_id = g->frames[(size_t)s].id;
                    out[k].num_matches = (int32_t)sc[(size_t)s].good_count;
                    out[k].similarity_score = sim;
                    ++k;
                }
                ++total;
            }
        }
        *n_out = k;
        if (total > k) return fail(LCM_ERR_CAPACITY, "%d loop candidates but room for %d", total, cap);
        return LCM_OK;
    });
}

}  // extern "C"
