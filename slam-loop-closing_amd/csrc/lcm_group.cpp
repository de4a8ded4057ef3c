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

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
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

namespace lcm {
// One persistent host thread per device beyond the first (the caller's thread serves device 0): per-shard planning,
// launches, online submits / collects and downloads run on all devices at once, without a thread being created per call.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    Worker() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || stop; });
                if (stop) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j); has_job = true; done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
    }
    ~Worker() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

// One asynchronous group query (lcm_group_query_submit_batch): a ticket per shard + what the interleave needs.
struct GTicket {
    bool busy = false;
    int n_queries = 0;
    int e[lcm::MAX_QUERY_BATCH] = {0};          // eligible stored frames per query, over ALL shards, at submit time
    size_t total = 0;
    uint64_t stamp = 0;                         // group's drop_stamp at submit: a clear / truncate in between voids the ticket
    std::vector<int> shard_ticket;
    std::vector<std::vector<lcm_score>> part;   // per shard: landing zone of its records
};
}  // namespace lcm
using lcm::GTicket;
using lcm::Worker;

struct lcm_group {
    int world = 0;
    // Rehearsal form (lcm_group_create_loopback): the W shards are W matchers on ONE device and the two exchange steps
    // are device-local copies instead of RCCL calls — every index computation of the multi-device path (cyclic
    // ownership, rank-major query buffer, gatherv offsets, merge) runs for W > 1 on a box with a single GPU.
    bool loopback = false;
    // Exchange steps as device-to-device copies (hipMemcpyAsync between the devices' arenas: xGMI peer copies) instead of
    // RCCL calls: always in the loopback form; in a real group when the caller asked for it (lcm_group_create_peer) or
    // when the RCCL communicator could not be created (the reason is kept in rccl_error).
    bool copies = false;
    std::string rccl_error;
    lcm_params params{};
    std::vector<int> devices;
    std::vector<lcm_handle*> h;
    std::vector<ncclComm_t> comms;
    std::vector<std::unique_ptr<Worker>> workers;   // W - 1: worker k serves shard k + 1
    std::vector<GFrame> frames;                 // every frame of every shard, arrival order == ascending id
    // per device: rank-major gathered query rows / counts, this shard's score records (+ index checksums)
    std::vector<uint8_t*> d_qrows;   std::vector<size_t> d_qrows_bytes;
    std::vector<int32_t*> d_qcounts; std::vector<size_t> d_qcounts_n;
    std::vector<lcm_score*> d_scores; std::vector<size_t> d_scores_n;
    std::vector<uint32_t*> d_isums;  std::vector<size_t> d_isums_n;
    // The gathered query buffers stay valid until the database changes: db_stamp is bumped by every append / clear /
    // truncate, gathered_stamp is the value it had when the arenas were last all-gathered (with that geometry).
    uint64_t db_stamp = 1, gathered_stamp = 0, drop_stamp = 1;
    int gathered_cap = 0, gathered_stride = 0;
    // device 0: gathered shards, merged result, merge metadata
    lcm_score* d_gather = nullptr; size_t d_gather_n = 0;
    lcm_score* d_merged = nullptr; size_t d_merged_n = 0;
    uint32_t* d_gather_idx = nullptr; size_t d_gather_idx_n = 0;
    uint32_t* d_merged_idx = nullptr; size_t d_merged_idx_n = 0;
    uint32_t* d_meta = nullptr;    size_t d_meta_n = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;   // device 0: search done / gather+merge done / download done
    hipEvent_t evg0 = nullptr, evg1 = nullptr;                // device 0: around the all-gather of the shard arenas
    GTicket tickets[lcm::QUERY_SLOTS];
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

// Enqueue, on `st` (current device), the upload of the merge metadata: d_meta (device scratch of at least
// (W + 1) * (N + 1) words) receives the W per-shard offset arrays and the merged offsets.
int upload_merge_meta(uint32_t* d_meta, const std::vector<uint32_t>& offs, const std::vector<std::vector<uint32_t>>& offr, hipStream_t st) {
    const size_t W = offr.size(), N1 = offs.size();
    for (size_t r = 0; r < W; ++r)
        HIP_TRY(hipMemcpyAsync(d_meta + r * N1, offr[r].data(), sizeof(uint32_t) * N1, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_meta + W * N1, offs.data(), sizeof(uint32_t) * N1, hipMemcpyHostToDevice, st));
    // the offset vectors are pageable host memory: the copies above have consumed them when the calls return
    return LCM_OK;
}

// Enqueue the un-permutation of W back-to-back shard arrays at d_gathered into d_merged (elements of elem_words dwords:
// 2 = score records, 1 = index checksums); d_meta as left by upload_merge_meta on the same stream.
int merge_on_device(const void* d_gathered, void* d_merged, const uint32_t* d_meta, const std::vector<uint32_t>& offs,
                    const std::vector<std::vector<uint32_t>>& offr, uint32_t elem_words, hipStream_t st) {
    const size_t W = offr.size(), N1 = offs.size();
    lcm::MergeArgs m{};
    m.world = (uint32_t)W; m.n_q = (uint32_t)(N1 - 1); m.n_total = offs.back(); m.elem_words = elem_words;
    m.shard_base[0] = 0;
    for (size_t r = 0; r < W; ++r) m.shard_base[r + 1] = m.shard_base[r] + offr[r].back();
    m.gathered = d_gathered; m.merged = d_merged;
    m.shard_offsets = d_meta; m.offsets = d_meta + W * N1;
    const hipError_t e = lcm::launch_merge_shards(m, st);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "merge kernel launch failed: %s", hipGetErrorString(e));
    return LCM_OK;
}

// Run fn(r) for every shard at once — shard 0 on the calling thread, shard r > 0 on its worker — and wait for all.
// Returns the first failure, its message prefixed with the shard (a worker's lcm_last_error is thread-local).
template <typename F>
int run_all(lcm_group* g, F&& fn) {
    const int W = g->world;
    std::vector<int> rcs((size_t)W, LCM_OK);
    std::vector<std::string> errs((size_t)W);
    auto job = [&](int r) {
        int rc;
        try { rc = fn(r); }
        catch (const std::bad_alloc&) { rc = fail(LCM_ERR_OOM, "host allocation failed (std::bad_alloc)"); }
        catch (const std::exception& e) { rc = fail(LCM_ERR_HIP, "unexpected C++ exception: %s", e.what()); }
        catch (...) { rc = fail(LCM_ERR_HIP, "unexpected C++ exception"); }
        rcs[(size_t)r] = rc;
        if (rc) errs[(size_t)r] = lcm::last_error();
    };
    for (int r = 1; r < W; ++r) g->workers[(size_t)r - 1]->post([&job, r] { job(r); });
    job(0);
    for (int r = 1; r < W; ++r) g->workers[(size_t)r - 1]->wait();
    for (int r = 0; r < W; ++r)
        if (rcs[(size_t)r]) return W > 1 ? fail(rcs[(size_t)r], "shard %d: %s", r, errs[(size_t)r].c_str()) : rcs[(size_t)r];
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
        rc = upload_merge_meta(reinterpret_cast<uint32_t*>(h->d_meta), offs, offr, h->stream); if (rc) return rc;
        rc = merge_on_device(d_gathered, d_merged, reinterpret_cast<uint32_t*>(h->d_meta), offs, offr, 2, h->stream); if (rc) return rc;
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


static int group_create(const lcm_params* params, int n_devices, const int* device_ids, bool loopback, int loop_device, bool peer_copies, lcm_group** out) {
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
        g->copies = loopback || peer_copies;
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
        g->d_isums.assign(W, nullptr); g->d_isums_n.assign(W, 0);
        for (int r = 0; r < n_devices; ++r) {
            const int rc = lcm_create(&g->params, g->devices[(size_t)r], nullptr, &g->h[(size_t)r]);
            if (rc) return bail(rc);
        }
        if (!g->copies) {
            g->comms.assign(W, nullptr);
            ncclResult_t nr = ncclCommInitAll(g->comms.data(), n_devices, g->devices.data());
            if (nr != ncclSuccess) {
                // no communicator: the group still works, its two exchange steps fall back to peer copies
                g->comms.clear();
                g->copies = true;
                g->rccl_error = ncclGetErrorString(nr);
                (void)hipGetLastError();
            }
        }
        if (g->copies && !loopback && n_devices > 1) {
            // let every device of the group reach every other one's memory directly (xGMI); where peer access is not
            // available the runtime stages the copies through the host — slower, still correct
            for (int r = 0; r < n_devices; ++r) {
                if (hipSetDevice(g->devices[(size_t)r]) != hipSuccess) return bail(fail(LCM_ERR_HIP, "hipSetDevice(%d) failed", g->devices[(size_t)r]));
                for (int q = 0; q < n_devices; ++q) {
                    if (q == r) continue;
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, g->devices[(size_t)r], g->devices[(size_t)q]) == hipSuccess && can) {
                        const hipError_t pe = hipDeviceEnablePeerAccess(g->devices[(size_t)q], 0);
                        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                        else (void)hipGetLastError();
                    }
                }
            }
        }
        if (hipSetDevice(g->devices[0]) != hipSuccess || hipEventCreate(&g->ev0) != hipSuccess ||
            hipEventCreate(&g->ev1) != hipSuccess || hipEventCreate(&g->ev2) != hipSuccess ||
            hipEventCreate(&g->evg0) != hipSuccess || hipEventCreate(&g->evg1) != hipSuccess)
            return bail(fail(LCM_ERR_HIP, "hipEventCreate failed"));
        for (int r = 1; r < n_devices; ++r) g->workers.emplace_back(new Worker());
        *out = g;
        return LCM_OK;
    });
}

int lcm_group_create(const lcm_params* params, int n_devices, const int* device_ids, lcm_group** out) {
    return group_create(params, n_devices, device_ids, false, 0, false, out);
}

int lcm_group_create_peer(const lcm_params* params, int n_devices, const int* device_ids, lcm_group** out) {
    return group_create(params, n_devices, device_ids, false, 0, true, out);
}

int lcm_group_create_loopback(const lcm_params* params, int n_shards, int device_id, lcm_group** out) {
    return group_create(params, n_shards, nullptr, true, device_id, false, out);
}

const char* lcm_group_transport(const lcm_group* g) {
    if (!g) return "";
    if (g->loopback) return "loopback (device-local copies)";
    if (!g->copies) return "rccl";
    return g->rccl_error.empty() ? "peer copies (requested)" : "peer copies (ncclCommInitAll failed)";
}

void lcm_group_destroy(lcm_group* g) {
    if (!g) return;
    g->workers.clear();                             // joins the worker threads (none has a job: every call waits for its own)
    for (size_t r = 0; r < g->h.size(); ++r) {
        if (!g->h[r]) continue;
        (void)hipSetDevice(g->devices[r]);
        (void)hipDeviceSynchronize();
        if (r < g->d_qrows.size()) { (void)hipFree(g->d_qrows[r]); (void)hipFree(g->d_qcounts[r]); (void)hipFree(g->d_scores[r]); (void)hipFree(g->d_isums[r]); }
    }
    if (!g->devices.empty()) {
        (void)hipSetDevice(g->devices[0]);
        (void)hipFree(g->d_gather); (void)hipFree(g->d_merged); (void)hipFree(g->d_meta);
        (void)hipFree(g->d_gather_idx); (void)hipFree(g->d_merged_idx);
        for (hipEvent_t e : {g->ev0, g->ev1, g->ev2, g->evg0, g->evg1}) if (e) (void)hipEventDestroy(e);
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

int lcm_group_set_tuning(lcm_group* g, int knob, int value) {
    if (!g) return fail(LCM_ERR_INVALID_ARG, "NULL group");
    for (lcm_handle* h : g->h) { const int rc = lcm_set_tuning(h, knob, value); if (rc) return rc; }
    return LCM_OK;
}

int lcm_group_set_kernel_variant(lcm_group* g, int variant) {
    if (!g) return fail(LCM_ERR_INVALID_ARG, "NULL group");
    for (lcm_handle* h : g->h) { const int rc = lcm_set_kernel_variant(h, variant); if (rc) return rc; }
    return LCM_OK;
}

int lcm_group_sync(lcm_group* g) {
    if (!g) return fail(LCM_ERR_INVALID_ARG, "NULL group");
    for (lcm_handle* h : g->h) { const int rc = lcm_sync(h); if (rc) return rc; }
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
        ++g->db_stamp;
        return LCM_OK;
    });
}

int lcm_group_clear(lcm_group* g) {
    if (!g) return fail(LCM_ERR_INVALID_ARG, "NULL group");
    for (lcm_handle* h : g->h) { const int rc = lcm_db_clear(h); if (rc) return rc; }
    g->frames.clear();
    ++g->db_stamp; ++g->drop_stamp;
    return LCM_OK;
}

/* Drop the most recently appended frames: the group keeps its first n_frames frames (arrival order).  What the host
 * class's rollback needs when a pipelined processFrames fails half-way (host list, loop list and device database must
 * agree); tickets submitted before the call are void, as after lcm_group_clear. */
int lcm_group_truncate(lcm_group* g, int n_frames) {
    if (!g || n_frames < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    if ((size_t)n_frames >= g->frames.size()) return LCM_OK;
    const int W = g->world;
    for (int r = 0; r < W; ++r) {
        const int keep = n_frames > r ? (n_frames - r + W - 1) / W : 0;       // positions r, r + W, ... below n_frames
        const int rc = lcm_db_truncate(g->h[(size_t)r], keep); if (rc) return rc;
    }
    g->frames.resize((size_t)n_frames);
    ++g->db_stamp; ++g->drop_stamp;
    return LCM_OK;
}

/* Snapshot / resume of a sharded database, in lcm_db_save's OWN file format: frames are written in arrival order (each
 * read back from the shard that owns it), so a file saved by a group of 8 loads into a single handle, a group of any
 * other size, or the other way round.  Load validates the whole header against the file's size before the group's
 * database is touched (lcm::snapshot_open), replaces the contents, and leaves an EMPTY group if a later step fails. */
int lcm_group_save(lcm_group* g, const char* path) {
    if (!g || !path) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    return guarded([&]() -> int {
        int rc = lcm_group_sync(g); if (rc) return rc;
        FILE* f = fopen(path, "wb");
        if (!f) return fail(LCM_ERR_INVALID_ARG, "cannot open %s for writing", path);
        std::vector<lcm::FrameMeta> metas(g->frames.size());
        for (size_t i = 0; i < metas.size(); ++i) metas[i] = {g->frames[i].id, g->frames[i].n, g->frames[i].n_kp};
        bool ok = lcm::snapshot_write_header(f, metas);
        std::vector<uint8_t> buf;
        const size_t W = (size_t)g->world;
        for (size_t i = 0; ok && !rc && i < metas.size(); ++i) {
            if (metas[i].n == 0) continue;
            buf.resize((size_t)metas[i].n * LCM_DESC_BYTES);
            rc = lcm_db_read(g->h[i % W], (int)(i / W), buf.data(), metas[i].n);
            if (!rc) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        }
        ok = (fclose(f) == 0) && ok;
        if (rc) return rc;
        return ok ? LCM_OK : fail(LCM_ERR_HIP, "writing %s failed", path);
    });
}

int lcm_group_load(lcm_group* g, const char* path) {
    if (!g || !path) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    return guarded([&]() -> int {
        std::vector<lcm::FrameMeta> metas;
        uint32_t max_rows = 0;
        FILE* f = nullptr;
        int rc = lcm::snapshot_open(path, metas, &max_rows, &f); if (rc) return rc;
        struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
        rc = lcm_group_clear(g); if (rc) return rc;
        rc = lcm_group_reserve(g, (int)metas.size(), (int)std::max<uint32_t>(max_rows, 1));
        std::vector<uint8_t> buf((size_t)max_rows * LCM_DESC_BYTES + 1);
        for (size_t i = 0; !rc && i < metas.size(); ++i) {
            if (metas[i].n && fread(buf.data(), LCM_DESC_BYTES, (size_t)metas[i].n, f) != (size_t)metas[i].n) { rc = fail(LCM_ERR_INVALID_ARG, "%s: read error", path); break; }
            rc = lcm_group_append(g, metas[i].id, buf.data(), metas[i].n, metas[i].n_kp);
        }
        if (!rc) rc = lcm_group_sync(g);
        if (rc) {
            const std::string why = lcm::last_error();
            (void)lcm_group_clear(g);
            lcm::last_error() = why;
        }
        return rc;
    });
}

}  // extern "C"

namespace {

enum SearchMode { SEARCH_SCORES = 0, SEARCH_ARGMIN = 1, SEARCH_LOOPS = 2 };

// The bulk search of a group, in its three forms (score records; + per-pair index checksums through the argmin kernel;
// loop test on every shard's device, only candidates leave the devices).
int group_search(lcm_group* g, SearchMode mode, lcm_score* out_scores, uint32_t* out_isums, size_t cap, size_t* n_pairs,
                 size_t* pair_offsets, lcm_loop_candidate* out_cands, size_t cand_cap, size_t* n_cands) {
    const int W = g->world;
    const int N = (int)g->frames.size();
    // ---- bookkeeping: merged offsets, per-shard per-query offsets
    std::vector<uint32_t> offs;
    std::vector<std::vector<uint32_t>> offr;
    uint64_t total = 0;
    { const int rc0 = shard_layout(g->frames, W, g->params.min_gap, offs, offr, total); if (rc0) return rc0; }
    *n_pairs = (size_t)total;
    if (pair_offsets) for (int c = 0; c <= N; ++c) pair_offsets[c] = offs[(size_t)c];
    if (mode != SEARCH_LOOPS) {
        if (!out_scores) return LCM_OK;                     // sizing call
        if (cap < total) return fail(LCM_ERR_CAPACITY, "scores buffer holds %zu records, need %llu", cap, (unsigned long long)total);
    }
    g->info = lcm_group_info{};
    g->info.n_devices = W;
    g->info.loopback = g->loopback ? 1 : 0;
    g->info.pairs = total;
    if (!g->copies && !g->comms.empty()) { int cnt = 0; if (ncclCommCount(g->comms[0], &cnt) == ncclSuccess) g->info.rccl_ranks = cnt; }
    if (total == 0) return LCM_OK;

    // ---- 1. equal shard geometry on every device, then — only if the database changed since the last search — the
    //         all-gather of the shard arenas into every device's rank-major query buffer
    const int shard_cap = (N + W - 1) / W;
    int stride = 4;
    for (lcm_handle* h : g->h) stride = std::max(stride, h->stride_rows);
    for (const GFrame& f : g->frames) stride = std::max(stride, (f.n + 3) / 4 * 4);
    for (int r = 0; r < W; ++r) {
        int rc = lcm_db_reserve(g->h[(size_t)r], shard_cap, stride); if (rc) return rc;
        if (g->h[(size_t)r]->stride_rows != stride || g->h[(size_t)r]->cap_frames < shard_cap)
            return fail(LCM_ERR_HIP, "shard %d arena geometry mismatch", r);
    }
    const size_t shard_bytes = (size_t)shard_cap * (size_t)stride * LCM_DESC_BYTES;
    const bool gather = g->gathered_stamp != g->db_stamp || g->gathered_cap != shard_cap || g->gathered_stride != stride;
    g->info.arena_gather_skipped = gather ? 0 : 1;
    if (gather) {
        g->gathered_stamp = 0;
        for (int r = 0; r < W; ++r) {
            int rc = set_dev(g, r); if (rc) return rc;
            rc = ensure_dev(g->d_qrows[(size_t)r], g->d_qrows_bytes[(size_t)r], shard_bytes * (size_t)W, 512); if (rc) return rc;
            rc = ensure_dev(g->d_qcounts[(size_t)r], g->d_qcounts_n[(size_t)r], (size_t)shard_cap * (size_t)W); if (rc) return rc;
        }
        int rc = set_dev(g, 0); if (rc) return rc;
        if (g->copies) {
            // what ncclAllGather delivers, as device-to-device copies (every shard's appends have landed: host wait)
            for (int r = 0; r < W; ++r) { rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc; }
            HIP_TRY(hipEventRecord(g->evg0, g->h[0]->stream));
            for (int r = 0; r < W; ++r) {
                rc = set_dev(g, r); if (rc) return rc;           // the copies INTO device r are enqueued on its stream
                for (int s = 0; s < W; ++s) {
                    HIP_TRY(hipMemcpyAsync(g->d_qrows[(size_t)r] + (size_t)s * shard_bytes, g->h[(size_t)s]->d_rows, shard_bytes, hipMemcpyDeviceToDevice, g->h[(size_t)r]->stream));
                    HIP_TRY(hipMemcpyAsync(g->d_qcounts[(size_t)r] + (size_t)s * (size_t)shard_cap, g->h[(size_t)s]->d_counts, sizeof(int32_t) * (size_t)shard_cap, hipMemcpyDeviceToDevice, g->h[(size_t)r]->stream));
                }
            }
            rc = set_dev(g, 0); if (rc) return rc;
            HIP_TRY(hipEventRecord(g->evg1, g->h[0]->stream));
        } else {
            // every device's stream first waits (on the device, no host wait) for the appends of its own shard
            for (int r = 0; r < W; ++r) { rc = set_dev(g, r); if (rc) return rc; rc = lcm::wait_db(g->h[(size_t)r]); if (rc) return rc; }
            rc = set_dev(g, 0); if (rc) return rc;
            HIP_TRY(hipEventRecord(g->evg0, g->h[0]->stream));
            NCCL_TRY(ncclGroupStart());
            for (int r = 0; r < W; ++r) {
                lcm_handle* h = g->h[(size_t)r];
                NCCL_TRY(ncclAllGather(h->d_rows, g->d_qrows[(size_t)r], shard_bytes, ncclUint8, g->comms[(size_t)r], h->stream));
                NCCL_TRY(ncclAllGather(h->d_counts, g->d_qcounts[(size_t)r], (size_t)shard_cap, ncclInt32, g->comms[(size_t)r], h->stream));
            }
            NCCL_TRY(ncclGroupEnd());
            rc = set_dev(g, 0); if (rc) return rc;
            HIP_TRY(hipEventRecord(g->evg1, g->h[0]->stream));
        }
        g->info.gathered_query_bytes = (uint64_t)shard_bytes * (uint64_t)W;
        g->gathered_stamp = g->db_stamp; g->gathered_cap = shard_cap; g->gathered_stride = stride;
    }

    // ---- 2. per-shard search: every device plans + launches on its own stream from its own host thread
    std::vector<int32_t> ids((size_t)N), counts((size_t)N), kps((size_t)N);
    std::vector<uint32_t> q_frame_of((size_t)N);
    for (int p = 0; p < N; ++p) {
        ids[(size_t)p] = g->frames[(size_t)p].id;
        counts[(size_t)p] = g->frames[(size_t)p].n;
        kps[(size_t)p] = g->frames[(size_t)p].n_kp;
        q_frame_of[(size_t)p] = (uint32_t)((p % W) * shard_cap + p / W);
    }
    const bool argmin = mode == SEARCH_ARGMIN;
    const bool loops = mode == SEARCH_LOOPS;
    int rc = set_dev(g, 0); if (rc) return rc;
    if (!loops) {
        // device 0's shard is written straight into the gather buffer (it is the first segment)
        rc = ensure_dev(g->d_gather, g->d_gather_n, (size_t)total); if (rc) return rc;
        rc = ensure_dev(g->d_merged, g->d_merged_n, (size_t)total); if (rc) return rc;
        if (argmin) {
            rc = ensure_dev(g->d_gather_idx, g->d_gather_idx_n, (size_t)total); if (rc) return rc;
            rc = ensure_dev(g->d_merged_idx, g->d_merged_idx_n, (size_t)total); if (rc) return rc;
        }
    }
    for (int r = loops ? 0 : 1; r < W; ++r) {
        rc = set_dev(g, r); if (rc) return rc;
        rc = ensure_dev(g->d_scores[(size_t)r], g->d_scores_n[(size_t)r], (size_t)offr[(size_t)r][(size_t)N]); if (rc) return rc;
        if (argmin) { rc = ensure_dev(g->d_isums[(size_t)r], g->d_isums_n[(size_t)r], (size_t)offr[(size_t)r][(size_t)N]); if (rc) return rc; }
    }
    std::vector<std::vector<lcm_loop_candidate>> shard_cands(loops ? (size_t)W : 0);
    std::vector<size_t> shard_found((size_t)W, 0);
    std::vector<int> shard_over((size_t)W, 0);
    rc = run_all(g, [&](int r) -> int {
        lcm_handle* h = g->h[(size_t)r];
        const size_t want = (size_t)offr[(size_t)r][(size_t)N];
        void* dst = (r == 0 && !loops) ? (void*)g->d_gather : (void*)g->d_scores[(size_t)r];
        uint32_t* dsum = !argmin ? nullptr : (r == 0 ? g->d_gather_idx : g->d_isums[(size_t)r]);
        size_t n = 0;
        int rc2 = lcm::all_vs_all(h, g->d_qrows[(size_t)r], g->d_qcounts[(size_t)r], ids.data(), N, stride, dst, want, &n, nullptr, dsum,
                                  q_frame_of.data(), counts.data());
        if (rc2) return rc2;
        if (n != want) return fail(LCM_ERR_HIP, "scored %zu pairs, expected %zu", n, want);
        if (!loops || want == 0) return LCM_OK;
        // loop test on this shard's device over its own records: the frames it owns are positions r, r + W, ...
        std::vector<int32_t> oid, okp;
        for (int p = r; p < N; p += W) { oid.push_back(ids[(size_t)p]); okp.push_back(kps[(size_t)p]); }
        size_t found = 0;
        rc2 = lcm::loop_test_device(h, dst, want, offr[(size_t)r].data(), N, ids.data(), kps.data(), (int)oid.size(), oid.data(), okp.data(),
                                    out_cands ? cand_cap : 0, &found);
        shard_found[(size_t)r] = found;
        if (rc2 == LCM_ERR_CAPACITY) { shard_over[(size_t)r] = 1; return LCM_OK; }     // the caller sums the counts and reports
        if (rc2) return rc2;
        shard_cands[(size_t)r].resize(found);
        // each shard's candidates go to the host over that device's OWN PCIe link, all links at once
        if (found) {
            HIP_TRY(hipMemcpyAsync(shard_cands[(size_t)r].data(), h->d_cands, sizeof(lcm_loop_candidate) * found, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        return LCM_OK;
    });
    if (rc) return rc;

    auto read_timings = [&]() -> int {            // device clocks: per-shard scoring kernels, the arena all-gather
        for (int r = 0; r < W; ++r) {
            lcm_launch_info li{};
            int rc3 = lcm_last_launch_info(g->h[(size_t)r], &li); if (rc3) return rc3;
            g->info.kernel_ms[r] = li.kernel_ms;
            g->info.shard_pairs[r] = li.pairs;
            g->info.kernel_ms_max = std::max(g->info.kernel_ms_max, li.kernel_ms);
            g->info.distances += li.distances;
            g->info.algo_bytes += li.algo_bytes;
        }
        if (gather) {
            int rc3 = set_dev(g, 0); if (rc3) return rc3;
            float ms = 0.f;
            HIP_TRY(hipEventSynchronize(g->evg1));
            HIP_TRY(hipEventElapsedTime(&ms, g->evg0, g->evg1)); g->info.allgather_ms = ms;
        }
        return LCM_OK;
    };

    if (loops) {
        size_t total_found = 0;
        bool over = false;
        for (int r = 0; r < W; ++r) { total_found += shard_found[(size_t)r]; over = over || shard_over[(size_t)r]; }
        *n_cands = total_found;
        rc = read_timings(); if (rc) return rc;
        if (over || total_found > cand_cap || (!out_cands && total_found))
            return fail(LCM_ERR_CAPACITY, "%zu loop candidates but room for %zu", total_found, out_cands ? cand_cap : (size_t)0);
        // W sorted lists -> one, ordered by (current id, matched id): for a given current frame the shards' candidates
        // interleave by stored position, so this is a W-way merge on the host over candidates only
        std::vector<size_t> at((size_t)W, 0);
        for (size_t k = 0; k < total_found; ++k) {
            int best = -1;
            for (int r = 0; r < W; ++r) {
                if (at[(size_t)r] >= shard_cands[(size_t)r].size()) continue;
                const lcm_loop_candidate& c = shard_cands[(size_t)r][at[(size_t)r]];
                if (best < 0) { best = r; continue; }
                const lcm_loop_candidate& b = shard_cands[(size_t)best][at[(size_t)best]];
                if (c.current_frame_id < b.current_frame_id || (c.current_frame_id == b.current_frame_id && c.matched_frame_id < b.matched_frame_id)) best = r;
            }
            out_cands[k] = shard_cands[(size_t)best][at[(size_t)best]++];
        }
        return LCM_OK;
    }

    // ---- 3. gather the shards' records to device 0 (gatherv by grouped send / recv), merge there, one download
    rc = set_dev(g, 0); if (rc) return rc;
    HIP_TRY(hipEventRecord(g->ev0, g->h[0]->stream));
    std::vector<uint32_t> shard_base((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) shard_base[(size_t)r + 1] = shard_base[(size_t)r] + offr[(size_t)r][(size_t)N];
    if (W > 1 && g->copies) {
        // what the grouped ncclSend / ncclRecv delivers: each shard's records behind device 0's
        for (int r = 1; r < W; ++r) {
            const size_t n = (size_t)offr[(size_t)r][(size_t)N];
            if (!n) continue;
            rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(g->d_gather + shard_base[(size_t)r], g->d_scores[(size_t)r], n * sizeof(lcm_score), hipMemcpyDeviceToDevice, g->h[0]->stream));
            if (argmin) HIP_TRY(hipMemcpyAsync(g->d_gather_idx + shard_base[(size_t)r], g->d_isums[(size_t)r], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, g->h[0]->stream));
        }
    } else if (W > 1) {
        NCCL_TRY(ncclGroupStart());
        for (int r = 1; r < W; ++r) {
            const size_t n = (size_t)offr[(size_t)r][(size_t)N];
            if (!n) continue;
            NCCL_TRY(ncclSend(g->d_scores[(size_t)r], n * sizeof(lcm_score), ncclUint8, 0, g->comms[(size_t)r], g->h[(size_t)r]->stream));
            NCCL_TRY(ncclRecv(g->d_gather + shard_base[(size_t)r], n * sizeof(lcm_score), ncclUint8, r, g->comms[0], g->h[0]->stream));
            if (argmin) {
                NCCL_TRY(ncclSend(g->d_isums[(size_t)r], n, ncclUint32, 0, g->comms[(size_t)r], g->h[(size_t)r]->stream));
                NCCL_TRY(ncclRecv(g->d_gather_idx + shard_base[(size_t)r], n, ncclUint32, r, g->comms[0], g->h[0]->stream));
            }
        }
        NCCL_TRY(ncclGroupEnd());
        rc = set_dev(g, 0); if (rc) return rc;
    }
    rc = ensure_dev(g->d_meta, g->d_meta_n, (size_t)(W + 1) * ((size_t)N + 1)); if (rc) return rc;
    rc = upload_merge_meta(g->d_meta, offs, offr, g->h[0]->stream); if (rc) return rc;
    rc = merge_on_device(g->d_gather, g->d_merged, g->d_meta, offs, offr, 2, g->h[0]->stream); if (rc) return rc;
    if (argmin) { rc = merge_on_device(g->d_gather_idx, g->d_merged_idx, g->d_meta, offs, offr, 1, g->h[0]->stream); if (rc) return rc; }
    HIP_TRY(hipEventRecord(g->ev1, g->h[0]->stream));
    HIP_TRY(hipMemcpyAsync(out_scores, g->d_merged, sizeof(lcm_score) * (size_t)total, hipMemcpyDeviceToHost, g->h[0]->stream));
    if (argmin) HIP_TRY(hipMemcpyAsync(out_isums, g->d_merged_idx, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost, g->h[0]->stream));
    HIP_TRY(hipEventRecord(g->ev2, g->h[0]->stream));
    for (int r = 0; r < W; ++r) { rc = lcm_sync(g->h[(size_t)r]); if (rc) return rc; }

    // ---- timings (device clocks): per-shard kernels, gather + merge, download
    rc = read_timings(); if (rc) return rc;
    rc = set_dev(g, 0); if (rc) return rc;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1)); g->info.gather_merge_ms = ms;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev1, g->ev2)); g->info.download_ms = ms;
    g->info.gathered_score_bytes = (uint64_t)(total - offr[0][(size_t)N]) * (sizeof(lcm_score) + (argmin ? sizeof(uint32_t) : 0));
    return LCM_OK;
}

}  // namespace

extern "C" {

int lcm_group_all_vs_all(lcm_group* g, lcm_score* out_scores, size_t cap, size_t* n_pairs, size_t* pair_offsets) {
    if (!g || !n_pairs) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    return guarded([&] { return group_search(g, SEARCH_SCORES, out_scores, nullptr, cap, n_pairs, pair_offsets, nullptr, 0, nullptr); });
}

int lcm_group_all_vs_all_argmin(lcm_group* g, lcm_score* out_scores, uint32_t* out_index_sums, size_t cap, size_t* n_pairs, size_t* pair_offsets) {
    if (!g || !n_pairs) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    if (out_scores && !out_index_sums) return fail(LCM_ERR_INVALID_ARG, "out_index_sums is NULL");
    return guarded([&] { return group_search(g, SEARCH_ARGMIN, out_scores, out_index_sums, cap, n_pairs, pair_offsets, nullptr, 0, nullptr); });
}

int lcm_group_all_vs_all_loops(lcm_group* g, lcm_loop_candidate* out, size_t cap, size_t* n_out, size_t* n_pairs_out) {
    if (!g || !n_out) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    size_t n_pairs = 0;
    const int rc = guarded([&] { return group_search(g, SEARCH_LOOPS, nullptr, nullptr, 0, &n_pairs, nullptr, out, cap, n_out); });
    if (n_pairs_out) *n_pairs_out = n_pairs;
    return rc;
}

int lcm_group_last_info(const lcm_group* g, lcm_group_info* info) {
    if (!g || !info) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *info = g->info;
    return LCM_OK;
}

/* Online totals over the shards (lcm_online_stats_read per shard): the work is summed, kernel_ms is the slowest
 * shard's — the devices run side by side. */
int lcm_group_online_stats_read(lcm_group* g, lcm_online_stats* out, int reset) {
    if (!g || !out) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *out = lcm_online_stats{};
    for (lcm_handle* h : g->h) {
        lcm_online_stats s{};
        const int rc = lcm_online_stats_read(h, &s, reset); if (rc) return rc;
        out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);
        out->launches += s.launches; out->queries = std::max(out->queries, s.queries);
        out->pairs += s.pairs; out->distances += s.distances; out->algo_bytes += s.algo_bytes;
    }
    return LCM_OK;
}

/* Online query: the frame goes to every device (pinned staging + H2D per device, each from that device's own host
 * thread, all enqueued before any is awaited), each scores it against its shard, and the per-shard records — a few KB —
 * are interleaved on the host: for a message this small a collective would only add latency. */
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
        const int rc_s = run_all(g, [&](int r) { return lcm_query_submit(g->h[(size_t)r], query, nq, query_frame_id, &tickets[(size_t)r]); });
        const std::string why = rc_s ? lcm::last_error() : std::string();
        // collect every ticket that was issued, also after a failed submit (nothing may stay in flight)
        const int rc_c = run_all(g, [&](int r) -> int {
            if (tickets[(size_t)r] < 0) return LCM_OK;
            const int want = e > r ? (e - r + W - 1) / W : 0;
            std::vector<lcm_score> part((size_t)std::max(want, 1));
            int n = 0;
            const int rc = lcm_query_collect(g->h[(size_t)r], tickets[(size_t)r], part.data(), nullptr, (int)part.size(), &n);
            if (rc || rc_s) return rc;
            if (n != want) return fail(LCM_ERR_HIP, "returned %d records, expected %d", n, want);
            for (int k = 0; k < n; ++k) out_scores[(size_t)r + (size_t)k * (size_t)W] = part[(size_t)k];
            return LCM_OK;
        });
        if (rc_s) { lcm::last_error() = why; return rc_s; }
        if (rc_c) return rc_c;
        if (out_frame_ids) for (int s = 0; s < e; ++s) out_frame_ids[s] = g->frames[(size_t)s].id;
        *n_out = e;
        return LCM_OK;
    });
}

/* Asynchronous micro-batched online queries over all shards (BASELINE.json configs[4] inside one process): the batch goes
 * to every device with ONE lcm_query_submit_batch each, issued from the devices' own host threads; the call returns as
 * soon as every device has its work ENQUEUED (the callers' buffers are free then), with one group ticket.  Up to 4 may
 * be in flight; lcm_group_query_collect_batch waits for one, and each device's thread writes its records straight into
 * their interleaved places of the single-device order (query 0's records first, offsets[n_queries + 1]). */
int lcm_group_query_submit_batch(lcm_group* g, const uint8_t* const* queries, const int* nq, const int* query_frame_ids, int n_queries, int* ticket) {
    if (!g || !ticket || !queries || !nq || !query_frame_ids) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *ticket = -1;
    if (n_queries < 1 || n_queries > lcm::MAX_QUERY_BATCH) return fail(LCM_ERR_INVALID_ARG, "a batch holds 1..%d queries", lcm::MAX_QUERY_BATCH);
    return guarded([&]() -> int {
        const int W = g->world;
        int t = -1;
        for (int i = 0; i < lcm::QUERY_SLOTS; ++i) if (!g->tickets[i].busy) { t = i; break; }
        if (t < 0) return fail(LCM_ERR_CAPACITY, "%d group queries already in flight: collect one first", lcm::QUERY_SLOTS);
        GTicket& T = g->tickets[t];
        T.n_queries = n_queries; T.total = 0; T.stamp = g->drop_stamp;
        for (int b = 0; b < n_queries; ++b) { T.e[b] = eligible_count(g->frames, query_frame_ids[b], g->params.min_gap); T.total += (size_t)T.e[b]; }
        T.shard_ticket.assign((size_t)W, -1);
        T.part.resize((size_t)W);
        const int rc = run_all(g, [&](int r) { return lcm_query_submit_batch(g->h[(size_t)r], queries, nq, query_frame_ids, n_queries, &T.shard_ticket[(size_t)r]); });
        if (rc) {
            // some shards may hold a ticket: drain them, keep the first error
            const std::string why = lcm::last_error();
            (void)run_all(g, [&](int r) -> int {
                if (T.shard_ticket[(size_t)r] < 0) return LCM_OK;
                std::vector<lcm_score> sink(std::max<size_t>(T.total, 1));
                size_t n = 0;
                (void)lcm_query_collect_batch(g->h[(size_t)r], T.shard_ticket[(size_t)r], sink.data(), sink.size(), &n, nullptr);
                return LCM_OK;
            });
            lcm::last_error() = why;
            return rc;
        }
        T.busy = true;
        *ticket = t;
        return LCM_OK;
    });
}

int lcm_group_query_collect_batch(lcm_group* g, int ticket, lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets) {
    if (!g || !n_out || ticket < 0 || ticket >= lcm::QUERY_SLOTS || !g->tickets[ticket].busy) return fail(LCM_ERR_INVALID_ARG, "bad group ticket");
    *n_out = 0;
    return guarded([&]() -> int {
        const int W = g->world;
        GTicket& T = g->tickets[ticket];
        const int B = T.n_queries;
        std::vector<size_t> offs((size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) offs[(size_t)b + 1] = offs[(size_t)b] + (size_t)T.e[b];
        if (T.stamp == g->drop_stamp) {
            // recoverable argument errors keep the ticket (as lcm_query_collect_batch does)
            if (T.total > cap) return fail(LCM_ERR_CAPACITY, "%zu score records but room for %zu (the ticket stays valid)", T.total, cap);
            if (T.total > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL (the ticket stays valid)");
        }
        const bool is_void = T.stamp != g->drop_stamp;      // frames were dropped after the submit: the shards refuse their tickets too
        const int rc = run_all(g, [&](int r) -> int {
            std::vector<lcm_score>& part = T.part[(size_t)r];
            size_t want_total = 0;
            for (int b = 0; b < B; ++b) want_total += T.e[b] > r ? (size_t)((T.e[b] - r + W - 1) / W) : 0;
            if (part.size() < std::max<size_t>(want_total, 1)) part.resize(std::max<size_t>(want_total, 1));
            std::vector<size_t> poffs((size_t)B + 1, 0);
            size_t n = 0;
            const int rc2 = lcm_query_collect_batch(g->h[(size_t)r], T.shard_ticket[(size_t)r], part.data(), part.size(), &n, poffs.data());
            if (is_void) return LCM_OK;                     // drained; the group reports the void ticket itself
            if (rc2) return rc2;
            for (int b = 0; b < B; ++b) {
                const int want = T.e[b] > r ? (T.e[b] - r + W - 1) / W : 0;
                if ((int)(poffs[(size_t)b + 1] - poffs[(size_t)b]) != want) return fail(LCM_ERR_HIP, "returned a wrong record count for query %d", b);
                lcm_score* dst = out_scores + offs[(size_t)b] + (size_t)r;
                const lcm_score* src = part.data() + poffs[(size_t)b];
                for (int k = 0; k < want; ++k) dst[(size_t)k * (size_t)W] = src[k];       // disjoint places per shard: no two threads share a record
            }
            return LCM_OK;
        });
        T.busy = false;
        if (is_void) return fail(LCM_ERR_NOT_FOUND, "group ticket %d was submitted before lcm_group_clear / lcm_group_truncate: its result is void", ticket);
        if (rc) return rc;
        if (offsets) memcpy(offsets, offs.data(), sizeof(size_t) * offs.size());
        *n_out = T.total;
        return LCM_OK;
    });
}

/* The synchronous form: submit + collect. */
int lcm_group_query_scores_batch(lcm_group* g, const uint8_t* const* queries, const int* nq, const int* query_frame_ids, int n_queries,
                                 lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets) {
    if (!g || !n_out || !queries || !nq || !query_frame_ids || n_queries < 1) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    return guarded([&]() -> int {
        size_t total = 0;
        for (int b = 0; b < n_queries && b < lcm::MAX_QUERY_BATCH; ++b) total += (size_t)eligible_count(g->frames, query_frame_ids[b], g->params.min_gap);
        if (offsets) {
            size_t o = 0;
            for (int b = 0; b < n_queries && b < lcm::MAX_QUERY_BATCH; ++b) { offsets[b] = o; o += (size_t)eligible_count(g->frames, query_frame_ids[b], g->params.min_gap); }
            if (n_queries <= lcm::MAX_QUERY_BATCH) offsets[n_queries] = o;
        }
        if (total > cap) return fail(LCM_ERR_CAPACITY, "%zu score records but room for %zu", total, cap);
        if (total > 0 && !out_scores) return fail(LCM_ERR_INVALID_ARG, "out_scores is NULL");
        int t = -1;
        int rc = lcm_group_query_submit_batch(g, queries, nq, query_frame_ids, n_queries, &t); if (rc) return rc;
        return lcm_group_query_collect_batch(g, t, out_scores, cap, n_out, offsets);
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
