// lcm_api.cpp — host side of the C ABI declared in include/lcm.h (compiled with hipcc, gfx950 only).
//
// Owns: handle lifetime and parameters, the device-resident stored-frame descriptor database (the `frames_` vector of
// loop_closing::LoopClosingSystem, include/loop_closing.hpp:69, reduced to what the Hamming path reads: id, row count,
// keypoint count, 32-byte rows) with pinned staging for streaming appends and snapshot / restore, the host-side
// IEEE-double loop test (README.md:123-126), launch information and the device scratch helpers.  Pair mode, online
// queries, the bulk search, cross-check and the matrix-core host code live in lcm_pair / lcm_online / lcm_bulk /
// lcm_cross / lcm_mfma_host .cpp (see lcm_internal.h).
//
// There is no CPU compute path anywhere in the library: every distance is computed by the kernels in lcm_kernels.hip
// (or, opt-in, lcm_mfma.hip).
#include "lcm_internal.h"

namespace lcm {
std::string& last_error() { static thread_local std::string e; return e; }
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
void set_last_error(const char* msg) { last_error() = msg ? msg : ""; }
}  // namespace lcm

namespace lcm {

int set_device(const lcm_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    return LCM_OK;
}

int wait_db(lcm_handle* h) {   // make the match stream see every append issued so far
    if (h->pending_copy) {
        HIP_TRY(hipStreamWaitEvent(h->stream, h->db_ready, 0));
        h->pending_copy = false;
    }
    return LCM_OK;
}

int sync_online_streams(lcm_handle* h) {
    for (QuerySlot& q : h->qslots)
        if (q.stream) HIP_TRY(hipStreamSynchronize(q.stream));
    return LCM_OK;
}

// Eligible stored slots for a query id are a prefix because ids are strictly increasing by slot.
int eligible_prefix(const lcm_handle* h, int query_id, int gap) {
    // largest e with frames[e-1].id <= query_id - gap; a frame is never compared with itself (gap >= 1)
    long long lim = (long long)query_id - std::max(gap, 1);
    int lo = 0, hi = (int)h->frames.size();
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if ((long long)h->frames[mid].id <= lim) lo = mid + 1; else hi = mid;
    }
    return lo;
}

int pick_chunk(const lcm_handle* h, size_t total_pairs) {
    if (h->tune_item_slots >= 1) return h->tune_item_slots;          // lcm_set_tuning(LCM_TUNE_ITEM_SLOTS)
    // Small items keep the tail of the launch short (an item is the unit the dispatcher balances): measured on cfg2,
    // 2 frames per item 656.4 ms, 4: 657.8, 8: 660.4, 16: 667.6.  Very large runs use 4 to bound the item list.
    return total_pairs >= (1u << 21) ? 4 : (total_pairs >= 4096 ? 2 : 1);
}

int launch_and_time(lcm_handle* h, const ScoreArgs& a, uint32_t n_items, int max_q_rows, bool write_keys) {
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    hipError_t e = lcm::launch_score(a, n_items, max_q_rows, write_keys, h->variant, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = 1;
    h->info.workgroups = n_items;
    h->info.route = LCM_ROUTE_PLAIN;
    return LCM_OK;
}

}  // namespace lcm

namespace {

int grow_arena(lcm_handle* h, int need_frames, int need_rows) {
    int new_stride = std::max(h->stride_rows, round_up(std::max(need_rows, 1), ROW_PAD));
    int new_cap = h->cap_frames;
    if (need_frames > new_cap) new_cap = std::max(need_frames, std::max(64, h->cap_frames * 2));
    if (new_stride == h->stride_rows && new_cap == h->cap_frames && h->d_rows) return LCM_OK;
    if (new_stride > 65535) return fail(LCM_ERR_CAPACITY, "a stored frame may hold at most 65535 rows (got %d)", need_rows);

    uint8_t* nrows = nullptr;
    int32_t* ncounts = nullptr;
    size_t bytes = (size_t)new_cap * new_stride * LCM_DESC_BYTES + ARENA_SLACK;
    // the two new buffers belong to this function until they are installed: every failure path below gives them back
    struct Pending { uint8_t*& r; int32_t*& c; bool keep = false; ~Pending() { if (!keep) { (void)hipFree(r); (void)hipFree(c); } } } pending{nrows, ncounts};
    HIP_TRY(hipMalloc((void**)&nrows, bytes));
    HIP_TRY(hipMalloc((void**)&ncounts, sizeof(int32_t) * (size_t)new_cap));
    HIP_TRY(hipMemsetAsync(nrows, 0, bytes, h->stream));
    HIP_TRY(hipMemsetAsync(ncounts, 0, sizeof(int32_t) * (size_t)new_cap, h->stream));
    int n = (int)h->frames.size();
    if (n > 0 && h->d_rows) {
        // all appends must have landed before the old arena is re-pitched
        HIP_TRY(hipStreamSynchronize(h->copy_stream));
        HIP_TRY(hipMemcpy2DAsync(nrows, (size_t)new_stride * LCM_DESC_BYTES, h->d_rows,
                                 (size_t)h->stride_rows * LCM_DESC_BYTES, (size_t)h->stride_rows * LCM_DESC_BYTES,
                                 (size_t)n, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(ncounts, h->d_counts, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    { int rc = sync_online_streams(h); if (rc) return rc; }      // queries in flight still read the old arena
    pending.keep = true;                                          // from here on the handle owns them
    (void)hipFree(h->d_rows); (void)hipFree(h->d_counts);
    h->d_rows = nrows; h->d_counts = ncounts;
    h->cap_frames = new_cap; h->stride_rows = new_stride;
    h->plan.key = 0;

    if (new_cap > h->h_counts_cap) {
        int32_t* nc = nullptr;
        HIP_TRY(hipHostMalloc((void**)&nc, sizeof(int32_t) * (size_t)new_cap, hipHostMallocDefault));
        if (h->h_counts) { memcpy(nc, h->h_counts, sizeof(int32_t) * (size_t)h->h_counts_cap); HIP_TRY(hipHostFree(h->h_counts)); }
        h->h_counts = nc; h->h_counts_cap = new_cap;
    }
    size_t stage_need = (size_t)new_stride * LCM_DESC_BYTES;
    if (stage_need > h->h_stage_bytes) {
        for (int i = 0; i < STAGE_BUFS; ++i) {
            if (h->h_stage[i]) { HIP_TRY(hipEventSynchronize(h->stage_done[i])); HIP_TRY(hipHostFree(h->h_stage[i])); }
            HIP_TRY(hipHostMalloc((void**)&h->h_stage[i], stage_need, hipHostMallocDefault));
        }
        h->h_stage_bytes = stage_need;
    }
    return LCM_OK;
}

int check_append(lcm_handle* h, int frame_id, int n) {
    if (n < 0) return fail(LCM_ERR_INVALID_ARG, "negative row count %d", n);
    if (!h->frames.empty() && frame_id <= h->frames.back().id)
        return fail(LCM_ERR_ORDER, "frame id %d appended after id %d: ids must be strictly increasing", frame_id,
                    h->frames.back().id);
    int need_frames = (int)h->frames.size() + 1;
    if (need_frames > h->cap_frames || n > h->stride_rows || !h->d_rows) {
        int want_rows = std::max(n, h->stride_rows > 0 ? h->stride_rows : DEFAULT_MAX_DESC);
        int rc = grow_arena(h, need_frames, want_rows);
        if (rc) return rc;
    }
    return LCM_OK;
}

}  // namespace

extern "C" {

void lcm_params_default(lcm_params* p) {
    if (!p) return;
    p->ratio = 2;
    p->dist_floor = 0;
    p->min_matches = 50;
    p->min_gap = 30;
    p->sim_threshold = 0.15;
    p->cross_check = 0;
    p->reserved = 0;
}

const char* lcm_last_error(void) { return lcm::last_error().c_str(); }
const char* lcm_backend_name(void) { return "hip-gfx950"; }

int lcm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int lcm_create(const lcm_params* params, int device_id, void* stream, lcm_handle** out) {
    if (!out) return fail(LCM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(LCM_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device_id < 0 || device_id >= ndev) return fail(LCM_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", device_id, ndev);
    lcm_handle* h = new (std::nothrow) lcm_handle();
    if (!h) return fail(LCM_ERR_OOM, "host allocation failed");
    lcm_params_default(&h->params);
    h->device = device_id;
    if (params && lcm_set_params(h, params) != LCM_OK) { delete h; return LCM_ERR_INVALID_ARG; }
    auto bail = [&](int rc) { lcm_destroy(h); return rc; };
    if (hipSetDevice(device_id) != hipSuccess) return bail(fail(LCM_ERR_HIP, "hipSetDevice(%d) failed", device_id));
    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(LCM_ERR_HIP, "hipStreamCreate failed"));
        h->own_stream = true;
    }
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(LCM_ERR_HIP, "hipStreamCreate(copy) failed"));
    if (hipEventCreateWithFlags(&h->db_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev_start) != hipSuccess || hipEventCreate(&h->ev_stop) != hipSuccess ||
        hipEventCreate(&h->ev_aux_start) != hipSuccess || hipEventCreate(&h->ev_aux_stop) != hipSuccess)
        return bail(fail(LCM_ERR_HIP, "hipEventCreate failed"));
    for (int i = 0; i < STAGE_BUFS; ++i) {
        if (hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming) != hipSuccess)
            return bail(fail(LCM_ERR_HIP, "hipEventCreate failed"));
    }
    *out = h;
    return LCM_OK;
}

void lcm_destroy(lcm_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    for (QuerySlot& q : h->qslots) if (q.stream) (void)hipStreamSynchronize(q.stream);
    (void)hipFree(h->d_rows); (void)hipFree(h->d_counts);
    (void)hipFree(h->d_keys); (void)hipFree(h->plan.d_items); (void)hipFree(h->plan.d_pk_tab);
    (void)hipFree(h->d_bulk_scores); (void)hipFree(h->d_meta); (void)hipFree(h->d_cands);
    for (QuerySlot& q : h->qslots) {
        (void)hipFree(q.d_query); (void)hipFree(q.d_scores); (void)hipFree(q.d_dist); (void)hipFree(q.d_meta);
        if (q.h_meta) (void)hipHostFree(q.h_meta);
        if (q.h_query) (void)hipHostFree(q.h_query);
        if (q.h_scores) (void)hipHostFree(q.h_scores);
        if (q.done) (void)hipEventDestroy(q.done);
        if (q.k0) (void)hipEventDestroy(q.k0);
        if (q.k1) (void)hipEventDestroy(q.k1);
        if (q.fence) (void)hipEventDestroy(q.fence);
        if (q.stream) (void)hipStreamDestroy(q.stream);
    }
    for (int i = 0; i < STAGE_BUFS; ++i) {
        if (h->h_stage[i]) (void)hipHostFree(h->h_stage[i]);
        if (h->stage_done[i]) (void)hipEventDestroy(h->stage_done[i]);
    }
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    if (h->h_pair_stage) (void)hipHostFree(h->h_pair_stage);
    if (h->h_final_keys) (void)hipHostFree(h->h_final_keys);
    (void)hipFree(h->d_pair_stage); (void)hipFree(h->d_xq);
    (void)hipFree(h->d_pm1); (void)hipFree(h->d_qpm1); (void)hipFree(h->d_mdist); (void)hipFree(h->d_mitems); (void)hipFree(h->d_mmeta);
    if (h->db_ready) (void)hipEventDestroy(h->db_ready);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    if (h->ev_aux_start) (void)hipEventDestroy(h->ev_aux_start);
    if (h->ev_aux_stop) (void)hipEventDestroy(h->ev_aux_stop);
    for (hipEvent_t ev : h->fold_ev) (void)hipEventDestroy(ev);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int lcm_set_params(lcm_handle* h, const lcm_params* p) {
    if (!h || !p) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    if (p->ratio < 0 || p->dist_floor < 0 || p->min_gap < 0 || p->min_matches < 0) return fail(LCM_ERR_INVALID_ARG, "negative parameter");
    if (p->ratio > 65536 || p->dist_floor > 65536) return fail(LCM_ERR_INVALID_ARG, "ratio / dist_floor above 65536 (distances are <= 256)");
    if (!(p->sim_threshold == p->sim_threshold)) return fail(LCM_ERR_INVALID_ARG, "sim_threshold is NaN");
    if (p->cross_check < 0 || p->cross_check > 2) return fail(LCM_ERR_INVALID_ARG, "cross_check must be 0, 1 or 2");
    h->params = *p;
    h->plan.key = 0;
    return LCM_OK;
}

int lcm_get_params(const lcm_handle* h, lcm_params* p) {
    if (!h || !p) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *p = h->params;
    return LCM_OK;
}

int lcm_sync(lcm_handle* h) {
    if (!h) return fail(LCM_ERR_INVALID_ARG, "NULL handle");
    int rc = set_device(h); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->copy_stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(hipStreamSynchronize(h->stream2));
    return sync_online_streams(h);
}

int lcm_set_kernel_variant(lcm_handle* h, int variant) {
    if (!h) return fail(LCM_ERR_INVALID_ARG, "NULL handle");
    if (variant < 0 || variant > 5) return fail(LCM_ERR_INVALID_ARG, "unknown kernel variant %d", variant);
    h->variant = variant;
    return LCM_OK;
}

int lcm_set_tuning(lcm_handle* h, int knob, int value) {
    if (!h) return fail(LCM_ERR_INVALID_ARG, "NULL handle");
    switch (knob) {
        case LCM_TUNE_ITEM_SLOTS:
            if (value < 0 || value > 64) return fail(LCM_ERR_INVALID_ARG, "item slots must be 0 (automatic) .. 64");
            h->tune_item_slots = value; h->plan.key = 0; return LCM_OK;
        case LCM_TUNE_ONLINE_SPLIT:
            if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 16 && value != 32)
                return fail(LCM_ERR_INVALID_ARG, "online split must be -1 (automatic), 0 (off), 1, 2, 4 (rows per lane of a 256-thread workgroup), 16 or 32 (64 / 128 threads x 8 rows)");
            h->tune_online_split = value; return LCM_OK;
        case LCM_TUNE_ONLINE_STREAMS:
            if (value != 0 && value != 1) return fail(LCM_ERR_INVALID_ARG, "online streams must be 0 (the handle's stream) or 1 (one per query slot)");
            h->tune_online_streams = value; return LCM_OK;
        case LCM_TUNE_PAIR_UPLOAD_KERNEL:
            if (value != 0 && value != 1) return fail(LCM_ERR_INVALID_ARG, "pair upload must be 0 (hipMemcpyAsync) or 1 (kernel)");
            h->tune_pair_upload_kernel = value; return LCM_OK;
        case LCM_TUNE_PAIR_HOST_FOLD:
            if (value != 0 && value != 1) return fail(LCM_ERR_INVALID_ARG, "pair host fold must be 0 (device buffer + copy) or 1 (pinned host memory)");
            h->tune_pair_host_fold = value; return LCM_OK;
        case LCM_TUNE_PACKED_SCRATCH_MB:
            if (value < 1 || value > 65536) return fail(LCM_ERR_INVALID_ARG, "packed scratch must be 1 .. 65536 MiB per chunk");
            h->pk_scratch_cfg_words = h->pk_scratch_words = (size_t)value << 18; h->plan.key = 0; return LCM_OK;
        case LCM_TUNE_PACKED:
            if (value < -1 || value > 2) return fail(LCM_ERR_INVALID_ARG, "packed rows must be -1 (automatic), 0 (off), 1 (on) or 2 (on, 1536-row columns)");
            h->tune_packed = value; h->plan.key = 0; return LCM_OK;
        default: return fail(LCM_ERR_INVALID_ARG, "unknown tuning knob %d", knob);
    }
}

/* ---- database ---------------------------------------------------------------------------------------- */

static int db_reserve_impl(lcm_handle* h, int n_frames, int max_desc) {
    if (!h || n_frames < 0 || max_desc < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    int rc = set_device(h); if (rc) return rc;
    return grow_arena(h, std::max(n_frames, 1), std::max(max_desc, 1));
}

static int db_append_impl(lcm_handle* h, int frame_id, const uint8_t* desc, int n, int n_keypoints) {
    if (!h || (n > 0 && !desc)) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    rc = check_append(h, frame_id, n); if (rc) return rc;
    const int slot = (int)h->frames.size();
    const int b = h->stage_next;
    h->stage_next = (b + 1) % STAGE_BUFS;
    // the staging buffer is free once the copy that last used it has completed
    HIP_TRY(hipEventSynchronize(h->stage_done[b]));
    const int np = padded_rows(n);
    if (n > 0) {
        memcpy(h->h_stage[b], desc, (size_t)n * LCM_DESC_BYTES);
        for (int r = n; r < np; ++r) memcpy(h->h_stage[b] + (size_t)r * LCM_DESC_BYTES, desc + (size_t)(n - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
        HIP_TRY(hipMemcpyAsync(h->d_rows + (size_t)slot * h->stride_rows * LCM_DESC_BYTES, h->h_stage[b],
                               (size_t)np * LCM_DESC_BYTES, hipMemcpyHostToDevice, h->copy_stream));
    }
    h->h_counts[slot] = n;
    HIP_TRY(hipMemcpyAsync(h->d_counts + slot, h->h_counts + slot, sizeof(int32_t), hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(h->stage_done[b], h->copy_stream));
    HIP_TRY(hipEventRecord(h->db_ready, h->copy_stream));
    h->pending_copy = true; h->db_ready_recorded = true;
    h->frames.push_back({frame_id, n, n_keypoints < 0 ? n : n_keypoints});
    h->plan.key = 0;
    return LCM_OK;
}

static int db_append_device_impl(lcm_handle* h, int frame_id, const void* d_desc, int n, int n_keypoints) {
    if (!h || (n > 0 && !d_desc)) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    rc = check_append(h, frame_id, n); if (rc) return rc;
    const int slot = (int)h->frames.size();
    uint8_t* dst = h->d_rows + (size_t)slot * h->stride_rows * LCM_DESC_BYTES;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(dst, d_desc, (size_t)n * LCM_DESC_BYTES, hipMemcpyDeviceToDevice, h->stream));
        for (int r = n; r < padded_rows(n); ++r)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)r * LCM_DESC_BYTES, (const uint8_t*)d_desc + (size_t)(n - 1) * LCM_DESC_BYTES,
                                   LCM_DESC_BYTES, hipMemcpyDeviceToDevice, h->stream));
    }
    h->h_counts[slot] = n;
    HIP_TRY(hipMemcpyAsync(h->d_counts + slot, h->h_counts + slot, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    h->frames.push_back({frame_id, n, n_keypoints < 0 ? n : n_keypoints});
    h->plan.key = 0;
    return LCM_OK;
}

int lcm_db_size(const lcm_handle* h) { return h ? (int)h->frames.size() : 0; }

int lcm_db_clear(lcm_handle* h) {
    if (!h) return fail(LCM_ERR_INVALID_ARG, "NULL handle");
    int rc = lcm_sync(h); if (rc) return rc;
    h->frames.clear();
    h->plan.key = 0;
    // Tickets of queries submitted against the dropped frames stay collectable, but only to learn that: their records
    // refer to slots that no longer exist (lcm_query_collect returns LCM_ERR_NOT_FOUND and frees the ticket).
    ++h->db_generation;
    return LCM_OK;
}

int lcm_db_truncate(lcm_handle* h, int n_frames) {
    if (!h || n_frames < 0) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    if ((size_t)n_frames >= h->frames.size()) return LCM_OK;
    int rc = lcm_sync(h); if (rc) return rc;        // nothing in flight may still read the slots that go away
    h->frames.resize((size_t)n_frames);
    h->plan.key = 0;
    ++h->db_generation;                             // tickets submitted against the longer database are void (as after a clear)
    return LCM_OK;
}

int lcm_db_frame_info(const lcm_handle* h, int slot, int* frame_id, int* n_desc, int* n_keypoints) {
    if (!h || slot < 0 || slot >= (int)h->frames.size()) return fail(LCM_ERR_INVALID_ARG, "slot %d out of range", slot);
    if (frame_id) *frame_id = h->frames[slot].id;
    if (n_desc) *n_desc = h->frames[slot].n;
    if (n_keypoints) *n_keypoints = h->frames[slot].n_kp;
    return LCM_OK;
}

int lcm_db_read(lcm_handle* h, int slot, uint8_t* desc_out, int cap_rows) {
    if (!h || slot < 0 || slot >= (int)h->frames.size()) return fail(LCM_ERR_INVALID_ARG, "slot out of range");
    int n = h->frames[slot].n;
    if (cap_rows < n) return fail(LCM_ERR_CAPACITY, "need room for %d rows", n);
    int rc = lcm_sync(h); if (rc) return rc;
    if (n > 0) HIP_TRY(hipMemcpy(desc_out, h->d_rows + (size_t)slot * h->stride_rows * LCM_DESC_BYTES, (size_t)n * LCM_DESC_BYTES, hipMemcpyDeviceToHost));
    return LCM_OK;
}

/* ---- snapshot / restore (N4): raw little-endian file, no parsing of anything executable --------------------- */
namespace {
struct SnapHeader { char magic[8]; uint32_t version, n_frames, max_rows, reserved; };
}

static int db_save_impl(lcm_handle* h, const char* path) {
    if (!h || !path) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = lcm_sync(h); if (rc) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(LCM_ERR_INVALID_ARG, "cannot open %s for writing", path);
    bool ok = lcm::snapshot_write_header(f, h->frames);
    std::vector<uint8_t> buf;
    for (size_t s = 0; ok && s < h->frames.size(); ++s) {
        const size_t bytes = (size_t)h->frames[s].n * LCM_DESC_BYTES;
        if (!bytes) continue;
        buf.resize(bytes);
        if (hipMemcpy(buf.data(), h->d_rows + s * (size_t)h->stride_rows * LCM_DESC_BYTES, bytes, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = fwrite(buf.data(), 1, bytes, f) == bytes;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? LCM_OK : fail(LCM_ERR_HIP, "writing %s failed", path);
}

extern "C++" {
namespace lcm {
// Opens and VALIDATES a database snapshot: everything the header claims is checked against the file's real size BEFORE
// anything is allocated or any database is touched — a corrupt or hostile header can neither trigger a huge allocation
// nor cost the caller the frames it already has.  On LCM_OK *f_out is positioned at the first frame's rows (caller closes).
int snapshot_open(const char* path, std::vector<FrameMeta>& metas, uint32_t* real_max_rows, FILE** f_out) {
    *f_out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LCM_ERR_NOT_FOUND, "cannot open %s", path);
    struct Closer { FILE* f; bool keep = false; ~Closer() { if (!keep) fclose(f); } } closer{f};
    struct stat st{};
    if (fstat(fileno(f), &st) != 0 || st.st_size < 0) return fail(LCM_ERR_INVALID_ARG, "cannot stat %s", path);
    const uint64_t file_bytes = (uint64_t)st.st_size;
    SnapHeader hd{};
    if (fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, "LCMDB01", 8) != 0 || hd.version != 1 || hd.max_rows > 65535u)
        return fail(LCM_ERR_INVALID_ARG, "%s is not an lcm database snapshot", path);
    if (hd.n_frames > 0x7FFFFFFFu || sizeof hd + (uint64_t)hd.n_frames * sizeof(FrameMeta) > file_bytes)
        return fail(LCM_ERR_INVALID_ARG, "%s: header claims %u frames but the file holds %llu bytes", path, hd.n_frames,
                    (unsigned long long)file_bytes);
    metas.assign(hd.n_frames, FrameMeta{});
    if (hd.n_frames && fread(metas.data(), sizeof(FrameMeta), hd.n_frames, f) != hd.n_frames)
        return fail(LCM_ERR_INVALID_ARG, "%s is truncated", path);
    uint64_t need = sizeof hd + (uint64_t)hd.n_frames * sizeof(FrameMeta);
    *real_max_rows = 0;                  // the arena is sized from the frames themselves, not from the header's claim
    for (size_t s = 0; s < metas.size(); ++s) {
        if (metas[s].n < 0 || (uint32_t)metas[s].n > hd.max_rows) return fail(LCM_ERR_INVALID_ARG, "%s: bad row count in frame %zu", path, s);
        if (s > 0 && metas[s].id <= metas[s - 1].id) return fail(LCM_ERR_INVALID_ARG, "%s: frame ids are not increasing", path);
        need += (uint64_t)metas[s].n * LCM_DESC_BYTES;
        *real_max_rows = std::max(*real_max_rows, (uint32_t)metas[s].n);
    }
    if (need > file_bytes) return fail(LCM_ERR_INVALID_ARG, "%s is truncated (%llu bytes, needs %llu)", path,
                                       (unsigned long long)file_bytes, (unsigned long long)need);
    closer.keep = true;
    *f_out = f;
    return LCM_OK;
}

// Writes the header + frame table of a snapshot (the rows follow, frame after frame); false on an I/O error.
bool snapshot_write_header(FILE* f, const std::vector<FrameMeta>& metas) {
    SnapHeader hd{{'L', 'C', 'M', 'D', 'B', '0', '1', 0}, 1u, (uint32_t)metas.size(), 0u, 0u};
    for (const FrameMeta& m : metas) hd.max_rows = std::max<uint32_t>(hd.max_rows, (uint32_t)m.n);
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1;
    if (!metas.empty()) ok = ok && fwrite(metas.data(), sizeof(FrameMeta), metas.size(), f) == metas.size();
    return ok;
}
}  // namespace lcm
}  // extern "C++"

static int db_load_impl(lcm_handle* h, const char* path) {
    if (!h || !path) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    std::vector<FrameMeta> metas;
    uint32_t real_max_rows = 0;
    FILE* f = nullptr;
    int rc = lcm::snapshot_open(path, metas, &real_max_rows, &f); if (rc) return rc;
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    rc = lcm_db_clear(h); if (rc) return rc;
    rc = lcm_db_reserve(h, (int)metas.size(), (int)std::max<uint32_t>(real_max_rows, 1));
    std::vector<uint8_t> buf((size_t)real_max_rows * LCM_DESC_BYTES + 1);
    for (size_t s = 0; !rc && s < metas.size(); ++s) {
        if (metas[s].n && fread(buf.data(), LCM_DESC_BYTES, (size_t)metas[s].n, f) != (size_t)metas[s].n) { rc = fail(LCM_ERR_INVALID_ARG, "%s: read error", path); break; }
        rc = lcm_db_append(h, metas[s].id, buf.data(), metas[s].n, metas[s].n_kp);
    }
    if (!rc) rc = lcm_sync(h);
    if (rc) {
        // an I/O or device error after the old contents were dropped: leave an EMPTY database, not half of one
        const std::string why = lcm::last_error();
        (void)lcm_db_clear(h);
        lcm::last_error() = why;
    }
    return rc;
}

/* ---- loop search ------------------------------------------------------------------------------------- */

int lcm_loop_test(const lcm_params* p, const lcm_score* s, int n_query_kp, int n_train_kp, double* similarity) {
    if (similarity) *similarity = 0.0;
    if (!p || !s) return 0;
    const int den = std::min(n_query_kp, n_train_kp);
    if (den <= 0) return 0;
    const double sim = (double)s->good_count / (double)den;     // README.md:126
    if (similarity) *similarity = sim;
    return (sim > p->sim_threshold) && ((long long)s->good_count >= (long long)p->min_matches);   // README.md:123-124
}

int lcm_last_launch_info(const lcm_handle* hc, lcm_launch_info* info) {
    lcm_handle* h = const_cast<lcm_handle*>(hc);
    if (!h || !info) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    if (h->info_pending || h->aux_pending) { const int rc = set_device(h); if (rc) return rc; }     // its events live on its device
    if (h->info_pending) {
        HIP_TRY(hipEventSynchronize(h->ev_stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_stop));
        h->info.kernel_ms = ms;
        h->info.aux_kernel_ms = 0.0;
        h->info_pending = false;
        h->info.score_launches = 0; h->info.score_ms_sum = 0.0;
        if (h->info.route == LCM_ROUTE_PACKED) {          // per chunk: its score kernel, its fold kernel (ev_stop has passed: all are done)
            double fold = 0.0, score = 0.0;
            for (size_t k = 0; k + 3 <= h->fold_ev_used && k + 3 <= h->fold_ev.size(); k += 3) {
                float f = 0.f;
                if (hipEventElapsedTime(&f, h->fold_ev[k], h->fold_ev[k + 1]) == hipSuccess) score += f; else (void)hipGetLastError();
                if (hipEventElapsedTime(&f, h->fold_ev[k + 1], h->fold_ev[k + 2]) == hipSuccess) fold += f; else (void)hipGetLastError();
                ++h->info.score_launches;
            }
            h->info.aux_kernel_ms = fold;
            h->info.score_ms_sum = score;
        } else {
            h->info.launches_in_flight = 1;
        }
    }
    if (h->aux_pending) {
        HIP_TRY(hipEventSynchronize(h->ev_aux_stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_aux_start, h->ev_aux_stop));
        h->info.aux_kernel_ms = ms;
        h->aux_pending = false;
    }
    *info = h->info;
    return LCM_OK;
}

int lcm_online_stats_read(lcm_handle* h, lcm_online_stats* out, int reset) {
    if (!h || !out) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *out = h->online;
    if (reset) h->online = lcm_online_stats{};
    return LCM_OK;
}

int lcm_last_bulk_scores(const lcm_handle* h, const void** d_scores, size_t* n_records) {
    if (!h || !d_scores || !n_records) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    *d_scores = h->bulk_scores_valid ? h->d_bulk_scores : nullptr;
    *n_records = h->bulk_scores_valid;
    return LCM_OK;
}

/* ---- device scratch helpers -------------------------------------------------------------------------- */

int lcm_dev_alloc(lcm_handle* h, size_t bytes, void** d_ptr) {
    if (!h || !d_ptr) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    HIP_TRY(hipMalloc(d_ptr, std::max<size_t>(bytes, 16)));
    return LCM_OK;
}
int lcm_dev_free(lcm_handle* h, void* d_ptr) {
    if (!h) return fail(LCM_ERR_INVALID_ARG, "NULL handle");
    int rc = set_device(h); if (rc) return rc;
    HIP_TRY(hipFree(d_ptr));
    return LCM_OK;
}
int lcm_dev_upload(lcm_handle* h, void* d_dst, const void* src, size_t bytes) {
    if (!h || (bytes && (!d_dst || !src))) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return LCM_OK;
}
int lcm_dev_download(lcm_handle* h, void* dst, const void* d_src, size_t bytes) {
    if (!h || (bytes && (!dst || !d_src))) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return LCM_OK;
}

/* ---- exported entry points of the functions above, behind the exception guard ---- */

int lcm_db_reserve(lcm_handle* h, int n_frames, int max_desc) {
    return guarded([&] { return db_reserve_impl(h, n_frames, max_desc); });
}
int lcm_db_append(lcm_handle* h, int frame_id, const uint8_t* desc, int n, int n_keypoints) {
    return guarded([&] { return db_append_impl(h, frame_id, desc, n, n_keypoints); });
}
int lcm_db_append_device(lcm_handle* h, int frame_id, const void* d_desc, int n, int n_keypoints) {
    return guarded([&] { return db_append_device_impl(h, frame_id, d_desc, n, n_keypoints); });
}
int lcm_db_save(lcm_handle* h, const char* path) {
    return guarded([&] { return db_save_impl(h, path); });
}
int lcm_db_load(lcm_handle* h, const char* path) {
    return guarded([&] { return db_load_impl(h, path); });
}

}  // extern "C"
