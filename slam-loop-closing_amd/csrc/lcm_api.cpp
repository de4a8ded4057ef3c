// lcm_api.cpp — host side of the C ABI declared in include/lcm.h (compiled with hipcc, gfx950 only).
//
// Owns: the device-resident stored-frame descriptor database (the `frames_` vector of
// loop_closing::LoopClosingSystem, include/loop_closing.hpp:69, reduced to what the Hamming path reads:
// id, row count, keypoint count, 32-byte rows), pinned staging for streaming appends, work-list planning for the
// pair-scoring kernel, and the host-side IEEE-double loop test (README.md:123-126).
//
// There is no CPU compute path in this file: every distance is computed by the kernels in lcm_kernels.hip.
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

namespace {

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

int grow_arena(lcm_handle* h, int need_frames, int need_rows) {
    int new_stride = std::max(h->stride_rows, round_up(std::max(need_rows, 1), ROW_PAD));
    int new_cap = h->cap_frames;
    if (need_frames > new_cap) new_cap = std::max(need_frames, std::max(64, h->cap_frames * 2));
    if (new_stride == h->stride_rows && new_cap == h->cap_frames && h->d_rows) return LCM_OK;
    if (new_stride > 65535) return fail(LCM_ERR_CAPACITY, "a stored frame may hold at most 65535 rows (got %d)", need_rows);

    uint8_t* nrows = nullptr;
    int32_t* ncounts = nullptr;
    size_t bytes = (size_t)new_cap * new_stride * LCM_DESC_BYTES + ARENA_SLACK;
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
    if (h->d_rows) HIP_TRY(hipFree(h->d_rows));
    if (h->d_counts) HIP_TRY(hipFree(h->d_counts));
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

// rows [n, round_up(n,4)) of a stored frame are copies of row n-1 (see lcm_kernels.hip)
inline int padded_rows(int n) { return round_up(n, ROW_PAD); }

uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h;
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

int launch_and_time(lcm_handle* h, const lcm::ScoreArgs& a, uint32_t n_items, int max_q_rows, bool write_keys) {
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    hipError_t e = lcm::launch_score(a, n_items, max_q_rows, write_keys, h->variant, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    h->info_pending = true;
    h->info.launches = 1;
    h->info.workgroups = n_items;
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
    (void)hipFree(h->d_rows); (void)hipFree(h->d_counts);
    (void)hipFree(h->d_keys); (void)hipFree(h->plan.d_items);
    (void)hipFree(h->d_bulk_scores); (void)hipFree(h->d_meta); (void)hipFree(h->d_cands);
    for (QuerySlot& q : h->qslots) {
        (void)hipFree(q.d_query); (void)hipFree(q.d_scores); (void)hipFree(q.d_dist); (void)hipFree(q.d_meta);
        if (q.h_meta) (void)hipHostFree(q.h_meta);
        if (q.h_query) (void)hipHostFree(q.h_query);
        if (q.h_scores) (void)hipHostFree(q.h_scores);
        if (q.done) (void)hipEventDestroy(q.done);
        if (q.k0) (void)hipEventDestroy(q.k0);
        if (q.k1) (void)hipEventDestroy(q.k1);
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
    return LCM_OK;
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
            if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4)
                return fail(LCM_ERR_INVALID_ARG, "online split must be -1 (automatic), 0 (off), 1, 2 or 4 rows per lane");
            h->tune_online_split = value; return LCM_OK;
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
    h->pending_copy = true;
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
    SnapHeader hd{{'L', 'C', 'M', 'D', 'B', '0', '1', 0}, 1u, (uint32_t)h->frames.size(), 0u, 0u};
    for (const FrameMeta& m : h->frames) hd.max_rows = std::max<uint32_t>(hd.max_rows, (uint32_t)m.n);
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1;
    if (!h->frames.empty()) ok = ok && fwrite(h->frames.data(), sizeof(FrameMeta), h->frames.size(), f) == h->frames.size();
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

static int db_load_impl(lcm_handle* h, const char* path) {
    if (!h || !path) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LCM_ERR_NOT_FOUND, "cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    // Everything the header claims is checked against the file's real size BEFORE anything is allocated or the
    // current database is touched: a corrupt or hostile header can neither trigger a huge allocation nor cost the
    // caller the frames it already has.
    struct stat st{};
    if (fstat(fileno(f), &st) != 0 || st.st_size < 0) return fail(LCM_ERR_INVALID_ARG, "cannot stat %s", path);
    const uint64_t file_bytes = (uint64_t)st.st_size;
    SnapHeader hd{};
    if (fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, "LCMDB01", 8) != 0 || hd.version != 1 || hd.max_rows > 65535u)
        return fail(LCM_ERR_INVALID_ARG, "%s is not an lcm database snapshot", path);
    if (hd.n_frames > 0x7FFFFFFFu || sizeof hd + (uint64_t)hd.n_frames * sizeof(FrameMeta) > file_bytes)
        return fail(LCM_ERR_INVALID_ARG, "%s: header claims %u frames but the file holds %llu bytes", path, hd.n_frames,
                    (unsigned long long)file_bytes);
    std::vector<FrameMeta> metas(hd.n_frames);
    if (hd.n_frames && fread(metas.data(), sizeof(FrameMeta), hd.n_frames, f) != hd.n_frames)
        return fail(LCM_ERR_INVALID_ARG, "%s is truncated", path);
    uint64_t need = sizeof hd + (uint64_t)hd.n_frames * sizeof(FrameMeta);
    for (size_t s = 0; s < metas.size(); ++s) {
        if (metas[s].n < 0 || (uint32_t)metas[s].n > hd.max_rows) return fail(LCM_ERR_INVALID_ARG, "%s: bad row count in frame %zu", path, s);
        if (s > 0 && metas[s].id <= metas[s - 1].id) return fail(LCM_ERR_INVALID_ARG, "%s: frame ids are not increasing", path);
        need += (uint64_t)metas[s].n * LCM_DESC_BYTES;
    }
    if (need > file_bytes) return fail(LCM_ERR_INVALID_ARG, "%s is truncated (%llu bytes, needs %llu)", path,
                                       (unsigned long long)file_bytes, (unsigned long long)need);
    int rc = lcm_db_clear(h); if (rc) return rc;
    rc = lcm_db_reserve(h, (int)hd.n_frames, (int)std::max<uint32_t>(hd.max_rows, 1));
    std::vector<uint8_t> buf((size_t)hd.max_rows * LCM_DESC_BYTES + 1);
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
    const int CH = lcm::MAX_FUSED_QUERY_ROWS;
    const size_t P = jobs.size();
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
    HIP_TRY(hipMemcpyAsync(h->d_pair_stage + up_from, h->h_pair_stage + up_from, up_bytes - up_from, hipMemcpyHostToDevice, h->stream));

    lcm::ScoreArgs a{};
    a.q_rows = (const uint32_t*)(q_in_stage ? h->d_pair_stage : d_q_base);
    a.db_rows = (const uint32_t*)(t_in_stage ? h->d_pair_stage : d_t_base);
    a.pair_items = reinterpret_cast<const lcm::PairItem*>(h->d_pair_stage + off_items);
    a.scores = nullptr; a.keys = h->d_keys; a.keys_stride = CH;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    rc = launch_and_time(h, a, (uint32_t)n_items, max_nq > CH ? CH : max_nq, true); if (rc) return rc;
    lcm::FoldArgs f{};
    f.seg_keys = h->d_keys;
    f.pairs = reinterpret_cast<const lcm::PairDesc*>(h->d_pair_stage + off_descs);
    f.final_keys = h->d_keys + n_items * (size_t)CH;
    f.n_pairs = (uint32_t)P;
    hipError_t e = lcm::launch_fold_pair_keys(f, (uint32_t)max_nq, h->stream);
    if (e != hipSuccess) return fail(LCM_ERR_HIP, "fold kernel launch failed: %s", hipGetErrorString(e));
    h->info.launches = 2;
    h->info.pairs = P; h->info.distances = 0; h->info.algo_bytes = 0;
    for (const PairJob& jb : jobs) {
        h->info.distances += (uint64_t)jb.nq * (uint64_t)jb.nt;
        h->info.algo_bytes += (uint64_t)jb.nt * 32 + (uint64_t)jb.nq * 32 + 8;
    }
    HIP_TRY(hipMemcpyAsync(h->h_final_keys, f.final_keys, sizeof(uint32_t) * total_rows, hipMemcpyDeviceToHost, h->stream));
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
static int pair_keys(lcm_handle* h, RowSrc q, RowSrc t, std::vector<uint32_t>& keys_out) {
    int rc = set_device(h); if (rc) return rc;
    if (h->params.cross_check) {
        if (q.n > LCM_MAX_TRAIN_ROWS) return fail(LCM_ERR_CAPACITY, "cross_check: at most %d query rows", LCM_MAX_TRAIN_ROWS);
        lcm_params saved = h->params;
        h->params.cross_check = 0;
        std::vector<uint32_t> fk, bk;
        rc = pair_keys(h, q, t, fk);
        if (!rc) rc = pair_keys(h, t, q, bk);
        h->params = saved;
        if (rc) return rc;
        cross_combine(saved.cross_check, fk.data(), q.n, bk.data(), t.n, keys_out);
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
    int rc = pair_keys(h, RowSrc{query, nullptr, nq}, RowSrc{train, nullptr, nt}, keys); if (rc) return rc;
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
    int rc = pair_keys(h, RowSrc{query, nullptr, nq}, RowSrc{train, nullptr, nt}, keys); if (rc) return rc;
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
    rc = pair_keys(h, q, t, keys); if (rc) return rc;
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

// scores of (query frame at q_rows[q_frame]) against stored slots [0, n_elig)
static void account_prefix(lcm_handle* h, int nq, int n_elig) {
    uint64_t dist = 0, bytes = (uint64_t)nq * 32;
    for (int s = 0; s < n_elig; ++s) { dist += (uint64_t)nq * h->frames[s].n; bytes += (uint64_t)h->frames[s].n * 32 + 8; }
    h->info.pairs = (uint64_t)n_elig; h->info.distances = dist; h->info.algo_bytes = bytes;
}

// cross_check scoring of "query c against stored slots [0, elig[c])" for n_q queries, records written to d_scores in
// (query, slot) order starting at index 0.  Query c is nq[c] rows starting at row q_row0[c] of the matrix at d_qbase,
// padded for the train role.  Per chunk of pairs (bounded key scratch): forward keys (query rows -> stored frame),
// backward keys (stored rows -> query frame: roles swapped), k_cross_score folds both into the score record on the device.
static int cross_score_prefixes(lcm_handle* h, const uint8_t* d_qbase, const uint32_t* q_row0, const int* nq, const int* elig,
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
    h->info.launches = launches; h->info.workgroups = 0;
    h->info.pairs = pairs; h->info.distances = dist; h->info.algo_bytes = bytes;
    return LCM_OK;
}

// OPT-IN variant 4: the bulk search on the matrix cores (lcm_mfma.hip).  Same records as variants 0 / 1, bit for bit.
// qbase / q_pitch_rows / q_frame_of describe the query set's packed rows (the arena itself in self mode); nqv[c] and
// offsets come from the plan.  Work goes out in chunks of <= 524,288 pairs (4 GiB of per-row distances).
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
static int mfma_online(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int pitch_rows, int B, const int* nq, const int* elig) {
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
    h->info.launches = 3; h->info.workgroups = (uint32_t)items.size();
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

static int mfma_bulk(lcm_handle* h, bool self, const uint8_t* q_rows, const int32_t* d_q_counts, uint32_t q_pitch_rows,
                     const uint32_t* q_frame_of, const int* nqv, int n_q, const std::vector<size_t>& offsets, lcm_score* d_scores) {
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
            hipError_t e = fp4 ? lcm::launch_score_mfma_fp4(a, (uint32_t)items.size(), h->stream)
                               : lcm::launch_score_mfma(a, (uint32_t)items.size(), h->stream);
            if (e != hipSuccess) return fail(LCM_ERR_HIP, "MFMA kernel launch failed: %s", hipGetErrorString(e));
            lcm::FinalizeBulkArgs f{};
            f.dist = h->d_mdist; f.offsets = h->d_mmeta; f.nq = reinterpret_cast<const int32_t*>(h->d_mmeta + n_q + 1);
            f.db_counts = h->d_counts; f.scores = d_scores; f.n_q = (uint32_t)n_q; f.pair_base = (uint32_t)offsets[(size_t)c0];
            f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
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
    h->info.launches = launches; h->info.workgroups = biggest;
    h->info.pairs = offsets[(size_t)n_q]; h->info.distances = dist; h->info.algo_bytes = bytes;
    return LCM_OK;
}

// Enqueue (no host synchronisation) the scoring of ONE query frame — `nq` rows at device address d_q — against stored
// slots [0, n_elig), and the download of the n_elig score records into the slot's pinned buffer.  Work items are
// implicit (derived from blockIdx), so nothing but the query itself crosses PCIe.  Short databases use the split
// mode (lcm_kernels.hip): 2 / 4 / 8 workgroups per pair + the on-device fold.
static int enqueue_query(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int nq, int n_elig) {
    q.n_elig = n_elig; q.nq = nq; q.n_batch = 0;
    q.acc_pairs = q.acc_distances = q.acc_bytes = 0; q.acc_launches = 0; q.acc_queries = 1;
    if (n_elig <= 0) { HIP_TRY(hipEventRecord(q.done, h->stream)); return LCM_OK; }
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
    const int split_env = h->tune_online_split;              // lcm_set_tuning(LCM_TUNE_ONLINE_SPLIT); -1 = automatic
    int qpt = 0;
    if (h->variant == 0 && nq > 512) {
        if (split_env >= 0) qpt = split_env;                 // 0 = never split, 1/2/4 = force that many rows per lane
        // measured (bench.py --mode stream, LCM_SPLIT sweep): finer pieces balance 256 CUs better whenever a launch
        // holds only a few thousand pairs — 1000 frames: 1.71e12 unsplit -> 2.18e12; 2500 frames: 2.32e12 -> 2.57e12
        else if (n_elig < 256) qpt = 1;
        else if (n_elig < 3072) qpt = 2;
        else if (n_elig < 6144) qpt = 4;
    }
    rc = ensure_dev(q.d_scores, q.d_scores_n, (size_t)n_elig); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, (size_t)n_elig); if (rc) return rc;
    lcm::ScoreArgs a{};
    a.q_rows = d_q; a.q_counts = nullptr; a.items = nullptr;
    a.db_rows = (const uint32_t*)h->d_rows; a.db_counts = h->d_counts; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    a.imp_nq = nq; a.imp_total = (uint32_t)n_elig;
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    HIP_TRY(hipEventRecord(q.k0, h->stream));
    if (qpt == 1 || qpt == 2 || qpt == 4) {
        const int chunk_rows = 256 * qpt;
        const int n_chunks = (nq + chunk_rows - 1) / chunk_rows;
        const size_t n_items = (size_t)n_elig * n_chunks;
        rc = ensure_dev(q.d_dist, q.d_dist_n, n_items * chunk_rows); if (rc) return rc;
        a.q_stride_words = (uint32_t)chunk_rows * LCM_DESC_WORDS;
        a.imp_chunks = (uint32_t)n_chunks; a.imp_chunk_rows = (uint32_t)chunk_rows; a.imp_spi = 1;
        a.scores = nullptr /* split mode writes no per-chunk records */; a.keys = q.d_dist; a.keys_stride = (uint32_t)chunk_rows;
        hipError_t e = lcm::launch_score_split(a, (uint32_t)n_items, qpt, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeArgs f{};
        f.dist = q.d_dist; f.padded_rows = (uint32_t)(n_chunks * chunk_rows); f.nq = nq;
        f.db_counts = h->d_counts; f.slot_begin = 0; f.scores = q.d_scores;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        e = lcm::launch_finalize(f, (uint32_t)n_elig, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
        h->info.launches = 2; h->info.workgroups = (uint32_t)n_items;
    } else {
        const int spi = n_elig >= 8192 ? 4 : (n_elig >= 4096 ? 2 : 1);
        const uint32_t n_items = (uint32_t)((n_elig + spi - 1) / spi);
        a.q_stride_words = 0;
        a.imp_chunks = 1; a.imp_chunk_rows = (uint32_t)std::max(nq, 1); a.imp_spi = (uint32_t)spi;
        a.scores = q.d_scores; a.keys = nullptr; a.keys_stride = 0;
        hipError_t e = lcm::launch_score(a, n_items, nq, false, h->variant, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        h->info.launches = 1; h->info.workgroups = n_items;
    }
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    HIP_TRY(hipEventRecord(q.k1, h->stream));
    h->info_pending = true;
    account_prefix(h, nq, n_elig);
    q.acc_pairs = h->info.pairs; q.acc_distances = h->info.distances; q.acc_bytes = h->info.algo_bytes;
    q.acc_launches = h->info.launches; q.acc_queries = 1;
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * (size_t)n_elig, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(q.done, h->stream));
    return LCM_OK;
}

// Micro-batch of online queries: B query frames (already in device memory at d_q, query b at row b * rows_per_query)
// against stored slots [0, elig[b]) each, ONE score launch (+ one finalize launch in split mode), one download.
// A launch of B x n_elig pairs fills the chip where a single query's few hundred pairs leave its tail idle, and the
// host pays one submit / collect round trip per B frames.
static int enqueue_batch(lcm_handle* h, QuerySlot& q, const uint32_t* d_q, int rows_per_query, int B, const int* nq, const int* elig) {
    size_t total = 0;
    int max_nq = 0;
    for (int b = 0; b < B; ++b) { total += (size_t)elig[b]; max_nq = std::max(max_nq, nq[b]); q.bat_elig[b] = elig[b]; }
    q.n_batch = B; q.n_elig = (int)total; q.nq = max_nq;
    q.acc_pairs = q.acc_distances = q.acc_bytes = 0; q.acc_launches = 0; q.acc_queries = (uint32_t)B;
    if (total == 0) { HIP_TRY(hipEventRecord(q.done, h->stream)); return LCM_OK; }
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
    int qpt = 0;
    if (max_nq > 512) {
        if (h->tune_online_split >= 0) qpt = h->tune_online_split;
        else if (total < 1536) qpt = 1;          // same rule as a single query, on the batch's total pair count
        else if (total < 6144) qpt = 2;
        else if (total < 12288) qpt = 4;
    }
    rc = ensure_dev(q.d_scores, q.d_scores_n, total); if (rc) return rc;
    rc = ensure_pinned(q.h_scores, q.h_scores_n, total); if (rc) return rc;
    lcm::ScoreArgs a{};
    a.q_rows = d_q; a.q_counts = nullptr; a.items = nullptr;
    a.db_rows = (const uint32_t*)h->d_rows; a.db_counts = h->d_counts; a.db_stride_words = (uint32_t)h->stride_rows * LCM_DESC_WORDS;
    a.ratio = h->params.ratio; a.dist_floor = h->params.dist_floor;
    a.imp_nbatch = (uint32_t)B;
    const bool split = (qpt == 1 || qpt == 2 || qpt == 4);
    const int chunk_rows = split ? 256 * qpt : rows_per_query;
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
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    HIP_TRY(hipEventRecord(q.k0, h->stream));
    if (split) {
        rc = ensure_dev(q.d_dist, q.d_dist_n, total * (size_t)n_chunks * chunk_rows); if (rc) return rc;
        a.scores = nullptr; a.keys = q.d_dist; a.keys_stride = (uint32_t)chunk_rows;
        hipError_t e = lcm::launch_score_split(a, wg, qpt, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        lcm::FinalizeArgs f{};
        f.dist = q.d_dist; f.padded_rows = (uint32_t)(n_chunks * chunk_rows); f.nq = 0;
        f.db_counts = h->d_counts; f.slot_begin = 0; f.scores = q.d_scores;
        f.ratio = h->params.ratio; f.dist_floor = h->params.dist_floor;
        f.n_batch = (uint32_t)B;
        for (int b = 0; b <= B; ++b) f.bat_pair[b] = a.bat_pair[b];
        for (int b = 0; b < B; ++b) f.bat_nq[b] = nq[b];
        e = lcm::launch_finalize(f, pair, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "finalize launch failed: %s", hipGetErrorString(e));
        h->info.launches = 2;
    } else {
        a.scores = q.d_scores; a.keys = nullptr; a.keys_stride = 0;
        hipError_t e = lcm::launch_score(a, wg, max_nq, false, h->variant >= 2 ? 0 : h->variant, h->stream);
        if (e != hipSuccess) return fail(LCM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        h->info.launches = 1;
    }
    h->info.workgroups = wg;
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    HIP_TRY(hipEventRecord(q.k1, h->stream));
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
    HIP_TRY(hipMemcpyAsync(q.h_scores, q.d_scores, sizeof(lcm_score) * total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(q.done, h->stream));
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
            *ticket = i;
            return LCM_OK;
        }
    return fail(LCM_ERR_CAPACITY, "%d queries already in flight: collect one first", QUERY_SLOTS);
}

static int query_submit_impl(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, int* ticket) {
    if (!h || nq < 0 || !ticket || (nq > 0 && !query)) return fail(LCM_ERR_INVALID_ARG, "bad argument");
    *ticket = -1;
    if (nq > lcm::MAX_FUSED_QUERY_ROWS) return fail(LCM_ERR_CAPACITY, "a query frame may hold at most %d rows", lcm::MAX_FUSED_QUERY_ROWS);
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
    if (nq > 0 && n_elig > 0) {
        memcpy(q.h_query, query, (size_t)nq * LCM_DESC_BYTES);       // the caller's buffer is free when we return
        for (int r = nq; r < rows_up; ++r) memcpy(q.h_query + (size_t)r * LCM_DESC_BYTES, query + (size_t)(nq - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
        HIP_TRY(hipMemcpyAsync(q.d_query, q.h_query, (size_t)rows_up * LCM_DESC_BYTES, hipMemcpyHostToDevice, h->stream));
    }
    rc = enqueue_query(h, q, (const uint32_t*)q.d_query, nq, n_elig); if (rc) return rc;
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
        if (nq[b] > lcm::MAX_FUSED_QUERY_ROWS) return fail(LCM_ERR_CAPACITY, "a query frame may hold at most %d rows", lcm::MAX_FUSED_QUERY_ROWS);
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
        int qpt = 0;
        if (max_nq > 512) {
            if (h->tune_online_split >= 0) qpt = h->tune_online_split;
            else if (total < 1536) qpt = 1;
            else if (total < 6144) qpt = 2;
            else if (total < 12288) qpt = 4;
        }
        if (qpt == 1 || qpt == 2 || qpt == 4) pitch = round_up(max_nq, 256 * qpt);
    }
    if (h->params.cross_check) pitch = padded_rows(std::max(max_nq, 1)) + 2 * ROW_PAD;   // room for every query's padding rows
    const size_t bytes = (size_t)pitch * (size_t)n_queries * LCM_DESC_BYTES;
    rc = ensure_pinned(q.h_query, q.h_query_bytes, bytes); if (rc) return rc;
    rc = ensure_dev(q.d_query, q.d_query_bytes, bytes, ARENA_SLACK); if (rc) return rc;
    if (total > 0) {
        for (int b = 0; b < n_queries; ++b)          // the callers' buffers are free when we return
            if (nq[b] > 0) {
                uint8_t* dst = q.h_query + (size_t)b * pitch * LCM_DESC_BYTES;
                memcpy(dst, queries[b], (size_t)nq[b] * LCM_DESC_BYTES);
                if (h->params.cross_check)
                    for (int r = nq[b]; r < padded_rows(nq[b]) + ROW_PAD; ++r) memcpy(dst + (size_t)r * LCM_DESC_BYTES, queries[b] + (size_t)(nq[b] - 1) * LCM_DESC_BYTES, LCM_DESC_BYTES);
            }
        HIP_TRY(hipMemcpyAsync(q.d_query, q.h_query, bytes, hipMemcpyHostToDevice, h->stream));
    }
    rc = enqueue_batch(h, q, (const uint32_t*)q.d_query, pitch, n_queries, nq, elig); if (rc) return rc;
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
        if (nq > lcm::MAX_FUSED_QUERY_ROWS) return fail(LCM_ERR_CAPACITY, "a query frame may hold at most %d rows", lcm::MAX_FUSED_QUERY_ROWS);
        rc = acquire_query_slot(h, &t); if (rc) return rc;
        QuerySlot& q = h->qslots[t];
        q.query_id = current_frame_id;
        rc = enqueue_query(h, q, (const uint32_t*)(h->d_rows + (size_t)slot * h->stride_rows * LCM_DESC_BYTES), nq,
                           eligible_prefix(h, current_frame_id, h->params.min_gap));
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
    if (q_stride_rows > lcm::MAX_FUSED_QUERY_ROWS && !self) return fail(LCM_ERR_CAPACITY, "query frames may hold at most %d rows", lcm::MAX_FUSED_QUERY_ROWS);

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
    uint64_t key = mix(mix(mix(0x1234, (uint64_t)h->frames.size()), (uint64_t)n_q_frames), (uint64_t)h->params.min_gap);
    key = mix(mix(key, self ? 1 : 2), (uint64_t)q_stride_rows);
    key = mix(key, h->db_generation);
    for (int i = 0; i < n_q_frames; ++i) key = mix(key, (uint64_t)(uint32_t)q_ids[i]);
    if (q_frame_of) for (int i = 0; i < n_q_frames; ++i) key = mix(key, 0x51ull + q_frame_of[i]);
    for (int32_t c : qc) key = mix(key, (uint64_t)(uint32_t)c);
    if (!h->frames.empty()) key = mix(mix(key, (uint64_t)h->frames.front().id), (uint64_t)h->frames.back().id);
    if (key == 0) key = 1;
    Plan& P = h->plan;
    if (P.key != key) {
        P.key = 0;                        // a failed rebuild must not leave a half-built plan behind the old key
        P.items.clear();
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
        std::vector<int32_t> qn;
        if (self) { qn.resize(n_q_frames); for (int i = 0; i < n_q_frames; ++i) qn[i] = h->frames[i].n; }
        // heaviest query frames first so the tail of the launch is made of short items
        for (int c = n_q_frames - 1; c >= 0; --c) {
            const int e = (int)(P.offsets[c + 1] - P.offsets[c]);
            for (int b = 0; b < e; b += chunk)
                P.items.push_back({q_frame_of ? q_frame_of[c] : (uint32_t)c, (uint32_t)b, (uint32_t)std::min(chunk, e - b), (uint32_t)(P.offsets[c] + b)});
            if (self && e > 0) {
                P.distances += (uint64_t)qn[c] * pre[e];
                P.algo_bytes += pre[e] * 32 + (uint64_t)qn[c] * 32 + 8ull * e;
                P.max_q_rows = std::max(P.max_q_rows, (int)qn[c]);
            }
        }
        if (!self) {
            for (int c = 0; c < n_q_frames; ++c) {
                const int e = (int)(P.offsets[c + 1] - P.offsets[c]);
                if (e > 0) {
                    P.distances += (uint64_t)qc[c] * pre[e];
                    P.algo_bytes += pre[e] * 32 + (uint64_t)qc[c] * 32 + 8ull * e;
                    P.max_q_rows = std::max(P.max_q_rows, (int)qc[c]);
                }
            }
        }
        if (P.max_q_rows > lcm::MAX_FUSED_QUERY_ROWS) return fail(LCM_ERR_CAPACITY, "query frames may hold at most %d rows", lcm::MAX_FUSED_QUERY_ROWS);
        P.n_pairs = total;
        if (!P.items.empty()) {
            rc = ensure_dev(P.d_items, P.d_items_cap, P.items.size()); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(P.d_items, P.items.data(), sizeof(lcm::WorkItem) * P.items.size(), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
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
    if ((h->variant == 4 || h->variant == 5) && !d_idx_sums) {
        std::vector<int> nqv((size_t)n_q_frames);
        for (int c = 0; c < n_q_frames; ++c) nqv[(size_t)c] = self ? h->frames[(size_t)c].n : qc[(size_t)c];
        return mfma_bulk(h, self, self ? h->d_rows : (const uint8_t*)d_query_rows, d_query_counts,
                         (uint32_t)(self ? h->stride_rows : q_stride_rows), q_frame_of, nqv.data(), n_q_frames, P.offsets, (lcm_score*)d_scores);
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
    h->info.launches = launches; h->info.workgroups = biggest;
    h->info.pairs = P.n_pairs; h->info.distances = P.distances; h->info.algo_bytes = P.algo_bytes;
    return LCM_OK;
}

}  // extern "C"  (reopened below)
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

int lcm_last_launch_info(const lcm_handle* hc, lcm_launch_info* info) {
    lcm_handle* h = const_cast<lcm_handle*>(hc);
    if (!h || !info) return fail(LCM_ERR_INVALID_ARG, "NULL argument");
    if (h->info_pending) {
        HIP_TRY(hipEventSynchronize(h->ev_stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_stop));
        h->info.kernel_ms = ms;
        h->info.aux_kernel_ms = 0.0;
        h->info_pending = false;
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

/* ---- exported entry points of the functions above, behind the exception guard ----------------------------- */

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
int lcm_match_pair(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* train_idx, uint16_t* dist, int* n_matches) {
    return guarded([&] { return match_pair_impl(h, query, nq, train, nt, train_idx, dist, n_matches); });
}
int lcm_match_features(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt, lcm_dmatch* out, int* n_out, int* min_dist) {
    return guarded([&] { return match_features_impl(h, query, nq, train, nt, out, n_out, min_dist); });
}
int lcm_match_stored(lcm_handle* h, int query_frame_id, int train_frame_id, lcm_dmatch* out, int cap, int* n_out, int* min_dist) {
    return guarded([&] { return match_stored_impl(h, query_frame_id, train_frame_id, out, cap, n_out, min_dist); });
}
int lcm_query_submit(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, int* ticket) {
    return guarded([&] { return query_submit_impl(h, query, nq, query_frame_id, ticket); });
}
int lcm_query_collect(lcm_handle* h, int ticket, lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out) {
    return guarded([&] { return query_collect_impl(h, ticket, out_scores, out_frame_ids, cap, n_out); });
}
int lcm_match_stored_batch(lcm_handle* h, const lcm_pair_ref* pairs, int n_pairs, lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists) {
    return guarded([&] { return match_batch_impl(h, nullptr, 0, pairs, nullptr, n_pairs, out, cap, offsets, min_dists); });
}
int lcm_match_query_batch(lcm_handle* h, const uint8_t* query, int nq, const int32_t* train_frame_ids, int n_trains, lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists) {
    if (nq < 0 || (nq > 0 && !query)) return fail(LCM_ERR_INVALID_ARG, "bad query rows");
    static const uint8_t none[LCM_DESC_BYTES] = {0};
    return guarded([&] { return match_batch_impl(h, query ? query : none, nq, nullptr, train_frame_ids, n_trains, out, cap, offsets, min_dists); });
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
