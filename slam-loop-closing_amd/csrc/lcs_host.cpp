// lcs_host.cpp — loop_closing::LoopClosingSystem on top of the C ABI, plus its C shim (include/lcm_host.h).
#include "loop_closing_system.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <new>

#include "../../include/lcm_host.h"

namespace lcm { void set_last_error(const char* msg); }

namespace loop_closing {

namespace {
[[noreturn]] void raise(const char* what) {
    throw std::runtime_error(std::string(what) + ": " + lcm_last_error());
}
}  // namespace

LoopClosingSystem::LoopClosingSystem(double loop_threshold, int min_loop_gap, int device_id, int shard_rank, int shard_world)
    : loop_threshold_(loop_threshold), min_loop_gap_(min_loop_gap), shard_rank_(shard_rank), shard_world_(shard_world) {
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) throw std::invalid_argument("bad shard_rank / shard_world");
    lcm_params p;
    lcm_params_default(&p);
    p.sim_threshold = loop_threshold;
    p.min_gap = min_loop_gap;
    if (lcm_create(&p, device_id, nullptr, &matcher_) != LCM_OK) raise("LoopClosingSystem: lcm_create");
}

LoopClosingSystem::LoopClosingSystem(double loop_threshold, int min_loop_gap, const std::vector<int>& device_ids)
    : loop_threshold_(loop_threshold), min_loop_gap_(min_loop_gap), shard_rank_(0), shard_world_(1) {
    if (device_ids.empty()) throw std::invalid_argument("LoopClosingSystem: empty device list");
    lcm_params p;
    lcm_params_default(&p);
    p.sim_threshold = loop_threshold;
    p.min_gap = min_loop_gap;
    if (lcm_group_create(&p, (int)device_ids.size(), device_ids.data(), &group_) != LCM_OK) raise("LoopClosingSystem: lcm_group_create");
    if (lcm_group_handle(group_, 0, &matcher_) != LCM_OK) { lcm_group_destroy(group_); group_ = nullptr; raise("LoopClosingSystem: lcm_group_handle"); }
}

LoopClosingSystem::LoopClosingSystem(double loop_threshold, int min_loop_gap, Loopback rehearsal)
    : loop_threshold_(loop_threshold), min_loop_gap_(min_loop_gap), shard_rank_(0), shard_world_(1) {
    lcm_params p;
    lcm_params_default(&p);
    p.sim_threshold = loop_threshold;
    p.min_gap = min_loop_gap;
    if (lcm_group_create_loopback(&p, rehearsal.n_shards, rehearsal.device_id, &group_) != LCM_OK) raise("LoopClosingSystem: lcm_group_create_loopback");
    if (lcm_group_handle(group_, 0, &matcher_) != LCM_OK) { lcm_group_destroy(group_); group_ = nullptr; raise("LoopClosingSystem: lcm_group_handle"); }
}

LoopClosingSystem::~LoopClosingSystem() {
    if (group_) lcm_group_destroy(group_);      // owns every shard's matcher, matcher_ included
    else lcm_destroy(matcher_);
}

void LoopClosingSystem::setGapByPosition(bool on) {
    if (!frames_.empty()) throw std::invalid_argument("setGapByPosition: only before the first frame is processed");
    gap_by_position_ = on;
}

const Frame* LoopClosingSystem::findFrame(int frame_id) const {
    auto it = std::lower_bound(frames_.begin(), frames_.end(), frame_id, [](const Frame& f, int id) { return f.id < id; });
    return (it != frames_.end() && it->id == frame_id) ? &*it : nullptr;
}

// ---- the three steps every frame goes through, for one device, one shard of a process-per-GPU run, or a group ----------
// submit: frames_[first, first + count) — on the host list already, none of them in the device database yet — are sent
// to the device(s) as ONE asynchronous micro-batch (one launch per device) against the database as it stands.
int LoopClosingSystem::submitBatch(size_t first, size_t count) {
    static const uint8_t dummy[32] = {0};
    std::vector<const uint8_t*> q(count);
    std::vector<int> nq(count), keys(count);
    for (size_t k = 0; k < count; ++k) {
        const Frame& f = frames_[first + k];
        q[k] = f.rows() > 0 ? f.descriptors.data() : dummy;
        nq[k] = f.rows();
        keys[k] = keyOf(first + k);
    }
    int ticket = -1;
    const int rc = group_ ? lcm_group_query_submit_batch(group_, q.data(), nq.data(), keys.data(), (int)count, &ticket)
                          : lcm_query_submit_batch(matcher_, q.data(), nq.data(), keys.data(), (int)count, &ticket);
    if (rc != LCM_OK) raise("processFrame: query submit");
    return ticket;
}

// collect: wait for that batch, apply the loop test (README.md:123-126) and record the closures, query by query.
void LoopClosingSystem::collectBatch(int ticket, size_t first, size_t count) {
    const size_t n_db = (size_t)std::max(group_ ? lcm_group_db_size(group_) : lcm_db_size(matcher_), 0);
    std::vector<lcm_score> sc(std::max<size_t>(n_db * count, 1));
    std::vector<size_t> offs(count + 1, 0);
    size_t n = 0;
    const int rc = group_ ? lcm_group_query_collect_batch(group_, ticket, sc.data(), sc.size(), &n, offs.data())
                          : lcm_query_collect_batch(matcher_, ticket, sc.data(), sc.size(), &n, offs.data());
    if (rc != LCM_OK) raise("processFrame: query collect");
    // Record k of query b is its k-th eligible stored frame.  One device (or a group): that is position k.  One shard of
    // a process-per-GPU run: the k-th frame THIS rank owns, position shard_rank + k * shard_world.
    lcm_params p;
    if (lcm_get_params(matcher_, &p) != LCM_OK) raise("processFrame: lcm_get_params");
    for (size_t b = 0; b < count; ++b) {
        const Frame& cur = frames_[first + b];
        for (size_t k = offs[b]; k < offs[b + 1]; ++k) {
            const size_t pos = group_ ? (k - offs[b]) : (size_t)shard_rank_ + (k - offs[b]) * (size_t)shard_world_;
            const Frame& past = frames_[pos];
            double sim = 0.0;
            if (lcm_loop_test(&p, &sc[k], cur.num_keypoints, past.num_keypoints, &sim))
                loop_closures_.push_back({cur.id, past.id, (int)sc[k].good_count, sim});
        }
    }
}

// a ticket whose result is no longer wanted (an error is on its way up): wait for it, keep the first error's message
void LoopClosingSystem::drainBatch(int ticket, size_t count) noexcept {
    try {
        const std::string why = lcm_last_error();
        const size_t n_db = (size_t)std::max(group_ ? lcm_group_db_size(group_) : lcm_db_size(matcher_), 0);
        std::vector<lcm_score> sink(std::max<size_t>(n_db * count, 1));
        size_t n = 0;
        if (group_) (void)lcm_group_query_collect_batch(group_, ticket, sink.data(), sink.size(), &n, nullptr);
        else (void)lcm_query_collect_batch(matcher_, ticket, sink.data(), sink.size(), &n, nullptr);
        lcm::set_last_error(why.c_str());
    } catch (...) {}
}

// store: the frame joins the device database (if this rank owns its position)
void LoopClosingSystem::storeFrame(size_t pos) {
    const Frame& s = frames_[pos];
    if (group_) {
        if (lcm_group_append(group_, keyOf(pos), s.descriptors.data(), s.rows(), s.num_keypoints) != LCM_OK) raise("processFrame: lcm_group_append");
    } else if (ownsPosition(pos)) {
        if (lcm_db_append(matcher_, keyOf(pos), s.descriptors.data(), s.rows(), s.num_keypoints) != LCM_OK) raise("processFrame: lcm_db_append");
    }
}

// the device database back to the first n_frames frames of frames_ (rollback; errors are swallowed: one is on its way up)
void LoopClosingSystem::truncateDevice(size_t n_frames) noexcept {
    const std::string why = lcm_last_error();
    if (group_) (void)lcm_group_truncate(group_, (int)n_frames);
    else {
        const size_t r = (size_t)shard_rank_, w = (size_t)shard_world_;
        (void)lcm_db_truncate(matcher_, (int)(n_frames > r ? (n_frames - r + w - 1) / w : 0));
    }
    lcm::set_last_error(why.c_str());
}

void LoopClosingSystem::processFrame(const uint8_t* descriptors, int rows, int num_keypoints, int frame_id) {
    if (rows < 0 || (rows > 0 && !descriptors)) throw std::invalid_argument("processFrame: bad descriptors");
    if (!frames_.empty() && frame_id <= frames_.back().id) throw std::invalid_argument("processFrame: frame ids must increase");
    Frame f;
    f.id = frame_id;
    f.num_keypoints = num_keypoints < 0 ? rows : num_keypoints;
    f.descriptors.assign(descriptors, descriptors + (size_t)rows * 32);
    frames_.push_back(std::move(f));
    const size_t pos = frames_.size() - 1;
    const size_t n_loops_before = loop_closures_.size();
    bool stored = false;
    try {
        // The loop-closure query (the current frame against the frames stored so far, README.md:100,122) is SUBMITTED first
        // — asynchronous, on a query slot's own stream, on every device of a group from that device's host thread — the
        // consecutive-frame pair match (README.md:96-97: previous frame = query, current = train, as the tree's own
        // incremental loop orders them, src/main.cpp:1154) then runs on the handle's stream beside it, and the records are
        // collected afterwards: the two steps overlap on the device instead of queueing.
        const int ticket = submitBatch(pos, 1);
        consecutive_matches_.clear();
        try {
            if (frames_.size() >= 2) consecutive_matches_ = matchFeatures(frames_[frames_.size() - 2], frames_.back());
        } catch (...) {
            drainBatch(ticket, 1);
            throw;
        }
        collectBatch(ticket, pos, 1);
        // ... then the frame joins the device database
        storeFrame(pos);
        stored = true;
    } catch (...) {
        // strong guarantee: a frame that could not be processed leaves no trace (host list, loop list, device DB agree)
        (void)stored;
        frames_.pop_back();
        loop_closures_.resize(n_loops_before);
        throw;
    }
}

void LoopClosingSystem::processFrames(const FrameInput* in, int n) {
    if (n < 0 || (n > 0 && !in)) throw std::invalid_argument("processFrames: bad argument");
    int last_id = frames_.empty() ? 0 : frames_.back().id;
    for (int i = 0; i < n; ++i) {
        if (in[i].rows < 0 || (in[i].rows > 0 && !in[i].descriptors)) throw std::invalid_argument("processFrames: bad descriptors");
        if ((i > 0 || !frames_.empty()) && in[i].frame_id <= last_id) throw std::invalid_argument("processFrames: frame ids must increase");
        last_id = in[i].frame_id;
    }
    const int span = std::max(min_loop_gap_, 1);            // frames of one launch must not be eligible for one another
    // Two micro-batches in flight: batch k + 1 is submitted (and its frames stored) before batch k is collected, so the
    // devices never wait for the host between launches.  `done` = frames whose loop check has been collected and recorded.
    struct InFlight { int ticket = -1; size_t first = 0, count = 0; };
    InFlight pending;
    size_t done = frames_.size();
    const size_t closures_before = loop_closures_.size();
    (void)closures_before;
    int i = 0;
    try {
        while (i < n) {
            const size_t first = frames_.size();
            int cnt = 0;
            // greedy cut: at most 16 frames, key(last) - key(first) < span
            while (i + cnt < n && cnt < 16) {
                const long long k0 = gap_by_position_ ? (long long)first : (long long)in[i].frame_id;
                const long long kc = gap_by_position_ ? (long long)(first + (size_t)cnt) : (long long)in[i + cnt].frame_id;
                if (cnt > 0 && kc - k0 >= span) break;
                ++cnt;
            }
            for (int k = 0; k < cnt; ++k) {
                Frame f;
                f.id = in[i + k].frame_id;
                f.num_keypoints = in[i + k].num_keypoints < 0 ? in[i + k].rows : in[i + k].num_keypoints;
                f.descriptors.assign(in[i + k].descriptors, in[i + k].descriptors + (size_t)in[i + k].rows * 32);
                frames_.push_back(std::move(f));
            }
            InFlight cur;
            cur.first = first; cur.count = (size_t)cnt;
            cur.ticket = submitBatch(first, (size_t)cnt);
            try {
                for (int k = 0; k < cnt; ++k) storeFrame(first + (size_t)k);     // copy stream: overlaps the launch just submitted
            } catch (...) {
                drainBatch(cur.ticket, cur.count);
                throw;
            }
            if (pending.ticket >= 0) {
                const InFlight p = pending;
                pending = cur;                              // if the collect throws, `cur` is what remains to be drained
                collectBatch(p.ticket, p.first, p.count);
                done = p.first + p.count;
            } else {
                pending = cur;
            }
            i += cnt;
        }
        if (pending.ticket >= 0) {
            const InFlight p = pending;
            pending.ticket = -1;
            collectBatch(p.ticket, p.first, p.count);
            done = p.first + p.count;
        }
    } catch (...) {
        // Host list, loop list and device database must agree: the frames whose loop check completed stay (with their
        // closures, which are the only ones recorded); everything after them is taken back, on the device(s) too.
        if (pending.ticket >= 0) drainBatch(pending.ticket, pending.count);
        truncateDevice(done);
        frames_.resize(done);
        while (!loop_closures_.empty() && !findFrame(loop_closures_.back().current_frame_id)) loop_closures_.pop_back();
        throw;
    }
    // consecutive-frame matches of the LAST frame (README.md:96-97): what getConsecutiveMatches() would hold after
    // frame-by-frame processing
    if (n > 0) {
        consecutive_matches_.clear();
        if (frames_.size() >= 2) consecutive_matches_ = matchFeatures(frames_[frames_.size() - 2], frames_.back());
    }
}

std::vector<DMatch> LoopClosingSystem::matchFeatures(const Frame& frame1, const Frame& frame2) {
    static_assert(sizeof(DMatch) == sizeof(lcm_dmatch), "DMatch layout");
    std::vector<DMatch> out((size_t)std::max(frame1.rows(), 1));
    int n = 0, md = 0;
    if (lcm_match_features(matcher_, frame1.descriptors.data(), frame1.rows(), frame2.descriptors.data(), frame2.rows(),
                           reinterpret_cast<lcm_dmatch*>(out.data()), &n, &md) != LCM_OK)
        raise("matchFeatures");
    out.resize((size_t)n);
    return out;
}

std::vector<LoopCandidate> LoopClosingSystem::detectLoops(int current_frame_id) {
    static_assert(sizeof(LoopCandidate) == sizeof(lcm_loop_candidate), "LoopCandidate layout");
    const Frame* cur = findFrame(current_frame_id);
    if (!cur) throw std::out_of_range("detectLoops: unknown frame id");
    const int cap = std::max(group_ ? lcm_group_db_size(group_) : lcm_db_size(matcher_), 1);
    std::vector<LoopCandidate> out((size_t)cap);
    int n = 0;
    static const uint8_t dummy[32] = {0};
    const uint8_t* q = cur->rows() > 0 ? cur->descriptors.data() : dummy;
    const int key = keyOf((size_t)(cur - frames_.data()));       // the frame's id, or its arrival position (setGapByPosition)
    const int rc = group_ ? lcm_group_detect_loops(group_, key, q, cur->rows(), cur->num_keypoints,
                                                   reinterpret_cast<lcm_loop_candidate*>(out.data()), cap, &n)
                          : lcm_detect_loops(matcher_, key, q, cur->rows(), cur->num_keypoints,
                                             reinterpret_cast<lcm_loop_candidate*>(out.data()), cap, &n);
    if (rc != LCM_OK) raise("detectLoops");
    out.resize((size_t)n);
    if (gap_by_position_)                                        // report the caller's frame ids, whatever keys the database
        for (LoopCandidate& c : out) { c.current_frame_id = current_frame_id; c.matched_frame_id = frames_[(size_t)c.matched_frame_id].id; }
    return out;
}

std::vector<std::vector<DMatch>> LoopClosingSystem::matchLoopClosures(int current_frame_id) {
    const Frame* cur = findFrame(current_frame_id);
    if (!cur) throw std::out_of_range("matchLoopClosures: unknown frame id");
    std::vector<int32_t> trains;
    for (const LoopCandidate& c : loop_closures_)
        if (c.current_frame_id == current_frame_id) trains.push_back(c.matched_frame_id);
    std::vector<std::vector<DMatch>> lists(trains.size());
    if (trains.empty()) return lists;
    static const uint8_t dummy[32] = {0};
    const uint8_t* q = cur->rows() > 0 ? cur->descriptors.data() : dummy;
    // one launch per device that stores some of the matched frames (one launch in all without a group)
    const int world = group_ ? lcm_group_size(group_) : 1;
    for (int r = 0; r < world; ++r) {
        std::vector<int32_t> mine;
        std::vector<size_t> where;
        for (size_t i = 0; i < trains.size(); ++i) {
            const Frame* t = findFrame(trains[i]);
            if (!t) throw std::out_of_range("matchLoopClosures: a recorded loop closure names an unknown frame");
            const size_t pos = (size_t)(t - frames_.data());
            if (!group_ || (int)(pos % (size_t)world) == r) { mine.push_back(keyOf(pos)); where.push_back(i); }
        }
        if (mine.empty()) continue;
        lcm_handle* h = matcher_;
        if (group_ && lcm_group_handle(group_, r, &h) != LCM_OK) raise("matchLoopClosures: lcm_group_handle");
        std::vector<DMatch> flat((size_t)std::max(cur->rows(), 1) * mine.size());
        std::vector<size_t> offs(mine.size() + 1, 0);
        if (lcm_match_query_batch(h, q, cur->rows(), mine.data(), (int)mine.size(), reinterpret_cast<lcm_dmatch*>(flat.data()), flat.size(),
                                  offs.data(), nullptr) != LCM_OK)
            raise("matchLoopClosures");
        for (size_t k = 0; k < mine.size(); ++k) lists[where[k]].assign(flat.begin() + (ptrdiff_t)offs[k], flat.begin() + (ptrdiff_t)offs[k + 1]);
    }
    return lists;
}

void LoopClosingSystem::saveResults(const std::string& output_dir) {
    if (mkdir(output_dir.c_str(), 0777) != 0 && errno != EEXIST)
        throw std::runtime_error("saveResults: cannot create " + output_dir + ": " + strerror(errno));
    const std::string path = output_dir + "/loop_closures.txt";
    std::ofstream os(path);
    if (!os) throw std::runtime_error("saveResults: cannot open " + path);
    // README.md:150-166
    os << "=== Processing Complete ===\n";
    os << "Total frames processed: " << frames_.size() << "\n";
    os << "Loop closures detected: " << loop_closures_.size() << "\n\n";
    os << "Loop Closures Detected:\n======================\n\n";
    for (const LoopCandidate& c : loop_closures_) {
        os << "Frame " << c.current_frame_id << " <-> Frame " << c.matched_frame_id << "\n";
        os << "  Matches: " << c.num_matches << "\n";
        os << "  Similarity: " << c.similarity_score << "\n\n";
    }
    if (!os) throw std::runtime_error("saveResults: write failed: " + path);
}

}  // namespace loop_closing

// ---------------------------------------------------------------------------------------------------------
// C shim
// ---------------------------------------------------------------------------------------------------------
struct lcs_system {
    loop_closing::LoopClosingSystem sys;
    lcs_system(double thr, int gap, int dev, int r, int w) : sys(thr, gap, dev, r, w) {}
    lcs_system(double thr, int gap, const std::vector<int>& devs) : sys(thr, gap, devs) {}
    lcs_system(double thr, int gap, loop_closing::LoopClosingSystem::Loopback lb) : sys(thr, gap, lb) {}
};

namespace lcm { void set_last_error(const char* msg); }

namespace {
// exceptions never cross the C boundary: they become a status + the lcm_last_error() message
template <typename F>
int guarded(F&& f) {
    try { f(); return LCM_OK; }
    catch (const std::invalid_argument& e) { lcm::set_last_error(e.what()); return LCM_ERR_INVALID_ARG; }
    catch (const std::out_of_range& e) { lcm::set_last_error(e.what()); return LCM_ERR_NOT_FOUND; }
    catch (const std::bad_alloc&) { lcm::set_last_error("out of memory"); return LCM_ERR_OOM; }
    catch (const std::exception& e) { lcm::set_last_error(e.what()); return LCM_ERR_HIP; }
}
}  // namespace

extern "C" {

int lcs_create(double loop_threshold, int min_loop_gap, int device_id, int shard_rank, int shard_world, lcs_system** out) {
    if (!out) return LCM_ERR_INVALID_ARG;
    *out = nullptr;
    if (lcm_device_count() <= 0) {
        lcm_handle* probe = nullptr;
        return lcm_create(nullptr, device_id, nullptr, &probe);   // sets the "no device, no CPU fallback" message
    }
    return guarded([&] { *out = new lcs_system(loop_threshold, min_loop_gap, device_id, shard_rank, shard_world); });
}
int lcs_create_group(double loop_threshold, int min_loop_gap, int n_devices, const int* device_ids, int loopback_device, lcs_system** out) {
    if (!out || n_devices < 1) return LCM_ERR_INVALID_ARG;
    *out = nullptr;
    if (lcm_device_count() <= 0) {
        lcm_handle* probe = nullptr;
        return lcm_create(nullptr, 0, nullptr, &probe);
    }
    return guarded([&] {
        if (loopback_device >= 0) { *out = new lcs_system(loop_threshold, min_loop_gap, loop_closing::LoopClosingSystem::Loopback{n_devices, loopback_device}); return; }
        std::vector<int> devs((size_t)n_devices);
        for (int i = 0; i < n_devices; ++i) devs[(size_t)i] = device_ids ? device_ids[i] : i;
        *out = new lcs_system(loop_threshold, min_loop_gap, devs);
    });
}
void lcs_destroy(lcs_system* s) { delete s; }

int lcs_process_frame(lcs_system* s, const uint8_t* desc, int rows, int n_keypoints, int frame_id) {
    if (!s) return LCM_ERR_INVALID_ARG;
    return guarded([&] { s->sys.processFrame(desc, rows, n_keypoints, frame_id); });
}

int lcs_process_frames(lcs_system* s, const uint8_t* const* desc, const int* rows, const int* n_keypoints, const int* frame_ids, int n) {
    if (!s || n < 0 || (n > 0 && (!desc || !rows || !frame_ids))) return LCM_ERR_INVALID_ARG;
    return guarded([&] {
        std::vector<loop_closing::LoopClosingSystem::FrameInput> in((size_t)n);
        for (int i = 0; i < n; ++i) in[(size_t)i] = {desc[i], rows[i], n_keypoints ? n_keypoints[i] : -1, frame_ids[i]};
        s->sys.processFrames(in.data(), n);
    });
}

int lcs_match_features(lcs_system* s, int frame1_id, int frame2_id, lcm_dmatch* out, int cap, int* n_out) {
    if (!s || !n_out) return LCM_ERR_INVALID_ARG;
    *n_out = 0;
    return guarded([&] {
        const loop_closing::Frame* a = s->sys.findFrame(frame1_id);
        const loop_closing::Frame* b = s->sys.findFrame(frame2_id);
        if (!a || !b) throw std::out_of_range("matchFeatures: unknown frame id");
        auto m = s->sys.matchFeatures(*a, *b);
        if ((int)m.size() > cap) throw std::invalid_argument("matchFeatures: output buffer too small");
        if (!m.empty()) memcpy(out, m.data(), m.size() * sizeof(lcm_dmatch));
        *n_out = (int)m.size();
    });
}

int lcs_detect_loops(lcs_system* s, int current_frame_id, lcm_loop_candidate* out, int cap, int* n_out) {
    if (!s || !n_out) return LCM_ERR_INVALID_ARG;
    *n_out = 0;
    return guarded([&] {
        auto c = s->sys.detectLoops(current_frame_id);
        if ((int)c.size() > cap) throw std::invalid_argument("detectLoops: output buffer too small");
        if (!c.empty()) memcpy(out, c.data(), c.size() * sizeof(lcm_loop_candidate));
        *n_out = (int)c.size();
    });
}

int lcs_get_consecutive_matches(const lcs_system* s, lcm_dmatch* out, int cap, int* n_out) {
    if (!s || !n_out) return LCM_ERR_INVALID_ARG;
    const auto& v = s->sys.getConsecutiveMatches();
    if ((int)v.size() > cap) return LCM_ERR_CAPACITY;
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(lcm_dmatch));
    *n_out = (int)v.size();
    return LCM_OK;
}

int lcs_match_loop_closures(lcs_system* s, int current_frame_id, lcm_dmatch* out, size_t cap, size_t* offsets, int offsets_cap, int* n_lists) {
    if (!s || !n_lists || !offsets) return LCM_ERR_INVALID_ARG;
    *n_lists = 0;
    return guarded([&] {
        auto lists = s->sys.matchLoopClosures(current_frame_id);
        if ((int)lists.size() + 1 > offsets_cap) throw std::invalid_argument("matchLoopClosures: offsets buffer too small");
        size_t k = 0;
        for (size_t i = 0; i < lists.size(); ++i) {
            offsets[i] = k;
            if (k + lists[i].size() > cap) throw std::invalid_argument("matchLoopClosures: output buffer too small");
            if (!lists[i].empty()) memcpy(out + k, lists[i].data(), lists[i].size() * sizeof(lcm_dmatch));
            k += lists[i].size();
        }
        offsets[lists.size()] = k;
        *n_lists = (int)lists.size();
    });
}

int lcs_set_gap_by_position(lcs_system* s, int on) {
    if (!s) return LCM_ERR_INVALID_ARG;
    return guarded([&] { s->sys.setGapByPosition(on != 0); });
}

int lcs_num_frames(const lcs_system* s) { return s ? (int)s->sys.getFrames().size() : 0; }
int lcs_num_loop_closures(const lcs_system* s) { return s ? (int)s->sys.getLoopClosures().size() : 0; }

int lcs_get_loop_closures(const lcs_system* s, lcm_loop_candidate* out, int cap, int* n_out) {
    if (!s || !n_out) return LCM_ERR_INVALID_ARG;
    const auto& v = s->sys.getLoopClosures();
    if ((int)v.size() > cap) return LCM_ERR_CAPACITY;
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(lcm_loop_candidate));
    *n_out = (int)v.size();
    return LCM_OK;
}

int lcs_save_results(lcs_system* s, const char* output_dir) {
    if (!s || !output_dir) return LCM_ERR_INVALID_ARG;
    return guarded([&] { s->sys.saveResults(output_dir); });
}

}  // extern "C"
