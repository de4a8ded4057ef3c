// loop_closing_system.hpp — host-side mirror of the reference's loop_closing::LoopClosingSystem
// (include/loop_closing.hpp:9-82) for the ORB/Hamming hot path, free of OpenCV types so it builds in an image
// without OpenCV.  Same class / method names, same argument meaning, same error behaviour (exceptions derived
// from std::exception, which the reference's main() catches: src/main.cpp:1043,1680).  Everything that computes
// a Hamming distance goes through the C ABI (include/lcm.h) into the gfx950 kernels.
//
// What differs from the reference header, and why:
//   * Frame::descriptors is a byte vector (rows x 32, row-major) instead of cv::Mat(CV_8UC1); Frame::num_keypoints
//     replaces keypoints.size().  image / pose / points3D are not on this path and are not stored.
//   * processFrame takes the descriptors ORB would have produced: feature detection (detectFeatures, hpp:37) is
//     outside the hot path.
//   * estimatePose / triangulatePoints / visualizeMatches are out of scope and absent.
// With OpenCV available, adapters/opencv/loop_closing.cpp defines the reference header's own members on top of this.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

struct lcm_handle;
struct lcm_group;

namespace loop_closing {

// cv::DMatch field order (consumed as m.queryIdx / m.trainIdx at src/main.cpp:553-554).
struct DMatch {
    int queryIdx;
    int trainIdx;
    int imgIdx;
    float distance;
};

struct Frame {                       // include/loop_closing.hpp:12-19, hot-path fields only
    int id = 0;
    int num_keypoints = 0;           // keypoints.size(): denominator of the similarity score (README.md:126)
    std::vector<uint8_t> descriptors;  // rows x 32 bytes
    int rows() const { return (int)(descriptors.size() / 32); }
};

struct LoopCandidate {               // include/loop_closing.hpp:22-27, identical
    int current_frame_id;
    int matched_frame_id;
    int num_matches;
    double similarity_score;
};

// exported from liblcm_hip.so (built with -fvisibility=hidden): C++ callers link against the class directly
class __attribute__((visibility("default"))) LoopClosingSystem {
public:
    // include/loop_closing.hpp:31 — same defaults.  device_id / shard_* are additions with defaults that keep the
    // reference's two-argument construction valid.
    explicit LoopClosingSystem(double loop_threshold = 0.7, int min_loop_gap = 30, int device_id = 0,
                               int shard_rank = 0, int shard_world = 1);
    // The same system over SEVERAL MI355X of one node (one process): stored frames are sharded cyclically over
    // `device_ids` behind an lcm_group (one matcher + host thread per device, RCCL inside); every member function
    // behaves as with one device and returns the same results.  A C++ host needs no RCCL code of its own.
    LoopClosingSystem(double loop_threshold, int min_loop_gap, const std::vector<int>& device_ids);
    // Rehearsal of the multi-device form on a box with ONE GPU: n_shards shards on device_id (lcm_group_create_loopback:
    // the exchange steps are device-local copies).  Same results; exists for tests.
    struct Loopback { int n_shards; int device_id; };
    LoopClosingSystem(double loop_threshold, int min_loop_gap, Loopback rehearsal);
    ~LoopClosingSystem();
    LoopClosingSystem(const LoopClosingSystem&) = delete;
    LoopClosingSystem& operator=(const LoopClosingSystem&) = delete;

    // Process a single frame (hpp:34): store it, then check for loop closures (README.md:94-100).
    void processFrame(const uint8_t* descriptors, int rows, int num_keypoints, int frame_id);

    // Several frames at once, same observable result as calling processFrame for each in order (frames, loop closures,
    // the consecutive matches of the last frame), but scored as micro-batches: up to 16 frames per kernel launch
    // (lcm_query_submit_batch), which is what keeps an MI355X busy when frames arrive faster than one launch per frame
    // can serve them.  Frames of one launch are not compared with each other, so a launch only ever
    // holds frames closer together than min_loop_gap (in ids, or in positions under setGapByPosition): the call cuts
    // its input accordingly and is exact for any input.  Two micro-batches are in flight (batch k + 1 is submitted and
    // its frames stored before batch k's records are collected), on one device, on one shard, or on every device of a
    // group.  On an exception the frames whose loop check completed stay — with their closures — and every later frame
    // is taken back from the host lists and from the device database(s).
    struct FrameInput { const uint8_t* descriptors; int rows; int num_keypoints; int frame_id; };
    void processFrames(const FrameInput* frames, int n);

    // Match features between two frames (hpp:40): BFMatcher(NORM_HAMMING).match + 2 x min-distance filter.
    std::vector<DMatch> matchFeatures(const Frame& frame1, const Frame& frame2);

    // Check for loop closure (hpp:48): current frame vs every stored frame >= min_loop_gap older.
    std::vector<LoopCandidate> detectLoops(int current_frame_id);
    // BASELINE.json's north_star spells this one detectLoopClosure.
    std::vector<LoopCandidate> detectLoopClosure(int current_frame_id) { return detectLoops(current_frame_id); }

    // "Re-match features on identified loop frames" (README.md:101): the DMatch lists of all loop closures recorded for
    // `current_frame_id` (same order as they appear in getLoopClosures()), query = that frame, train = the matched
    // frame — every pair in ONE kernel launch (lcm_match_query_batch).  Only closures whose matched frame this rank
    // stores can be re-matched (all of them when shard_world == 1).
    std::vector<std::vector<DMatch>> matchLoopClosures(int current_frame_id);

    // Matches between the previous and the current frame, as processFrame's consecutive-frame step computes them
    // (README.md:96-97 "Feature matching between consecutive frames"; what estimatePose / triangulatePoints would consume).
    const std::vector<DMatch>& getConsecutiveMatches() const { return consecutive_matches_; }

    const std::vector<Frame>& getFrames() const { return frames_; }                      // hpp:60
    const std::vector<LoopCandidate>& getLoopClosures() const { return loop_closures_; }  // hpp:63

    // Save results to file (hpp:66): writes <output_dir>/loop_closures.txt in the README's format (README.md:142-165).
    void saveResults(const std::string& output_dir);

    // How "at least min_loop_gap frames ago" (README.md:122) is counted.  Default (false): on frame IDS, c.id - p.id >=
    // min_loop_gap — the reading of the C ABI (include/lcm.h).  true: on ARRIVAL POSITIONS, the c-th processed frame
    // against the frames processed at least min_loop_gap frames before it — the reading of the tree's own loop
    // (src/main.cpp:1375-1379, `past <= curr - loopGap` over keyframe indices).  The two coincide for dense ids
    // 0, 1, 2, ... and differ for sparse ones (video frame numbers with frame_skip = 3).  The device database is then
    // keyed by position; every id this class reports is still the caller's frame id.  Only before the first frame.
    void setGapByPosition(bool on);
    bool gapByPosition() const { return gap_by_position_; }

    double loopThreshold() const { return loop_threshold_; }
    int minLoopGap() const { return min_loop_gap_; }
    const Frame* findFrame(int frame_id) const;
    bool ownsPosition(size_t position) const { return (int)(position % (size_t)shard_world_) == shard_rank_; }

private:
    std::vector<Frame> frames_;
    std::vector<LoopCandidate> loop_closures_;
    std::vector<DMatch> consecutive_matches_;
    lcm_handle* matcher_ = nullptr;      // stands where cv::Ptr<cv::BFMatcher> matcher_ stood (hpp:73)
    lcm_group* group_ = nullptr;  // multi-device construction: the shards' matchers live in here (matcher_ = shard 0's)
    double loop_threshold_;              // hpp:75
    int min_loop_gap_;                   // hpp:76
    int shard_rank_, shard_world_;
    bool gap_by_position_ = false;
    // frames_[first, first + count) as one micro-batch: asynchronous submit (one launch per device), collect + loop test,
    // store; drainBatch / truncateDevice serve the rollback when an error is on its way up
    int submitBatch(size_t first, size_t count);
    void collectBatch(int ticket, size_t first, size_t count);
    void drainBatch(int ticket, size_t count) noexcept;
    void storeFrame(size_t position);
    void truncateDevice(size_t n_frames) noexcept;
    int keyOf(size_t position) const { return gap_by_position_ ? (int)position : frames_[position].id; }   // the matcher's id of a frame
};

using LoopClosing = LoopClosingSystem;   // north_star's spelling of the class

}  // namespace loop_closing
