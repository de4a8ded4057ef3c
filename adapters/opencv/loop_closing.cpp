// adapters/opencv/loop_closing.cpp — the reference-side binding: defines the hot-path members that the reference's
// OWN header declares (include/loop_closing.hpp:31,34,37,40,48,66 of F-Fer/SLAM-Loop-Closing) on top of the C ABI in
// include/lcm.h, so that a program written against that header links against liblcm_hip.so unchanged.
//
// NOT built in this repository's image (no OpenCV here).  Build where OpenCV 4.x is installed:
//   g++ -std=c++17 -I<reference>/include -I<this repo>/include $(pkg-config --cflags opencv4)
//       -c adapters/opencv/loop_closing.cpp                                        (one command line)
//   ... link with -L<this repo>/slam-loop-closing_amd/lib -llcm_hip $(pkg-config --libs opencv4)
//
// The reference ships no src/loop_closing.cpp, so this file is written from the header and README only.  The header's
// private members cannot change ("drop-in"), so the GPU handle lives in a side table keyed by `this`; the
// cv::Ptr<cv::BFMatcher> matcher_ member (hpp:73) stays empty.  estimatePose / triangulatePoints / visualizeMatches
// are outside the Hamming path and are not defined here.
//
// Limitation of a drop-in for THIS header: it declares no destructor, so nothing tells the adapter when an object
// dies.  The constructor therefore always starts from a fresh matcher (an object built at a recycled address never
// inherits its predecessor's device database), and a program that creates many systems should call
// loop_closing::lcm_release(&system) before the object goes away; otherwise the last handle of each address stays
// allocated until exit.
//
// In this repository the file is only SYNTAX-checked (tests/test_adapter_syntax.py: g++ -fsyntax-only against the
// reference's real header and declaration stubs for the cv:: names) — hygiene, not parity.
#include "loop_closing.hpp"   // the reference's header

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

#include "lcm.h"

namespace loop_closing {
namespace {

std::mutex g_mu;
std::unordered_map<const LoopClosingSystem*, lcm_handle*> g_handles;

[[noreturn]] void raise(const char* what) { throw std::runtime_error(std::string(what) + ": " + lcm_last_error()); }

lcm_handle* handle_for(const LoopClosingSystem* self, double loop_threshold, int min_loop_gap, bool fresh = false) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find(self);
    if (it != g_handles.end()) {
        if (!fresh) return it->second;
        lcm_destroy(it->second);          // a new object at a recycled address: drop the old object's database
        g_handles.erase(it);
    }
    lcm_params p;
    lcm_params_default(&p);
    p.sim_threshold = loop_threshold;
    p.min_gap = min_loop_gap;
    lcm_handle* h = nullptr;
    if (lcm_create(&p, /*device*/ 0, /*stream*/ nullptr, &h) != LCM_OK) raise("lcm_create");
    g_handles.emplace(self, h);
    return h;
}

// cv::Mat(CV_8UC1, n x 32) -> contiguous rows (ORB::detectAndCompute output already is)
const uint8_t* rows_of(const cv::Mat& d, cv::Mat& keep) {
    if (d.empty()) return nullptr;
    CV_Assert(d.type() == CV_8UC1 && d.cols == LCM_DESC_BYTES);
    keep = d.isContinuous() ? d : d.clone();
    return keep.ptr<uint8_t>();
}

}  // namespace

// Explicit release hook (the reference header has no destructor to do it): frees the object's matcher and device database.
void lcm_release(const LoopClosingSystem* self) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find(self);
    if (it == g_handles.end()) return;
    lcm_destroy(it->second);
    g_handles.erase(it);
}

LoopClosingSystem::LoopClosingSystem(double loop_threshold, int min_loop_gap)
    : loop_threshold_(loop_threshold), min_loop_gap_(min_loop_gap) {
    feature_detector_ = cv::ORB::create(2000);   // README.md:114
    K_ = (cv::Mat_<double>(3, 3) << 800, 0, 640, 0, 800, 360, 0, 0, 1);   // README.md:131
    handle_for(this, loop_threshold_, min_loop_gap_, /*fresh*/ true);
}

void LoopClosingSystem::detectFeatures(Frame& frame) {
    feature_detector_->detectAndCompute(frame.image, cv::noArray(), frame.keypoints, frame.descriptors);
}

std::vector<cv::DMatch> LoopClosingSystem::matchFeatures(const Frame& frame1, const Frame& frame2) {
    static_assert(sizeof(cv::DMatch) == sizeof(lcm_dmatch), "cv::DMatch layout");
    lcm_handle* h = handle_for(this, loop_threshold_, min_loop_gap_);
    cv::Mat k1, k2;
    const uint8_t* q = rows_of(frame1.descriptors, k1);
    const uint8_t* t = rows_of(frame2.descriptors, k2);
    std::vector<cv::DMatch> out((size_t)std::max(frame1.descriptors.rows, 1));
    int n = 0, min_dist = 0;
    if (lcm_match_features(h, q, frame1.descriptors.rows, t, frame2.descriptors.rows,
                           reinterpret_cast<lcm_dmatch*>(out.data()), &n, &min_dist) != LCM_OK)
        raise("matchFeatures");
    out.resize((size_t)n);
    return out;
}

std::vector<LoopCandidate> LoopClosingSystem::detectLoops(int current_frame_id) {
    static_assert(sizeof(LoopCandidate) == sizeof(lcm_loop_candidate), "LoopCandidate layout");
    lcm_handle* h = handle_for(this, loop_threshold_, min_loop_gap_);
    const Frame* cur = nullptr;
    for (const Frame& f : frames_) if (f.id == current_frame_id) cur = &f;
    if (!cur) throw std::out_of_range("detectLoops: unknown frame id");
    cv::Mat keep;
    static const uint8_t dummy[LCM_DESC_BYTES] = {0};
    const uint8_t* q = rows_of(cur->descriptors, keep);
    std::vector<LoopCandidate> out((size_t)std::max(lcm_db_size(h), 1));
    int n = 0;
    if (lcm_detect_loops(h, current_frame_id, q ? q : dummy, cur->descriptors.rows, (int)cur->keypoints.size(),
                         reinterpret_cast<lcm_loop_candidate*>(out.data()), (int)out.size(), &n) != LCM_OK)
        raise("detectLoops");
    out.resize((size_t)n);
    return out;
}

void LoopClosingSystem::processFrame(const cv::Mat& image, int frame_id) {
    lcm_handle* h = handle_for(this, loop_threshold_, min_loop_gap_);
    Frame f;
    f.id = frame_id;
    f.image = image;
    detectFeatures(f);
    frames_.push_back(f);
    // consecutive-frame matching / pose / triangulation (README.md:96-99) stay with the caller: outside this path
    std::vector<LoopCandidate> found = detectLoops(frame_id);                    // README.md:100
    loop_closures_.insert(loop_closures_.end(), found.begin(), found.end());
    cv::Mat keep;
    const uint8_t* rows = rows_of(f.descriptors, keep);
    if (lcm_db_append(h, frame_id, rows, f.descriptors.rows, (int)f.keypoints.size()) != LCM_OK) raise("processFrame");
}

void LoopClosingSystem::saveResults(const std::string& output_dir) {
    if (mkdir(output_dir.c_str(), 0777) != 0 && errno != EEXIST)
        throw std::runtime_error("saveResults: cannot create " + output_dir + ": " + strerror(errno));
    std::ofstream os(output_dir + "/loop_closures.txt");
    if (!os) throw std::runtime_error("saveResults: cannot open " + output_dir + "/loop_closures.txt");
    os << "=== Processing Complete ===\n"
       << "Total frames processed: " << frames_.size() << "\n"
       << "Loop closures detected: " << loop_closures_.size() << "\n\n"
       << "Loop Closures Detected:\n======================\n\n";
    for (const LoopCandidate& c : loop_closures_)
        os << "Frame " << c.current_frame_id << " <-> Frame " << c.matched_frame_id << "\n"
           << "  Matches: " << c.num_matches << "\n  Similarity: " << c.similarity_score << "\n\n";
}

}  // namespace loop_closing
