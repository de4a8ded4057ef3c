// host_class_demo.cpp — a plain C++ program written the way a user of the reference's LoopClosingSystem would write
// it (construct, processFrame per frame, read getLoopClosures, matchFeatures on a detected loop, saveResults), linked
// against liblcm_hip.so only.  Exercised on the GPU box by tests/test_gpu_cpp_demo.py; prints a line per loop so the
// Python side can compare against the oracle.  Inputs: deterministic xorshift descriptors with planted revisits.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <vector>

#include "../../slam-loop-closing_amd/csrc/loop_closing_system.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {
    uint64_t x = rng_state;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return rng_state = x;
}

int main(int argc, char** argv) {
    const int n_frames = argc > 1 ? atoi(argv[1]) : 24;
    const int rows = argc > 2 ? atoi(argv[2]) : 300;
    const char* out_dir = argc > 3 ? argv[3] : "/tmp/lcm_demo_results";
    const bool group_mode = argc > 4 && argv[4][0] == 'g';  // "group": the multi-device constructor over device 0
    try {
        // north_star's spelling of the class; README.md:108-109 style values.  In group mode the SAME program runs on
        // an lcm_group (RCCL communicator, cyclic sharding) with no collective code of its own.
        std::unique_ptr<loop_closing::LoopClosing> holder(group_mode ? new loop_closing::LoopClosing(0.15, 5, std::vector<int>{0})
                                                                     : new loop_closing::LoopClosing(0.15, 5));
        loop_closing::LoopClosing& system = *holder;
        const int n_places = 4;
        std::vector<std::vector<uint8_t>> place(n_places, std::vector<uint8_t>((size_t)rows * 32));
        for (auto& p : place) for (auto& b : p) b = (uint8_t)next_u64();
        std::vector<std::vector<uint8_t>> all_frames;
        for (int f = 0; f < n_frames; ++f) {
            std::vector<uint8_t> d = place[f % n_places];      // revisit the place every 4 frames ...
            for (int r = 0; r < rows; ++r) {
                if (r % 3 == 0) for (int k = 0; k < 32; ++k) d[(size_t)r * 32 + k] = (uint8_t)next_u64();   // ... with 1/3 new rows
                else d[(size_t)r * 32 + (next_u64() % 32)] ^= (uint8_t)(1u << (next_u64() % 8));             // and a flipped bit elsewhere
            }
            system.processFrame(d.data(), rows, rows, f);
            for (size_t i = 0; i < d.size(); ++i) printf("%02x", d[i]);
            printf("\n");
            all_frames.push_back(std::move(d));
        }
        {   // the same sequence through processFrames (micro-batched scoring): same loop closures, same consecutive matches
            loop_closing::LoopClosing batched(0.15, 5);
            std::vector<loop_closing::LoopClosing::FrameInput> in;
            for (int f = 0; f < n_frames; ++f) in.push_back({all_frames[(size_t)f].data(), rows, rows, f});
            batched.processFrames(in.data(), (int)in.size());
            bool same = batched.getLoopClosures().size() == system.getLoopClosures().size() &&
                        batched.getConsecutiveMatches().size() == system.getConsecutiveMatches().size();
            for (size_t i = 0; same && i < batched.getLoopClosures().size(); ++i) {
                const auto& a = batched.getLoopClosures()[i];
                const auto& b = system.getLoopClosures()[i];
                same = a.current_frame_id == b.current_frame_id && a.matched_frame_id == b.matched_frame_id &&
                       a.num_matches == b.num_matches && a.similarity_score == b.similarity_score;
            }
            printf("BATCHED_EQUAL %d\n", same ? 1 : 0);
        }
        printf("FRAMES %zu LOOPS %zu\n", system.getFrames().size(), system.getLoopClosures().size());
        for (const auto& c : system.getLoopClosures())
            printf("LOOP %d %d %d %.17g\n", c.current_frame_id, c.matched_frame_id, c.num_matches, c.similarity_score);
        if (!system.getLoopClosures().empty()) {
            const auto& c = system.getLoopClosures().back();
            const auto m = system.matchFeatures(*system.findFrame(c.current_frame_id), *system.findFrame(c.matched_frame_id));
            printf("MATCHES %d %d %zu\n", c.current_frame_id, c.matched_frame_id, m.size());
            for (const auto& x : m) printf("M %d %d %d %g\n", x.queryIdx, x.trainIdx, x.imgIdx, x.distance);
        }
        {   // README.md:101: re-match features on identified loop frames — all of the busiest frame's closures, one launch
            int busiest = -1; size_t most = 0;
            for (const auto& c : system.getLoopClosures()) {
                size_t n = 0;
                for (const auto& d : system.getLoopClosures()) n += d.current_frame_id == c.current_frame_id;
                if (n > most) { most = n; busiest = c.current_frame_id; }
            }
            if (busiest >= 0) {
                const auto lists = system.matchLoopClosures(busiest);
                printf("RELISTS %d %zu", busiest, lists.size());
                for (const auto& l : lists) printf(" %zu", l.size());
                printf("\n");
            }
        }
        system.saveResults(out_dir);
        try {
            system.detectLoops(12345);                          // unknown id: must throw, like any std::exception user
            printf("ERROR no exception\n");
            return 2;
        } catch (const std::exception& e) {
            printf("EXPECTED_EXCEPTION %s\n", e.what());
        }
    } catch (const std::exception& e) {                         // the reference's main() catches std::exception too
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
