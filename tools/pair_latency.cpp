// pair_latency.cpp — wall clock of one matchFeatures call through the C ABI from C++ (no Python in the loop):
//   g++ -O2 -std=c++17 tools/pair_latency.cpp -Iinclude -Lslam-loop-closing_amd/lib -llcm_hip -Wl,-rpath,$PWD/slam-loop-closing_amd/lib -o tools/pair_latency_cpp
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "lcm.h"

int main() {
    lcm_handle* h = nullptr;
    if (lcm_create(nullptr, 0, nullptr, &h) != LCM_OK) { printf("lcm_create: %s\n", lcm_last_error()); return 1; }
    std::mt19937 rng(1);
    for (int mode = 0; mode < 4; ++mode) {
    lcm_set_tuning(h, LCM_TUNE_PAIR_UPLOAD_KERNEL, mode & 1);
    lcm_set_tuning(h, LCM_TUNE_PAIR_HOST_FOLD, (mode >> 1) & 1);
    printf("upload by %s, fold into %s\n", (mode & 1) ? "kernel" : "hipMemcpyAsync", (mode & 2) ? "pinned host memory" : "device memory + copy");
    const int shapes[][2] = {{2000, 2000}, {500, 500}, {2000, 20000}};
    for (auto& sh : shapes) {
        const int nq = sh[0], nt = sh[1];
        std::vector<uint8_t> q((size_t)nq * 32), t((size_t)nt * 32);
        for (auto& b : q) b = (uint8_t)rng();
        for (auto& b : t) b = (uint8_t)rng();
        std::vector<int32_t> idx((size_t)nq);
        std::vector<uint16_t> dist((size_t)nq);
        std::vector<lcm_dmatch> dm((size_t)nq);
        int n = 0, md = 0;
        for (int w = 0; w < 5; ++w) lcm_match_pair(h, q.data(), nq, t.data(), nt, idx.data(), dist.data(), &n);
        std::vector<double> a, b;
        for (int it = 0; it < 200; ++it) {
            auto t0 = std::chrono::steady_clock::now();
            lcm_match_pair(h, q.data(), nq, t.data(), nt, idx.data(), dist.data(), &n);
            auto t1 = std::chrono::steady_clock::now();
            lcm_match_features(h, q.data(), nq, t.data(), nt, dm.data(), &n, &md);
            auto t2 = std::chrono::steady_clock::now();
            a.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            b.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
        }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        lcm_launch_info li{};
        lcm_last_launch_info(h, &li);
        printf("%5d x %5d: lcm_match_pair median %.1f us (p10 %.1f), lcm_match_features median %.1f us; score kernel %.1f us on %u workgroups\n",
               nq, nt, a[a.size() / 2], a[a.size() / 10], b[b.size() / 2], li.kernel_ms * 1e3, li.workgroups);
    }
    }
    lcm_destroy(h);
    return 0;
}
