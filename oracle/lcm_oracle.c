/*
 * lcm_oracle.c — CPU restatement of the reference's ORB/Hamming loop-closure path.
 * TEST INFRASTRUCTURE ONLY / PARITY UNPINNED — see lcm_oracle.h for what that means and why.
 *
 * Part 1 (golden): byte-at-a-time, single-threaded, written for auditability, not speed.
 * Part 2 (baseline): the same results with 64-bit popcount or AVX-512 VPOPCNTDQ and pthreads over pairs;
 *                    it stands in for cv::BFMatcher's parallel_for_ + SIMD normHamming when bench.py
 *                    reports a CPU baseline.  tests/test_oracle_numpy.py and tests/test_golden.py check Part 2 == Part 1.
 */
#define _GNU_SOURCE
#include "lcm_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

void orc_params_default(orc_params* p) {
    p->ratio = 2;            /* README.md:117 */
    p->dist_floor = 0;       /* README states no floor */
    p->min_matches = 50;     /* README.md:124 */
    p->min_gap = 30;         /* README.md:109, include/loop_closing.hpp:31 */
    p->sim_threshold = 0.15; /* README.md:108 */
    p->cross_check = 0;      /* src/main.cpp:517: BFMatcher(norm, false) */
    p->reserved = 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* Part 1: golden restatement                                                                         */
/* ------------------------------------------------------------------------------------------------ */

static int popcount8(uint8_t v) {
    int c = 0;
    while (v) { c += v & 1; v >>= 1; }
    return c;
}

/* NORM_HAMMING between two CV_8U rows of 32 columns: number of differing bits. */
int orc_hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < ORC_DESC_BYTES; ++i) d += popcount8((uint8_t)(a[i] ^ b[i]));
    return d;
}

/* BFMatcher::match == knnMatch(k=1), no mask, crossCheck=false (include/loop_closing.hpp:73 `matcher_`;
 * calling convention of the tree's one executed matcher call: src/main.cpp:517-519, query first).
 * OpenCV's batchDistance keeps `dist = INT_MAX, idx = -1` per query row and, scanning train rows in
 * ascending order, updates on `d < dist` (strict) — so the lowest train index among equal minima wins.
 * The two-Mat match() overload compacts empty results: no train rows => no matches at all. */
int orc_bf_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist) {
    if (nq <= 0 || nt <= 0) return 0;
    for (int i = 0; i < nq; ++i) {
        int best = INT_MAX, best_j = -1;
        for (int j = 0; j < nt; ++j) {
            int d = orc_hamming256(q + (size_t)i * ORC_DESC_BYTES, t + (size_t)j * ORC_DESC_BYTES);
            if (d < best) { best = d; best_j = j; }
        }
        train_idx[i] = best_j;
        dist[i] = best;
    }
    return nq;
}

int orc_bf_match_cross(const uint8_t* q, int nq, const uint8_t* t, int nt, int mode, int32_t* train_idx, int32_t* dist) {
    for (int i = 0; i < nq; ++i) { train_idx[i] = -1; dist[i] = INT_MAX; }
    if (nq <= 0 || nt <= 0) return 0;
    int32_t* tidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nt);
    int32_t* tdist = (int32_t*)malloc(sizeof(int32_t) * (size_t)nt);
    int32_t* sidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
    int32_t* sdist = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
    orc_bf_match(t, nt, q, nq, tidx, tdist);      /* batchDistance(src2, src1): nearest QUERY of every train row */
    orc_bf_match(q, nq, t, nt, sidx, sdist);      /* batchDistance(src1, src2): nearest TRAIN of every query row */
    for (int j = 0; j < nt; ++j) {
        const int i = tidx[j];
        if (tdist[j] < dist[i] && (mode != 1 || sidx[i] == j)) { dist[i] = tdist[j]; train_idx[i] = j; }
    }
    int n = 0;
    for (int i = 0; i < nq; ++i) n += train_idx[i] >= 0;
    free(tidx); free(tdist); free(sidx); free(sdist);
    return n;
}

/* stage 1 of matchFeatures under the handle's cross_check setting: idx[i] = -1 marks "no match for query i" */
static void bf_match_any(const uint8_t* q, int nq, const uint8_t* t, int nt, int cross_check, int32_t* idx, int32_t* d) {
    if (cross_check) orc_bf_match_cross(q, nq, t, nt, cross_check, idx, d);
    else orc_bf_match(q, nq, t, nt, idx, d);
}

/* README.md:117: "Distance-based filtering (threshold: 2x minimum distance)".  Inclusive compare (the OpenCV
 * tutorial convention this rule comes from; with '<' and min_d == 0 nothing could ever survive). */
int orc_filter_good(const int32_t* dist, int n, int ratio, int dist_floor, uint8_t* keep, int* min_dist) {
    if (n <= 0) { if (min_dist) *min_dist = -1; return 0; }
    int m = INT_MAX;
    for (int i = 0; i < n; ++i) if (dist[i] < m) m = dist[i];
    int thr = ratio * m;
    if (dist_floor > thr) thr = dist_floor;
    int good = 0;
    for (int i = 0; i < n; ++i) {
        int k = dist[i] <= thr;
        if (keep) keep[i] = (uint8_t)k;
        good += k;
    }
    if (min_dist) *min_dist = m;
    return good;
}

/* LoopClosingSystem::matchFeatures(frame1, frame2) -> std::vector<cv::DMatch> (include/loop_closing.hpp:40):
 * DMatch{queryIdx, trainIdx, imgIdx = 0, distance = (float)d}, ascending queryIdx, filtered. */
int orc_match_features(const uint8_t* q, int nq, const uint8_t* t, int nt, const orc_params* p,
                       orc_dmatch* out, int* min_dist) {
    int n = (nq > 0 && nt > 0) ? nq : 0;
    if (n == 0) { if (min_dist) *min_dist = -1; return 0; }
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t* d = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t* keep = (uint8_t*)malloc((size_t)n);
    bf_match_any(q, nq, t, nt, p->cross_check, idx, d);
    /* the filter sees only the matches that exist (cross-check leaves queries unmatched): compact, filter, expand */
    int32_t* dm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t* qi = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int nm = 0;
    for (int i = 0; i < n; ++i) if (idx[i] >= 0) { dm[nm] = d[i]; qi[nm] = i; ++nm; }
    int m;
    orc_filter_good(dm, nm, p->ratio, p->dist_floor, keep, &m);
    int k = 0;
    for (int a = 0; a < nm; ++a) {
        if (!keep[a]) continue;
        out[k].query_idx = qi[a];
        out[k].train_idx = idx[qi[a]];
        out[k].img_idx = 0;
        out[k].distance = (float)dm[a];
        ++k;
    }
    if (min_dist) *min_dist = m;
    free(idx); free(d); free(keep); free(dm); free(qi);
    return k;
}

void orc_pair_score(const uint8_t* q, int nq, const uint8_t* t, int nt, const orc_params* p, orc_score* out) {
    out->n_train = (uint16_t)nt;
    if (nq <= 0 || nt <= 0) { out->good_count = 0; out->min_dist = 0xFFFF; return; }
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
    int32_t* d = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
    bf_match_any(q, nq, t, nt, p->cross_check, idx, d);
    int nm = 0;
    for (int i = 0; i < nq; ++i) if (idx[i] >= 0) d[nm++] = d[i];
    int m;
    out->good_count = (uint32_t)orc_filter_good(d, nm, p->ratio, p->dist_floor, NULL, &m);
    out->min_dist = nm > 0 ? (uint16_t)m : (uint16_t)0xFFFF;      /* no match survived the cross-check: an empty pair */
    free(idx); free(d);
}

/* README.md:123-126.  "exceeds" => strict '>'; "at least 50" => '>='; denominator = min(features1, features2). */
int orc_loop_test(const orc_params* p, uint32_t good_count, int n_query_kp, int n_train_kp, double* similarity) {
    int den = n_query_kp < n_train_kp ? n_query_kp : n_train_kp;
    double sim = 0.0;
    if (den > 0) sim = (double)good_count / (double)den;
    if (similarity) *similarity = sim;
    return den > 0 && sim > p->sim_threshold && (int64_t)good_count >= (int64_t)p->min_matches;
}

/* detectLoops(current_frame_id): README.md:122 "compared against all frames at least min_loop_gap frames ago";
 * the tree's executed analogue uses `past <= curr - loopGap` (src/main.cpp:1375,1379) => inclusive gap.
 * query = current frame, train = stored frame. */
int orc_detect_loops(const uint8_t* rows, const int32_t* counts, const int32_t* ids, int n_frames, int stride_rows,
                     int cur, const orc_params* p, orc_candidate* out, int cap) {
    int k = 0;
    const uint8_t* q = rows + (size_t)cur * stride_rows * ORC_DESC_BYTES;
    for (int i = 0; i < n_frames; ++i) {
        if (i == cur) continue;
        if (ids[cur] - ids[i] < p->min_gap) continue;
        orc_score s;
        orc_pair_score(q, counts[cur], rows + (size_t)i * stride_rows * ORC_DESC_BYTES, counts[i], p, &s);
        double sim;
        if (orc_loop_test(p, s.good_count, counts[cur], counts[i], &sim)) {
            if (k < cap) {
                out[k].current_frame_id = ids[cur];
                out[k].matched_frame_id = ids[i];
                out[k].num_matches = (int32_t)s.good_count;
                out[k].similarity_score = sim;
            }
            ++k;
        }
    }
    return k < cap ? k : cap;
}

size_t orc_all_vs_all(const uint8_t* rows, const int32_t* counts, const int32_t* ids, int n_frames, int stride_rows,
                      const orc_params* p, int shard_rank, int shard_world, orc_score* scores, size_t* offsets) {
    size_t k = 0;
    for (int c = 0; c < n_frames; ++c) {
        if (offsets) offsets[c] = k;
        for (int i = 0; i < n_frames; ++i) {
            if (shard_world > 1 && (i % shard_world) != shard_rank) continue;
            if (i == c || ids[c] - ids[i] < p->min_gap) continue;
            if (scores)
                orc_pair_score(rows + (size_t)c * stride_rows * ORC_DESC_BYTES, counts[c],
                               rows + (size_t)i * stride_rows * ORC_DESC_BYTES, counts[i], p, &scores[k]);
            ++k;
        }
    }
    if (offsets) offsets[n_frames] = k;
    return k;
}

/* ------------------------------------------------------------------------------------------------ */
/* Part 2: tuned CPU baseline (identical results)                                                     */
/* ------------------------------------------------------------------------------------------------ */

static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* best key per query = (dist << 32) | train index; min over keys == first minimum. */
static void best_keys_popcnt64(const uint8_t* q, int nq, const uint8_t* t, int nt, uint64_t* keys) {
    for (int i = 0; i < nq; ++i) {
        const uint8_t* qi = q + (size_t)i * 32;
        uint64_t q0 = ld64(qi), q1 = ld64(qi + 8), q2 = ld64(qi + 16), q3 = ld64(qi + 24);
        uint64_t best = ~0ull;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* tj = t + (size_t)j * 32;
            uint64_t d = (uint64_t)__builtin_popcountll(q0 ^ ld64(tj)) + (uint64_t)__builtin_popcountll(q1 ^ ld64(tj + 8)) +
                         (uint64_t)__builtin_popcountll(q2 ^ ld64(tj + 16)) + (uint64_t)__builtin_popcountll(q3 ^ ld64(tj + 24));
            uint64_t key = (d << 32) | (uint32_t)j;
            if (key < best) best = key;
        }
        keys[i] = best;
    }
}

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vpopcntdq")))
static void best_keys_avx512(const uint8_t* q, int nq, const uint8_t* t, int nt, uint64_t* keys) {
    /* lane order of the 8 distances produced per step: rows {0,2,1,3,4,6,5,7} of the 8-row group */
    const __m512i lane_idx = _mm512_set_epi64(7, 5, 6, 4, 3, 1, 2, 0);
    const int nt8 = nt & ~7;
    for (int i = 0; i < nq; ++i) {
        const uint8_t* qi = q + (size_t)i * 32;
        const __m512i Q = _mm512_broadcast_i64x4(_mm256_loadu_si256((const __m256i*)qi));
        __m512i best = _mm512_set1_epi64(-1);
        for (int j = 0; j < nt8; j += 8) {
            const uint8_t* tj = t + (size_t)j * 32;
            __m512i A = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(tj), Q));
            __m512i B = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(tj + 64), Q));
            __m512i C = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(tj + 128), Q));
            __m512i D = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512(tj + 192), Q));
            __m512i s1 = _mm512_add_epi64(_mm512_unpacklo_epi64(A, B), _mm512_unpackhi_epi64(A, B));
            __m512i s2 = _mm512_add_epi64(_mm512_unpacklo_epi64(C, D), _mm512_unpackhi_epi64(C, D));
            __m512i r = _mm512_add_epi64(_mm512_shuffle_i64x2(s1, s2, 0x88), _mm512_shuffle_i64x2(s1, s2, 0xDD));
            __m512i key = _mm512_or_si512(_mm512_slli_epi64(r, 32), _mm512_add_epi64(lane_idx, _mm512_set1_epi64(j)));
            best = _mm512_min_epu64(best, key);
        }
        uint64_t b = _mm512_reduce_min_epu64(best);
        if (nt8 < nt) {
            uint64_t q0 = ld64(qi), q1 = ld64(qi + 8), q2 = ld64(qi + 16), q3 = ld64(qi + 24);
            for (int j = nt8; j < nt; ++j) {
                const uint8_t* tj = t + (size_t)j * 32;
                uint64_t d = (uint64_t)__builtin_popcountll(q0 ^ ld64(tj)) + (uint64_t)__builtin_popcountll(q1 ^ ld64(tj + 8)) +
                             (uint64_t)__builtin_popcountll(q2 ^ ld64(tj + 16)) + (uint64_t)__builtin_popcountll(q3 ^ ld64(tj + 24));
                uint64_t key = (d << 32) | (uint32_t)j;
                if (key < b) b = key;
            }
        }
        keys[i] = b;
    }
}
static int have_avx512_vpopcnt(void) {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512vpopcntdq");
}
#else
static int have_avx512_vpopcnt(void) { return 0; }
#endif

typedef struct fast_job {
    const uint8_t* rows; const int32_t* counts; int stride_rows;
    const int32_t* pair_q; const int32_t* pair_t; size_t n_pairs;
    const orc_params* p; orc_score* scores;
    int tid, n_threads, use_avx512, max_rows;
    uint32_t* idx_sums;   /* optional: per pair, sum of the GOOD matches' train indices mod 2^32 */
} fast_job;

static void* fast_worker(void* arg) {
    fast_job* jb = (fast_job*)arg;
    const size_t kcap = (size_t)(jb->max_rows > 0 ? jb->max_rows : 1);
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * kcap);
    uint64_t* bkeys = jb->p->cross_check ? (uint64_t*)malloc(sizeof(uint64_t) * kcap) : NULL;    /* train -> query */
    uint64_t* ckeys = jb->p->cross_check ? (uint64_t*)malloc(sizeof(uint64_t) * kcap) : NULL;    /* cross-checked */
    for (size_t k = (size_t)jb->tid; k < jb->n_pairs; k += (size_t)jb->n_threads) {
        int qi = jb->pair_q[k], ti = jb->pair_t[k];
        int nq = jb->counts[qi], nt = jb->counts[ti];
        orc_score* s = &jb->scores[k];
        s->n_train = (uint16_t)nt;
        if (jb->idx_sums) jb->idx_sums[k] = 0;
        if (nq <= 0 || nt <= 0) { s->good_count = 0; s->min_dist = 0xFFFF; continue; }
        const uint8_t* q = jb->rows + (size_t)qi * jb->stride_rows * 32;
        const uint8_t* t = jb->rows + (size_t)ti * jb->stride_rows * 32;
#if defined(__x86_64__)
        if (jb->use_avx512) best_keys_avx512(q, nq, t, nt, keys); else
#endif
        best_keys_popcnt64(q, nq, t, nt, keys);
        const uint64_t* use = keys;
        if (jb->p->cross_check) {                   /* orc_bf_match_cross on (dist, idx) keys; ~0 = unmatched */
#if defined(__x86_64__)
            if (jb->use_avx512) best_keys_avx512(t, nt, q, nq, bkeys); else
#endif
            best_keys_popcnt64(t, nt, q, nq, bkeys);
            for (int i = 0; i < nq; ++i) ckeys[i] = ~0ull;
            for (int j = 0; j < nt; ++j) {
                const uint32_t i = (uint32_t)bkeys[j];
                const uint64_t cand = (bkeys[j] & 0xFFFFFFFF00000000ull) | (uint32_t)j;
                if ((cand >> 32) < (ckeys[i] >> 32) && (jb->p->cross_check != 1 || (uint32_t)keys[i] == (uint32_t)j)) ckeys[i] = cand;
            }
            use = ckeys;
        }
        uint32_t m = 0xFFFFFFFFu;
        for (int i = 0; i < nq; ++i) { uint32_t d = (uint32_t)(use[i] >> 32); if (d < m) m = d; }
        if (m == 0xFFFFFFFFu) { s->good_count = 0; s->min_dist = 0xFFFF; continue; }      /* nothing survived */
        uint32_t thr = (uint32_t)jb->p->ratio * m;
        if ((uint32_t)jb->p->dist_floor > thr) thr = (uint32_t)jb->p->dist_floor;
        uint32_t good = 0, isum = 0;
        for (int i = 0; i < nq; ++i) {
            const int ok = use[i] != ~0ull && ((uint32_t)(use[i] >> 32) <= thr);
            good += (uint32_t)ok;
            if (ok) isum += (uint32_t)use[i];       /* low word of the key = trainIdx */
        }
        s->good_count = good;
        s->min_dist = (uint16_t)m;
        if (jb->idx_sums) jb->idx_sums[k] = isum;
    }
    free(keys); free(bkeys); free(ckeys);
    return NULL;
}

double orc_fast_score_pairs_idx(const uint8_t* rows, const int32_t* counts, int stride_rows, const int32_t* pair_q, const int32_t* pair_t, size_t n_pairs, const orc_params* p, int n_threads, orc_score* scores, uint32_t* idx_sums, char* isa_out);
double orc_fast_score_pairs(const uint8_t* rows, const int32_t* counts, int stride_rows,
                            const int32_t* pair_q, const int32_t* pair_t, size_t n_pairs,
                            const orc_params* p, int n_threads, orc_score* scores, char* isa_out) {
    return orc_fast_score_pairs_idx(rows, counts, stride_rows, pair_q, pair_t, n_pairs, p, n_threads, scores, NULL, isa_out);
}

double orc_fast_score_pairs_idx(const uint8_t* rows, const int32_t* counts, int stride_rows,
                                const int32_t* pair_q, const int32_t* pair_t, size_t n_pairs,
                                const orc_params* p, int n_threads, orc_score* scores, uint32_t* idx_sums, char* isa_out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    int use512 = have_avx512_vpopcnt();
    if (isa_out) strcpy(isa_out, use512 ? "avx512-vpopcntdq" : "popcnt64");
    pthread_t th[256];
    fast_job jobs[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < n_threads; ++i) {
        fast_job jb = { rows, counts, stride_rows, pair_q, pair_t, n_pairs, p, scores, i, n_threads, use512, stride_rows, idx_sums };
        jobs[i] = jb;
        if (i > 0) pthread_create(&th[i], NULL, fast_worker, &jobs[i]);
    }
    fast_worker(&jobs[0]);
    for (int i = 1; i < n_threads; ++i) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
