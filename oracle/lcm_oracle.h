/*
 * lcm_oracle.h — CPU restatement of the reference's ORB/Hamming loop-closure path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The product (slam-loop-closing_amd/) never calls it and has no
 * CPU fallback.
 *
 * PARITY UNPINNED.  The reference ships no implementation of this path (src/loop_closing.cpp does not
 * exist), no tests and no golden vectors; its arithmetic lives in OpenCV ("4.x", unpinned,
 * CMakeLists.txt:13), which is absent from this image.  This file restates the published behaviour of
 * cv::BFMatcher(NORM_HAMMING, crossCheck=false).match plus the README rules; the known-answer tests in
 * tests/test_oracle_kat.py are the pin, and tests/test_oracle_numpy.py cross-checks it against an
 * independent numpy restatement.  oracle/_ref is not buildable here (needs OpenCV).
 *
 * Each function cites the reference lines it follows (paths relative to the reference checkout).
 */
#ifndef LCM_ORACLE_H_
#define LCM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DESC_BYTES 32

typedef struct orc_params {   /* mirrors lcm_params; README.md:108-126 */
    int32_t ratio, dist_floor, min_matches, min_gap;
    double  sim_threshold;
    int32_t cross_check;      /* 0 = BFMatcher(NORM_HAMMING, crossCheck=false), the default; 1 / 2: orc_bf_match_cross */
    int32_t reserved;
} orc_params;

typedef struct orc_score { uint32_t good_count; uint16_t min_dist; uint16_t n_train; } orc_score;
typedef struct orc_dmatch { int32_t query_idx, train_idx, img_idx; float distance; } orc_dmatch;
typedef struct orc_candidate { int32_t current_frame_id, matched_frame_id, num_matches; double similarity_score; } orc_candidate;

void orc_params_default(orc_params* p);

/* popcount(a XOR b) over 32 bytes, one byte at a time (cv::norm NORM_HAMMING on CV_8U rows). */
int orc_hamming256(const uint8_t* a, const uint8_t* b);

/* BFMatcher(NORM_HAMMING, crossCheck=false).match(query, train): include/loop_closing.hpp:40,73.
 * Scans train rows ascending with a strict '<' update from INT_MAX => FIRST minimum wins.
 * Returns the number of matches (nq, or 0 when nq == 0 or nt == 0). */
int orc_bf_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist);

/* BFMatcher(NORM_HAMMING, crossCheck = true).match — the one matcher option visible in the tree (src/main.cpp:517
 * passes `false`).  OpenCV upstream, modules/core/src/batch_distance.cpp, `if (crosscheck)` block: every TRAIN row j
 * finds its nearest query row tidx[j] (first minimum); then, scanning j ascending, query i = tidx[j] takes j if
 * tdist[j] < dist[i] (strict, dist starts at INT_MAX).  Recent 4.x releases add `&& sidx[i] == j` (sidx = the forward
 * nearest train row of query i), which makes the result the MUTUAL nearest neighbours; older releases lack it, so a
 * query can keep a train row that chose it although the query itself prefers another.  The release that changed it
 * cannot be verified offline — behaviour is version dependent, hence PARITY UNPINNED and off by default:
 *     mode 1 = with the sidx test (mutual nearest), mode 2 = without it (legacy).
 * Unmatched queries get train_idx = -1 (knnMatchImpl drops them: compactResult).  Returns the number matched. */
int orc_bf_match_cross(const uint8_t* q, int nq, const uint8_t* t, int nt, int mode, int32_t* train_idx, int32_t* dist);

/* README.md:117 "2x minimum distance": keep d <= max(ratio*min_d, dist_floor).  keep may be NULL.
 * Returns the number kept; *min_dist = min over matches (-1 if n == 0). */
int orc_filter_good(const int32_t* dist, int n, int ratio, int dist_floor, uint8_t* keep, int* min_dist);

/* matchFeatures (include/loop_closing.hpp:40 + README.md:116-117): returns #good, query order kept. */
int orc_match_features(const uint8_t* q, int nq, const uint8_t* t, int nt, const orc_params* p,
                       orc_dmatch* out, int* min_dist);

/* The 8-byte record of one (query frame, stored frame) pair. */
void orc_pair_score(const uint8_t* q, int nq, const uint8_t* t, int nt, const orc_params* p, orc_score* out);

/* README.md:123-126: similarity = good / min(n1, n2) in double; loop iff sim > thr && good >= min_matches.
 * min(n1,n2) == 0 => no loop, similarity 0. */
int orc_loop_test(const orc_params* p, uint32_t good_count, int n_query_kp, int n_train_kp, double* similarity);

/* detectLoops (include/loop_closing.hpp:48, README.md:121-126) for frame index `cur` of a frame set laid out
 * as rows + i*stride_rows*32 with counts[i] rows and ids[i]; every stored i with ids[cur]-ids[i] >= min_gap.
 * Returns #candidates written (cap-limited). */
int orc_detect_loops(const uint8_t* rows, const int32_t* counts, const int32_t* ids, int n_frames, int stride_rows,
                     int cur, const orc_params* p, orc_candidate* out, int cap);

/* All-vs-all scores in (query ascending, stored ascending) order; only stored frames with
 * (index % shard_world) == shard_rank are scored (shard_world = 1: all).  offsets (n_frames+1) optional.
 * Returns #pairs.  scores may be NULL to count. */
size_t orc_all_vs_all(const uint8_t* rows, const int32_t* counts, const int32_t* ids, int n_frames, int stride_rows,
                      const orc_params* p, int shard_rank, int shard_world, orc_score* scores, size_t* offsets);

/* ---- CPU baseline (same results, tuned): 64-bit popcount / AVX-512 VPOPCNTDQ, pthreads over pairs ---- */
/* Scores `n_pairs` pairs given as (query frame index, train frame index) and returns the wall seconds.
 * isa_out (optional, >= 32 bytes) receives "avx512-vpopcntdq" or "popcnt64". */
double orc_fast_score_pairs(const uint8_t* rows, const int32_t* counts, int stride_rows,
                            const int32_t* pair_q, const int32_t* pair_t, size_t n_pairs,
                            const orc_params* p, int n_threads, orc_score* scores, char* isa_out);

/* Same, also returning per pair the sum mod 2^32 of the train indices (DMatch::trainIdx, first minimum) of the GOOD
 * matches — the checksum lcm_all_vs_all_argmin produces on the device.  idx_sums may be NULL. */
double orc_fast_score_pairs_idx(const uint8_t* rows, const int32_t* counts, int stride_rows,
                                const int32_t* pair_q, const int32_t* pair_t, size_t n_pairs,
                                const orc_params* p, int n_threads, orc_score* scores, uint32_t* idx_sums, char* isa_out);

#ifdef __cplusplus
}
#endif
#endif
