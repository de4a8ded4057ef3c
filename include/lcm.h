/*
 * lcm.h — C ABI of the MI355X loop-closure descriptor matcher ("lcm").
 *
 * This is the drop-in boundary for the ORB / Hamming hot path that the reference project
 * F-Fer/SLAM-Loop-Closing declares in include/loop_closing.hpp but delegates to
 * cv::BFMatcher(NORM_HAMMING).  Every entry point cites the reference interface it replaces
 * (paths are relative to the reference checkout):
 *
 *   lcm_match_pair        <- the `matcher_->match(desc1, desc2, matches)` call that
 *                            LoopClosingSystem::matchFeatures makes (include/loop_closing.hpp:40,73)
 *   lcm_match_features    <- LoopClosingSystem::matchFeatures incl. the "2 x minimum distance"
 *                            filter (include/loop_closing.hpp:40, README.md:116-117)
 *   lcm_match_stored      <- matchFeatures on two frames of frames_ (README.md:101 re-match of loop frames)
 *   lcm_db_append*        <- `frames_.push_back(frame)` inside processFrame
 *                            (include/loop_closing.hpp:34,69) — the stored-frame descriptor database
 *   lcm_query_scores      <- the per-stored-frame loop inside detectLoops
 *                            (include/loop_closing.hpp:48, README.md:121-126)
 *   lcm_detect_loops      <- LoopClosingSystem::detectLoops (include/loop_closing.hpp:48)
 *   lcm_all_vs_all        <- the O(N^2) "every frame against every frame >= gap ago" search, whose only
 *                            executed analogue in the tree is src/main.cpp:1375-1388
 *   lcm_loop_candidate    <- struct LoopCandidate (include/loop_closing.hpp:22-27), same field order
 *   lcm_dmatch            <- cv::DMatch as consumed at src/main.cpp:551-555 (queryIdx, trainIdx, imgIdx, distance)
 *
 * Conventions
 *   - plain C: pointers + sizes only, no C++/torch/OpenCV types; caller owns every host buffer;
 *     the library owns device memory behind the opaque lcm_handle.
 *   - a descriptor is 32 bytes (256 bits); descriptor matrices are row-major, n rows x 32 bytes,
 *     contiguous — exactly `cv::Mat(CV_8UC1, rows=n, cols=32).ptr<uint8_t>()`.
 *   - every function returns an lcm_status (0 = OK, negative = error); nothing throws across the
 *     boundary; lcm_last_error() gives the message of the calling thread's last failure.
 *   - there is NO CPU fallback: without a usable HIP device lcm_create fails with
 *     LCM_ERR_NO_DEVICE.  The CPU restatement under oracle/ is test infrastructure only.
 *   - a handle is thread-compatible (one caller at a time).  A call leaves the calling thread's current HIP device set to
 *     the handle's device (lcm_group_* calls: to one of the group's devices); callers that run their own HIP work on
 *     another device set it again afterwards.
 *   - sizes: a frame holds up to 65535 descriptor rows, as stored ("train") frame and as query frame alike (ORB's
 *     nfeatures is the caller's choice; the reference uses 2000).  Query frames above 2048 rows take the packed bulk
 *     route (see lcm_route); with cross_check != 0, or with LCM_TUNE_PACKED = 0, query frames are limited to 2048
 *     rows and a larger one is refused with LCM_ERR_CAPACITY.
 */
#ifndef LCM_H_
#define LCM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define LCM_API __attribute__((visibility("default")))
#else
#define LCM_API
#endif

#define LCM_DESC_BYTES 32          /* 256-bit ORB descriptor (README.md:115) */
#define LCM_DESC_WORDS 8
#define LCM_MAX_TRAIN_ROWS (1 << 22) /* packed (dist,idx) key: 9 + 22 bits */

typedef enum lcm_status {
    LCM_OK = 0,
    LCM_ERR_INVALID_ARG = -1,
    LCM_ERR_NO_DEVICE = -2,     /* no HIP device / runtime: the product path never falls back to CPU */
    LCM_ERR_HIP = -3,           /* a HIP runtime call failed (message has hipGetErrorString) */
    LCM_ERR_CAPACITY = -4,      /* output buffer or reserved database too small */
    LCM_ERR_ORDER = -5,         /* frame ids must be appended in strictly increasing order */
    LCM_ERR_NOT_FOUND = -6,     /* frame id not in this handle's database */
    LCM_ERR_OOM = -7
} lcm_status;

/* Parameters the reference leaves to README prose (README.md:108-126).  Defaults (lcm_params_default):
 * ratio=2, dist_floor=0, min_matches=50, sim_threshold=0.15, min_gap=30, cross_check=0. */
typedef struct lcm_params {
    int32_t ratio;          /* good match: d <= max(ratio*min_d, dist_floor)      README.md:117 */
    int32_t dist_floor;     /* README states no floor -> 0 */
    int32_t min_matches;    /* loop needs good_count >= min_matches               README.md:124 */
    int32_t min_gap;        /* compare only frames with cur_id - id >= min_gap    README.md:122, hpp:31.
                             * The gap is taken on frame IDS, not on storage positions: pass the processed-frame
                             * counter as id (as processFrame's callers do), or scale min_gap, when ids are sparse
                             * (e.g. raw video frame numbers with frame_skip = 3).  The tree's only executed
                             * analogue, src/main.cpp:1374-1379, is positional over keyframe indices; the two readings
                             * coincide for dense ids.  Part of what "parity unpinned" covers. */
    double  sim_threshold;  /* loop needs similarity > sim_threshold (strict)     README.md:123 */
    int32_t cross_check;    /* BFMatcher crossCheck (src/main.cpp:517 passes false): 0 = off (default);
                             * 1 = mutual nearest neighbours (recent OpenCV 4.x: the `sidx` test in batchDistance);
                             * 2 = legacy OpenCV (a query keeps the best train row among those that chose it).
                             * Version dependent upstream, hence off by default; see oracle/lcm_oracle.h. */
    int32_t reserved;       /* 0 */
} lcm_params;

/* One record per (query frame, stored frame) pair: what the device ships back (8 bytes).
 * similarity = good_count / min(n_query, n_train) is formed on the host in IEEE double. */
typedef struct lcm_score {
    uint32_t good_count;    /* matches surviving the ratio*min_d filter */
    uint16_t min_dist;      /* min over queries of the best distance; 0xFFFF if the pair is empty */
    uint16_t n_train;       /* descriptor rows of the stored frame */
} lcm_score;

/* Same layout as cv::DMatch (int queryIdx, trainIdx, imgIdx; float distance) — 16 bytes. */
typedef struct lcm_dmatch {
    int32_t query_idx;
    int32_t train_idx;
    int32_t img_idx;        /* always 0: single train image */
    float   distance;       /* integer-valued 0..256, exact in float */
} lcm_dmatch;

/* Same field order/types as loop_closing::LoopCandidate (include/loop_closing.hpp:22-27) — 24 bytes. */
typedef struct lcm_loop_candidate {
    int32_t current_frame_id;
    int32_t matched_frame_id;
    int32_t num_matches;
    double  similarity_score;
} lcm_loop_candidate;

/* Timing of the most recent bulk / pair-mode launch on this handle, from hipEvents on the launch stream.  Meant for
 * the bulk calls: online queries record into the same events from their own streams, so while online tickets are
 * outstanding (or a processFrame-style pair match runs beside one) the structure describes whichever launch recorded
 * last and is not attributable to a call — read lcm_online_stats_read for online work. */
typedef struct lcm_launch_info {
    double   kernel_ms;         /* device time of the pair-match kernel(s) of the last bulk call */
    uint64_t pairs;             /* (query frame, stored frame) pairs scored */
    uint64_t distances;         /* sum over pairs of n_query * n_train */
    uint64_t algo_bytes;        /* sum n_train*32 per pair + n_query*32 per query frame + 8 per pair */
    uint32_t launches;          /* kernel launches that made up the call */
    uint32_t workgroups;        /* workgroups of the (largest) launch */
    double   aux_kernel_ms;     /* device time of the call's follow-up kernels, 0 if none: the loop-test kernels
                                 * (k_loop_count, k_block_scan, k_loop_emit and the count read-back between them) of
                                 * lcm_all_vs_all_loops; the fold kernels of a packed bulk search (summed over its chunks:
                                 * kernel_ms - aux_kernel_ms is then the score kernels' time) */
    uint32_t route;             /* which kernels served the call: lcm_route */
    uint32_t launches_in_flight; /* packed bulk search of several chunks: 2 (consecutive chunks run on two streams, so that one
                                 * chunk's workgroups fill the chip while the other's launch drains); else 1 */
    double   score_ms_sum;      /* packed bulk search: SUM of the score kernels' own durations, each from events around it on
                                 * its stream (what a kernel trace reports per launch).  With 2 launches in flight they
                                 * overlap pairwise: score_ms_sum ~ launches_in_flight x (kernel_ms - exposed folds) */
    uint32_t score_launches;    /* packed bulk search: number of score launches (= chunks) */
    uint32_t reserved_;
} lcm_launch_info;
/* LCM_ROUTE_PLAIN: one workgroup per (query frame, run of stored frames), records formed in the kernel;
 * LCM_ROUTE_PACKED: query rows of consecutive frames packed into full 2048-row workgroups + fold kernel (bulk);
 * LCM_ROUTE_SPLIT: query frame cut into row chunks + fold kernel (online, short databases);
 * LCM_ROUTE_MATRIX: opt-in matrix-core variants 4 / 5; LCM_ROUTE_CROSS: cross_check (two passes + combine). */
typedef enum lcm_route { LCM_ROUTE_PLAIN = 0, LCM_ROUTE_PACKED = 1, LCM_ROUTE_SPLIT = 2, LCM_ROUTE_MATRIX = 3, LCM_ROUTE_CROSS = 4 } lcm_route;

/* Totals over the online queries COLLECTED so far on a handle (single and batched): device time of their kernels from
 * HIP events on the launch stream, and the work they did — what a streaming run's roofline is computed from. */
typedef struct lcm_online_stats {
    double   kernel_ms;
    uint64_t launches, queries, pairs, distances, algo_bytes;
} lcm_online_stats;

typedef struct lcm_handle lcm_handle;

LCM_API void        lcm_params_default(lcm_params* p);
LCM_API const char* lcm_last_error(void);
LCM_API const char* lcm_backend_name(void);         /* "hip-gfx950" */
LCM_API int         lcm_device_count(void);         /* #HIP devices, 0 if none / no runtime */

/* Create a matcher bound to HIP device `device_id`.  `stream` is a hipStream_t (as void*) that the bulk, pair-mode
 * and database work is enqueued on, or NULL for a library-owned stream.  Online queries (lcm_query_submit*,
 * lcm_detect_loops) run on streams the library owns, one per query slot, each ordered AFTER everything `stream` and the
 * append path already hold at submit time; their results reach the caller through the collect calls only, so nothing
 * on `stream` has to wait for them (LCM_TUNE_ONLINE_STREAMS = 0 puts them back on `stream`). */
LCM_API int  lcm_create(const lcm_params* params, int device_id, void* stream, lcm_handle** out);
LCM_API void lcm_destroy(lcm_handle* h);
LCM_API int  lcm_set_params(lcm_handle* h, const lcm_params* params);
LCM_API int  lcm_get_params(const lcm_handle* h, lcm_params* params);
LCM_API int  lcm_sync(lcm_handle* h);

/* ---- stored-frame descriptor database (device resident) -------------------------------------------- */
/* Reserve room for `n_frames` frames of up to `max_desc` rows each.  Growing later is allowed (copying). */
LCM_API int  lcm_db_reserve(lcm_handle* h, int n_frames, int max_desc);
/* Append one frame (host rows).  Copies through pinned staging with hipMemcpyAsync on a copy stream that
 * overlaps matching; the caller may reuse `desc` as soon as the call returns.  `n_keypoints` is the
 * similarity denominator (Frame::keypoints.size(), == n for ORB); pass -1 to use `n`. */
LCM_API int  lcm_db_append(lcm_handle* h, int frame_id, const uint8_t* desc, int n, int n_keypoints);
/* Same, rows already in device memory (n x 32 bytes, contiguous). */
LCM_API int  lcm_db_append_device(lcm_handle* h, int frame_id, const void* d_desc, int n, int n_keypoints);
LCM_API int  lcm_db_size(const lcm_handle* h);                 /* frames stored */
LCM_API int  lcm_db_clear(lcm_handle* h);
/* Keep the first n_frames stored frames, drop the rest (no-op if fewer are stored).  Waits for everything in flight;
 * tickets submitted before the call are void afterwards, as after lcm_db_clear. */
LCM_API int  lcm_db_truncate(lcm_handle* h, int n_frames);
LCM_API int  lcm_db_frame_info(const lcm_handle* h, int slot, int* frame_id, int* n_desc, int* n_keypoints);
/* Copy a stored frame's rows back to the host (tests / snapshot). */
LCM_API int  lcm_db_read(lcm_handle* h, int slot, uint8_t* desc_out, int cap_rows);

/* Snapshot / restore of the stored-frame database (ids, row counts, keypoint counts, rows) as one raw file — the
 * reference has no checkpointing; this makes long streaming runs resumable. */
LCM_API int  lcm_db_save(lcm_handle* h, const char* path);
LCM_API int  lcm_db_load(lcm_handle* h, const char* path);

/* ---- pair mode: BFMatcher(NORM_HAMMING, crossCheck=false).match ------------------------------------ */
/* For each query row q (ascending): train_idx[q] = FIRST index of the minimum Hamming distance over the
 * nt train rows, dist[q] = that distance.  nq == 0 or nt == 0: nothing is written, *n_matches = 0. */
LCM_API int  lcm_match_pair(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                            int32_t* train_idx, uint16_t* dist, int* n_matches);
/* matchFeatures: the above + keep matches with d <= max(ratio*min_d, dist_floor), query order preserved.
 * `out` needs room for nq records. */
LCM_API int  lcm_match_features(lcm_handle* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                                lcm_dmatch* out, int* n_out, int* min_dist);

/* matchFeatures between two STORED frames (device-resident rows, no upload): the "re-match features on identified
 * loop frames" step (README.md:101) that turns a LoopCandidate into its DMatch list.  `out` needs room for the query
 * frame's row count (cap). */
LCM_API int  lcm_match_stored(lcm_handle* h, int query_frame_id, int train_frame_id,
                              lcm_dmatch* out, int cap, int* n_out, int* min_dist);

/* The same for MANY pairs in ONE launch (the match lists of all loop candidates of a frame): every pair is cut into
 * (query chunk x train segment) work items of one kernel launch, a second kernel folds the segments, one download.
 * out (host) receives the DMatch lists back to back; offsets[n_pairs + 1] (host, required) their bounds; min_dists
 * (host, optional) each pair's minimum distance, -1 if it has no matches.  A pair with an empty side has no matches. */
typedef struct lcm_pair_ref { int32_t query_frame_id, train_frame_id; } lcm_pair_ref;
LCM_API int  lcm_match_stored_batch(lcm_handle* h, const lcm_pair_ref* pairs, int n_pairs,
                                    lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists);
/* ... and one query frame given by the host (the current frame, not stored yet) against n_trains stored frames. */
LCM_API int  lcm_match_query_batch(lcm_handle* h, const uint8_t* query, int nq, const int32_t* train_frame_ids, int n_trains,
                                   lcm_dmatch* out, size_t cap, size_t* offsets, int32_t* min_dists);

/* ---- loop search against the stored database --------------------------------------------------------- */
/* Score `query` (id query_frame_id) against every stored frame with query_frame_id - id >= min_gap, ascending
 * slot order.  out_scores / out_frame_ids need room for lcm_db_size() records. */
LCM_API int  lcm_query_scores(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id,
                              lcm_score* out_scores, int32_t* out_frame_ids, int* n_out);
/* The same query, asynchronously (streaming mode): lcm_query_submit copies `query` into pinned staging, enqueues the
 * upload, the kernel(s) and the download of the records, and returns a ticket without waiting; up to 4 queries may be
 * in flight.  lcm_query_collect waits for that ticket and copies the records out (cap = room in out_scores /
 * out_frame_ids).  Frames appended after the submit are not part of its answer. */
LCM_API int  lcm_query_submit(lcm_handle* h, const uint8_t* query, int nq, int query_frame_id, int* ticket);
LCM_API int  lcm_query_collect(lcm_handle* h, int ticket, lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out);
/* Micro-batched online queries (streaming mode at full device rate): up to 16 frames — e.g. the last few camera frames,
 * none of them appended yet — are scored by ONE launch, each against the stored frames with its_id - id >= min_gap as the
 * database stands at submit time.  (With min_gap >= the id span of the batch this equals submitting them one by one,
 * each before its predecessors are appended.)  One ticket for the batch; lcm_query_collect_batch returns the records of
 * all queries back to back (query 0's first) and, optionally, offsets[n_queries + 1]. */
LCM_API int  lcm_query_submit_batch(lcm_handle* h, const uint8_t* const* queries, const int* nq, const int* query_frame_ids,
                                    int n_queries, int* ticket);
LCM_API int  lcm_query_collect_batch(lcm_handle* h, int ticket, lcm_score* out_scores, size_t cap, size_t* n_out,
                                     size_t* offsets);
LCM_API int  lcm_online_stats_read(lcm_handle* h, lcm_online_stats* out, int reset);
/* detectLoops for a frame that is already stored (or given explicitly with query != NULL). */
LCM_API int  lcm_detect_loops(lcm_handle* h, int current_frame_id, const uint8_t* query, int nq, int n_keypoints,
                              lcm_loop_candidate* out, int cap, int* n_out);
/* Host-side loop test on shipped integers: similarity in IEEE double, strict '>' on the threshold
 * (README.md:123-126).  Returns 1 if the pair is a loop candidate. */
LCM_API int  lcm_loop_test(const lcm_params* p, const lcm_score* s, int n_query_kp, int n_train_kp,
                           double* similarity);

/* ---- bulk all-vs-all (benchmark / re-scan mode) ---------------------------------------------------- */
/* Query set: `n_q_frames` frames at d_query_rows + i*q_stride_rows*32 (device memory), row counts
 * d_query_counts[i] (device, int32), ids q_ids[i] (host, strictly increasing).  If d_query_rows is NULL the
 * handle's own database is the query set.  Every query frame is scored against every stored frame of this
 * handle with q_id - id >= min_gap.  Scores are written to the device buffer d_scores (lcm_score records) in
 * (query ascending, stored slot ascending) order; *n_pairs receives the count; pair_offsets (host, optional,
 * n_q_frames+1 entries) receives the start of each query frame's run.  d_scores may be NULL to size first. */
LCM_API int  lcm_all_vs_all(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts,
                            const int32_t* q_ids, int n_q_frames, int q_stride_rows,
                            void* d_scores, size_t scores_cap, size_t* n_pairs, size_t* pair_offsets);
/* Same search through the ARGMIN kernel: besides every query row's best distance it finds the FIRST train row that
 * attains it (BFMatcher's trainIdx) — "per-query min/argmin" of BASELINE.json's north_star — and makes the indices
 * observable without 8 KB of keys per pair: d_index_sums (device, one uint32 per pair, same order as d_scores)
 * receives the sum mod 2^32 of the train indices of the pair's GOOD matches (0 for an empty pair). */
LCM_API int  lcm_all_vs_all_argmin(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts,
                                   const int32_t* q_ids, int n_q_frames, int q_stride_rows,
                                   void* d_scores, size_t scores_cap, void* d_index_sums,
                                   size_t* n_pairs, size_t* pair_offsets);
/* Same search with the loop test fused on the device (README.md:123-126; BASELINE.json configs[3]): scores never
 * leave HBM, a second kernel applies similarity > threshold && good >= min_matches per pair in IEEE double and
 * compacts the candidates; `out` (host) receives them sorted by (current_frame_id, matched_frame_id).
 * q_keypoints (host, optional) = keypoint counts of an external query set (NULL: row counts).  *n_out = number
 * found (LCM_ERR_CAPACITY if > cap); *n_pairs_out (optional) = pairs scored. */
LCM_API int  lcm_all_vs_all_loops(lcm_handle* h, const void* d_query_rows, const int32_t* d_query_counts,
                                  const int32_t* q_ids, const int32_t* q_keypoints, int n_q_frames, int q_stride_rows,
                                  lcm_loop_candidate* out, size_t cap, size_t* n_out, size_t* n_pairs_out);
LCM_API int  lcm_last_launch_info(const lcm_handle* h, lcm_launch_info* info);
/* Device address and record count of the score array the last lcm_all_vs_all_loops left in HBM (same order as
 * lcm_all_vs_all would write; valid until the next bulk call on this handle; NULL / 0 if there is none). */
LCM_API int  lcm_last_bulk_scores(const lcm_handle* h, const void** d_scores, size_t* n_records);

/* Select the kernel variant of the bulk / online scoring (A/B measurement; results are identical):
 * 0 = query-row-per-lane, best distance per query row; 1 = same + the first train row attaining it (argmin, by
 * 8-row group keys and a re-scan of the winning group); 2 / 3 = the train-row-per-lane mapping with LDS-staged
 * queries and wavefront shuffle reductions, distances only / (dist, idx) keys.
 * 4 = OPT-IN, NOT the product path: BASELINE.json's north_star rules the matrix cores out ("no MFMA"); this variant
 * runs the search on v_mfma_i32_32x32x32_i8 over a +1 / -1 int8 image of
 * the descriptors (<q,t> = 256 - 2 d, exact) to MEASURE what that rule costs; same records bit for bit.  5 = the same
 * on the block-scaled fp4 instruction (v_mfma_scale_f32_32x32x64_f8f6f4, +1/-1 as e2m1, all scales 1.0, exact in the
 * f32 accumulator).  4 / 5 serve the bulk search — lcm_all_vs_all_argmin included: the matrix instruction finds the
 * first 32-row tile that reaches the best dot product, an exact XOR + popcount re-scan of that tile finds the row —
 * and the online queries (single, stored-frame, micro-batched); pair mode and cross_check keep running the
 * vector-ALU kernels under them, as do query frames above 2048 rows. */
LCM_API int  lcm_set_kernel_variant(lcm_handle* h, int variant);

/* ---- multi-GPU: one process, W devices, stored frames sharded cyclically by arrival position ------------------- */
/* What a LoopClosingSystem (include/loop_closing.hpp:29-31) holds instead of one matcher when it is given more than
 * one MI355X: one lcm_handle + one host thread per device, an RCCL communicator (ncclCommInitAll) over them.  Stored
 * frame number p (arrival order) lives on device p mod W.  lcm_group_all_vs_all all-gathers the shard arenas over
 * xGMI so that every device sees every frame as a query, scores on all devices at once, gathers the 8-byte records to
 * the first device (grouped ncclSend / ncclRecv), un-permutes them on that device and returns the SAME array, in
 * the same order, that a single lcm_handle holding all frames would have produced with lcm_all_vs_all. */
typedef struct lcm_group lcm_group;
typedef struct lcm_group_info {
    int32_t  n_devices;
    int32_t  rccl_ranks;                     /* ncclCommCount of the group's communicator; 0 for a loopback rehearsal group */
    uint64_t pairs, distances, algo_bytes;   /* summed over the shards */
    double   kernel_ms_max;                  /* slowest shard's scoring kernel(s) */
    double   gather_merge_ms;                /* first device: its kernel's end -> records gathered (RCCL) and merged */
    double   download_ms;                    /* merged array -> host */
    uint64_t gathered_query_bytes;           /* per device: bytes received by THIS call's all-gather of the shard arenas
                                              * (0 when it was skipped: nothing appended since the last search) */
    uint64_t gathered_score_bytes;           /* bytes of score records (+ index checksums) that crossed xGMI to the first device */
    double   allgather_ms;                   /* first device: duration of the arena all-gather (0 when skipped) */
    double   kernel_ms[8];                   /* per device: its shard's scoring kernel(s) */
    uint64_t shard_pairs[8];                 /* per device: pairs its shard scored */
    int32_t  arena_gather_skipped;           /* 1: the gathered query buffers of the previous search were reused */
    int32_t  loopback;                       /* 1: rehearsal group (all shards on one device, no RCCL) */
} lcm_group_info;
/* device_ids == NULL: devices 0 .. n_devices-1.  n_devices <= 8. */
LCM_API int  lcm_group_create(const lcm_params* params, int n_devices, const int* device_ids, lcm_group** out);
/* The same group with its two exchange steps as device-to-device copies over xGMI (hipMemcpyAsync between the devices'
 * buffers, peer access enabled where available) instead of RCCL calls.  lcm_group_create falls back to this form by
 * itself when the RCCL communicator cannot be created; lcm_group_transport says which one a group uses ("rccl",
 * "peer copies (...)", "loopback (...)").  Results are identical. */
LCM_API int  lcm_group_create_peer(const lcm_params* params, int n_devices, const int* device_ids, lcm_group** out);
LCM_API const char* lcm_group_transport(const lcm_group* g);
/* Rehearsal form for boxes with ONE GPU: n_shards matchers on device_id, the two exchange steps as device-local copies
 * instead of RCCL calls.  Same results as any other group; exists so that the multi-device index arithmetic (cyclic
 * ownership, rank-major query buffer, gatherv offsets, device merge) is exercised for W > 1 where no second GPU is. */
LCM_API int  lcm_group_create_loopback(const lcm_params* params, int n_shards, int device_id, lcm_group** out);
LCM_API void lcm_group_destroy(lcm_group* g);
LCM_API int  lcm_group_size(const lcm_group* g);                    /* W */
LCM_API int  lcm_group_db_size(const lcm_group* g);                 /* frames over all shards */
LCM_API int  lcm_group_handle(lcm_group* g, int rank, lcm_handle** out);   /* the shard's own matcher (borrowed) */
LCM_API int  lcm_group_set_params(lcm_group* g, const lcm_params* params);
LCM_API int  lcm_group_reserve(lcm_group* g, int n_frames, int max_desc);
LCM_API int  lcm_group_append(lcm_group* g, int frame_id, const uint8_t* desc, int n, int n_keypoints);
LCM_API int  lcm_group_clear(lcm_group* g);
/* Keep the first n_frames frames (arrival order), drop the rest; tickets in flight become void (lcm_db_truncate per shard). */
LCM_API int  lcm_group_truncate(lcm_group* g, int n_frames);
/* Snapshot / resume in lcm_db_save's own file format, frames in arrival order: a file written by a group of 8 loads
 * into a single handle or a group of any size, and the other way round.  Load replaces the group's contents (same
 * validation as lcm_db_load: header checked against the file before anything is touched; empty group on a later error). */
LCM_API int  lcm_group_save(lcm_group* g, const char* path);
LCM_API int  lcm_group_load(lcm_group* g, const char* path);
LCM_API int  lcm_group_sync(lcm_group* g);                                   /* lcm_sync on every shard */
LCM_API int  lcm_group_set_tuning(lcm_group* g, int knob, int value);        /* lcm_set_tuning on every shard */
LCM_API int  lcm_group_set_kernel_variant(lcm_group* g, int variant);        /* lcm_set_kernel_variant on every shard */
/* out_scores (host) receives *n_pairs records in (query ascending, stored ascending) order; pair_offsets (host,
 * optional, db_size + 1 entries); out_scores == NULL sizes only.  The all-gather of the shard arenas is skipped when
 * nothing was appended since the previous search (lcm_group_info.arena_gather_skipped). */
LCM_API int  lcm_group_all_vs_all(lcm_group* g, lcm_score* out_scores, size_t cap, size_t* n_pairs, size_t* pair_offsets);
/* lcm_all_vs_all_argmin over the group: the same search through the ARGMIN kernel; out_index_sums (host, one uint32 per
 * pair, same order) receives the per-pair index checksums, gathered and merged like the records.  A group of ONE device
 * returns byte for byte what lcm_all_vs_all_argmin writes on a single handle. */
LCM_API int  lcm_group_all_vs_all_argmin(lcm_group* g, lcm_score* out_scores, uint32_t* out_index_sums, size_t cap,
                                         size_t* n_pairs, size_t* pair_offsets);
/* lcm_all_vs_all_loops over the group (BASELINE.json configs[3] on several devices): every device scores its shard and
 * applies the loop test to its own records ON THE DEVICE; only the candidates leave the devices — each shard's over its
 * own PCIe link, all links at once — and are merged on the host into (current id, matched id) order.  *n_out = number
 * found (LCM_ERR_CAPACITY if > cap, nothing written); *n_pairs_out (optional) = pairs scored. */
LCM_API int  lcm_group_all_vs_all_loops(lcm_group* g, lcm_loop_candidate* out, size_t cap, size_t* n_out, size_t* n_pairs_out);
LCM_API int  lcm_group_last_info(const lcm_group* g, lcm_group_info* info);
/* lcm_query_scores / lcm_detect_loops over all shards (query uploaded to every device; per-shard records interleaved
 * on the host: a few KB per query). */
LCM_API int  lcm_group_query_scores(lcm_group* g, const uint8_t* query, int nq, int query_frame_id,
                                    lcm_score* out_scores, int32_t* out_frame_ids, int cap, int* n_out);
/* lcm_query_submit_batch + lcm_query_collect_batch over all shards: up to 16 frames per launch per device (the
 * streaming mode of BASELINE.json configs[4] inside one process); records of query 0 first, offsets[n_queries + 1]. */
LCM_API int  lcm_group_query_scores_batch(lcm_group* g, const uint8_t* const* queries, const int* nq,
                                          const int* query_frame_ids, int n_queries,
                                          lcm_score* out_scores, size_t cap, size_t* n_out, size_t* offsets);
/* The asynchronous form (what keeps W devices busy in streaming mode): submit returns once every device has the batch
 * ENQUEUED — each from its own host thread; the callers' buffers are free then — with one group ticket; up to 4 may be in
 * flight (submit batch k + 1, append, then collect batch k).  Collect waits for that ticket; every device's thread writes
 * its records straight into their places of the single-device order.  A too-small `cap` keeps the ticket valid; a
 * lcm_group_clear / _truncate in between voids it (LCM_ERR_NOT_FOUND). */
LCM_API int  lcm_group_query_submit_batch(lcm_group* g, const uint8_t* const* queries, const int* nq,
                                          const int* query_frame_ids, int n_queries, int* ticket);
LCM_API int  lcm_group_query_collect_batch(lcm_group* g, int ticket, lcm_score* out_scores, size_t cap, size_t* n_out,
                                           size_t* offsets);
/* lcm_online_stats_read over the shards: work summed, kernel_ms = the slowest shard's (the devices run side by side). */
LCM_API int  lcm_group_online_stats_read(lcm_group* g, lcm_online_stats* out, int reset);
LCM_API int  lcm_group_detect_loops(lcm_group* g, int current_frame_id, const uint8_t* query, int nq, int n_keypoints,
                                    lcm_loop_candidate* out, int cap, int* n_out);
/* Host-only (no device needed): merge W per-shard score arrays — shard r in (query ascending, owned stored ascending)
 * order, as a process-per-GPU deployment gathers them (bench.py, torch.distributed) — into the single-device order.
 * ids: all n_frames frame ids, ascending.  out == NULL sizes only (*n_out, offsets[n_frames + 1] optional). */
LCM_API int  lcm_merge_shard_scores(const lcm_score* const* shard_scores, const size_t* shard_counts, int world,
                                    const int32_t* ids, int n_frames, int min_gap,
                                    lcm_score* out, size_t cap, size_t* n_out, size_t* offsets);

/* The same merge on the device (what lcm_group_all_vs_all runs on its first device): d_gathered holds the W shard
 * arrays back to back; d_merged receives the single-device order.  world <= 8. */
LCM_API int  lcm_merge_shard_scores_device(lcm_handle* h, const void* d_gathered, const size_t* shard_counts, int world,
                                           const int32_t* ids, int n_frames, int min_gap,
                                           void* d_merged, size_t cap, size_t* n_out);

/* Measurement knobs (defaults are the measured optima; results never depend on them):
 *   LCM_TUNE_ITEM_SLOTS   stored frames per work item of the bulk search, 1..64; 0 = automatic
 *   LCM_TUNE_ONLINE_SPLIT query rows per lane of the online split mode: 1, 2, 4 (256-thread workgroups), 16 / 32 (64- /
 *                         128-thread workgroups of 8 rows per lane); 0 = never split; -1 = automatic
 *   LCM_TUNE_PACKED       bulk search with the query rows of consecutive frames packed into full 2048-row workgroups:
 *                         1 = always, 0 = never, -1 = automatic (when packing saves lane slots), 2 = always, with
 *                         1536-row workgroups (6 rows per lane at 8 waves per SIMD: an A/B, within 0.5 % of 1)
 *   LCM_TUNE_ONLINE_STREAMS 1 (default) = each of the 4 query slots enqueues on its own stream, so consecutive online
 *                         queries overlap (upload and first workgroups of one under the draining tail of the other);
 *                         0 = everything on the handle's stream
 *   LCM_TUNE_PACKED_SCRATCH_MB  packed route: MiB of per-row scratch per chunk (default 1024; a search larger than one
 *                         chunk runs chunk after chunk; halved for the search at hand, down to 64, when the allocation
 *                         fails — the next plan starts from the configured size again).  Device footprint of a handle
 *                         beyond its database: this scratch (allocated at the first packed search, given back when a
 *                         later search needs less than a quarter of it), 8 bytes per pair of the fused loop search's
 *                         score array, and the staging of up to 4 online queries.
 *   LCM_TUNE_PAIR_UPLOAD_KERNEL  pair mode, calls of <= 64 M distances: the staging block is uploaded by a kernel on the
 *                         compute queue (1, default) or by hipMemcpyAsync (0)
 *   LCM_TUNE_PAIR_HOST_FOLD      ... and the fold kernel writes the folded keys straight into pinned host memory (1,
 *                         default) or into device memory followed by a copy (0) */
typedef enum lcm_tuning { LCM_TUNE_ITEM_SLOTS = 0, LCM_TUNE_ONLINE_SPLIT = 1, LCM_TUNE_PACKED = 2, LCM_TUNE_ONLINE_STREAMS = 3,
                          LCM_TUNE_PACKED_SCRATCH_MB = 4, LCM_TUNE_PAIR_UPLOAD_KERNEL = 5, LCM_TUNE_PAIR_HOST_FOLD = 6 } lcm_tuning;
LCM_API int  lcm_set_tuning(lcm_handle* h, int knob, int value);

/* Device scratch helpers so a host program needs no other allocator (plain hipMalloc/hipFree/hipMemcpy). */
LCM_API int  lcm_dev_alloc(lcm_handle* h, size_t bytes, void** d_ptr);
LCM_API int  lcm_dev_free(lcm_handle* h, void* d_ptr);
LCM_API int  lcm_dev_upload(lcm_handle* h, void* d_dst, const void* src, size_t bytes);
LCM_API int  lcm_dev_download(lcm_handle* h, void* dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* LCM_H_ */
