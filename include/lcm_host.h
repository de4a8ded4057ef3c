/*
 * lcm_host.h — C shim over the C++ host class loop_closing::LoopClosingSystem
 * (slam-loop-closing_amd/csrc/loop_closing_system.hpp), the host-side mirror of the reference's
 * include/loop_closing.hpp:29-80 for the Hamming path.  It exists so that non-C++ callers (the ctypes parity
 * tests) can drive the same object a C++ program would; C++ callers include loop_closing_system.hpp directly.
 * Same conventions as lcm.h (int status, lcm_last_error()).
 */
#ifndef LCM_HOST_H_
#define LCM_HOST_H_

#include "lcm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lcs_system lcs_system;

/* LoopClosingSystem(loop_threshold, min_loop_gap) — include/loop_closing.hpp:31 (defaults 0.7 / 30).
 * shard_rank/shard_world: this process owns stored frames whose arrival position % shard_world == shard_rank
 * (1-GPU: 0 / 1). */
LCM_API int  lcs_create(double loop_threshold, int min_loop_gap, int device_id, int shard_rank, int shard_world,
                        lcs_system** out);
/* The multi-device constructor (loop_closing_system.hpp): the same system over n_devices MI355X of one node behind an
 * lcm_group (device_ids == NULL: 0 .. n_devices-1).  loopback_device >= 0: rehearsal form — n_devices shards on that one
 * device (lcm_group_create_loopback); pass -1 for real devices. */
LCM_API int  lcs_create_group(double loop_threshold, int min_loop_gap, int n_devices, const int* device_ids,
                              int loopback_device, lcs_system** out);
LCM_API void lcs_destroy(lcs_system* s);
/* processFrame (include/loop_closing.hpp:34) with the ORB stage already done: `desc` is what
 * detectFeatures would have put in Frame::descriptors (rows x 32, CV_8U), n_keypoints = keypoints.size(). */
LCM_API int  lcs_process_frame(lcs_system* s, const uint8_t* desc, int rows, int n_keypoints, int frame_id);
/* processFrames: the same for n frames in order, scored in micro-batches of up to 16 frames per launch (the online
 * path's fast form; exact for any input: a launch only holds frames closer together than min_loop_gap).
 * n_keypoints may be NULL (= rows). */
LCM_API int  lcs_process_frames(lcs_system* s, const uint8_t* const* desc, const int* rows, const int* n_keypoints,
                                const int* frame_ids, int n);
/* matchFeatures(frame1, frame2) for two stored frames given by id (include/loop_closing.hpp:40). */
LCM_API int  lcs_match_features(lcs_system* s, int frame1_id, int frame2_id, lcm_dmatch* out, int cap, int* n_out);
/* detectLoops(current_frame_id) (include/loop_closing.hpp:48). */
LCM_API int  lcs_detect_loops(lcs_system* s, int current_frame_id, lcm_loop_candidate* out, int cap, int* n_out);
/* matches between the previous and the current frame computed by the last lcs_process_frame (README.md:96-97) */
LCM_API int  lcs_get_consecutive_matches(const lcs_system* s, lcm_dmatch* out, int cap, int* n_out);
/* matchLoopClosures(current_frame_id): DMatch lists of every loop closure recorded for that frame, one launch
 * (README.md:101 "Re-match features on identified loop frames"); lists back to back in `out`, bounds in offsets. */
LCM_API int  lcs_match_loop_closures(lcs_system* s, int current_frame_id, lcm_dmatch* out, size_t cap, size_t* offsets,
                                     int offsets_cap, int* n_lists);
/* setGapByPosition: count "min_loop_gap frames ago" (README.md:122) on arrival positions — the reading of the tree's own
 * loop, src/main.cpp:1375-1379 — instead of frame ids (default; identical for dense ids).  Before the first frame only. */
LCM_API int  lcs_set_gap_by_position(lcs_system* s, int on);
LCM_API int  lcs_num_frames(const lcs_system* s);                       /* getFrames().size()        hpp:60 */
LCM_API int  lcs_num_loop_closures(const lcs_system* s);                /* getLoopClosures().size()  hpp:63 */
LCM_API int  lcs_get_loop_closures(const lcs_system* s, lcm_loop_candidate* out, int cap, int* n_out);
LCM_API int  lcs_save_results(lcs_system* s, const char* output_dir);   /* saveResults               hpp:66 */

#ifdef __cplusplus
}
#endif
#endif
