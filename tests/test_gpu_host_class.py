"""The C++ host mirror of loop_closing::LoopClosingSystem (csrc/loop_closing_system.hpp), driven through its C shim:
processFrame / matchFeatures / detectLoops / getLoopClosures / saveResults vs the oracle."""
import os

import numpy as np
import pytest

from conftest import fast_detect_loops

pytestmark = pytest.mark.gpu


def cand_tuples(c):
    return [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"])) for r in c]


def test_process_frames_online_equals_oracle(pkg, oracle, tmp_path):
    fs = pkg.synth.make_frames(40, 400, seed=3, ragged=True, dup_frac=0.3)
    gap, thr = 6, 0.15
    sys_ = pkg.LoopClosingSystem(thr, gap)
    try:
        p = oracle.default_params(min_gap=gap, sim_threshold=thr)
        for f in range(fs.n_frames):
            sys_.processFrame(fs.frame(f), int(fs.ids[f]))
            if f in (1, 17, fs.n_frames - 1):                 # consecutive-frame matches: previous = query, current = train
                cm = sys_.getConsecutiveMatches()
                om, _ = oracle.match_features(fs.frame(f - 1), fs.frame(f), p)
                for k in ("query_idx", "train_idx", "img_idx", "distance"):
                    np.testing.assert_array_equal(cm[k], om[k])
        assert sys_.numFrames() == fs.n_frames
        fast = [t for c in range(fs.n_frames) for t in fast_detect_loops(oracle, fs, c, p)]
        got = sys_.getLoopClosures()
        assert len(fast) > 0 and cand_tuples(got) == fast
        want = got                                           # (scalar oracle spot check below)
        c0 = int(got["current_frame_id"][0])
        assert cand_tuples(oracle.detect_loops(fs.rows, fs.counts, fs.ids, c0, p)) == [t for t in fast if t[0] == c0]
        # detectLoops on a stored frame == what processFrame recorded for it
        c = int(want["current_frame_id"][0])
        one = sys_.detectLoops(c)
        np.testing.assert_array_equal(one["matched_frame_id"], want["matched_frame_id"][want["current_frame_id"] == c])
        # matchFeatures(frame1, frame2) on two stored frames
        a, b = int(want["current_frame_id"][0]), int(want["matched_frame_id"][0])
        m = sys_.matchFeatures(a, b)
        om, _ = oracle.match_features(fs.frame(a), fs.frame(b), p)     # ids == positions here
        for f in ("query_idx", "train_idx", "img_idx", "distance"):
            np.testing.assert_array_equal(m[f], om[f])
        assert len(m) == int(want["num_matches"][0])
        # matchLoopClosures: README.md:101 "Re-match features on identified loop frames" — all of a frame's closures, one launch
        cur_ids = sorted(set(int(x) for x in want["current_frame_id"]))
        busiest = max(cur_ids, key=lambda c: int((want["current_frame_id"] == c).sum()))
        lists = sys_.matchLoopClosures(busiest)
        mine = want[want["current_frame_id"] == busiest]
        assert len(lists) == len(mine) >= 1
        for rec, lst in zip(mine, lists):
            om, _ = oracle.match_features(fs.frame(busiest), fs.frame(int(rec["matched_frame_id"])), p)
            np.testing.assert_array_equal(lst, om.astype(lst.dtype))
            assert len(lst) == int(rec["num_matches"])           # the list IS what num_matches counted
        assert sys_.matchLoopClosures(0) == []
        # saveResults: README.md:142-165 text format
        out = tmp_path / "loop_closing_results"
        sys_.saveResults(str(out))
        txt = (out / "loop_closures.txt").read_text()
        assert f"Total frames processed: {fs.n_frames}" in txt
        assert f"Loop closures detected: {len(want)}" in txt
        first = want[0]
        assert f"Frame {first['current_frame_id']} <-> Frame {first['matched_frame_id']}\n  Matches: {first['num_matches']}\n  Similarity: " in txt
        assert txt.count(" <-> ") == len(want)
    finally:
        sys_.close()


def test_header_default_threshold_and_errors(pkg, oracle):
    fs = pkg.synth.make_frames(36, 300, seed=8)
    sys_ = pkg.LoopClosingSystem()                     # header defaults: 0.7 / 30 (include/loop_closing.hpp:31)
    try:
        for f in range(fs.n_frames):
            sys_.processFrame(fs.frame(f), int(fs.ids[f]))
        p = oracle.default_params(min_gap=30, sim_threshold=0.7)
        fast = [t for c in range(fs.n_frames) for t in fast_detect_loops(oracle, fs, c, p)]
        assert cand_tuples(sys_.getLoopClosures()) == fast
        with pytest.raises(pkg.LcmError):
            sys_.processFrame(fs.frame(0), 3)          # ids must increase
        with pytest.raises(pkg.LcmError):
            sys_.detectLoops(9999)
    finally:
        sys_.close()


@pytest.mark.parametrize("batched", [False, True])
def test_sharded_host_objects_partition_the_candidates(pkg, oracle, batched):
    """Two LoopClosingSystem objects with shard_world = 2 on the one GPU: their candidates merge to the full set —
    frame by frame, and through processFrames' micro-batches (record k of a shard = the k-th frame that shard owns)."""
    fs = pkg.synth.make_frames(30, 300, seed=5, dup_frac=0.3)
    gap = 4
    shards = [pkg.LoopClosingSystem(0.15, gap, 0, r, 2) for r in range(2)]
    try:
        if batched:
            for s in shards:
                s.processFrames([fs.frame(f) for f in range(fs.n_frames)], [int(x) for x in fs.ids])
        for f in range(0 if not batched else fs.n_frames, fs.n_frames):
            for s in shards:
                s.processFrame(fs.frame(f), int(fs.ids[f]))
        p = oracle.default_params(min_gap=gap)
        fast = [t for c in range(fs.n_frames) for t in fast_detect_loops(oracle, fs, c, p)]
        got = np.concatenate([s.getLoopClosures() for s in shards])
        order = np.lexsort((got["matched_frame_id"], got["current_frame_id"]))
        got = got[order]
        assert len(fast) > 0 and cand_tuples(got) == fast
        for r, s in enumerate(shards):
            assert all(int(mid) % 2 == r for mid in s.getLoopClosures()["matched_frame_id"])
    finally:
        for s in shards:
            s.close()


def test_sharded_loop_search_driver_world1(pkg, oracle):
    """sharding.ShardedLoopSearch with the real Matcher as its scorer (world 1: no process group needed)."""
    fs = pkg.synth.make_frames(30, 300, seed=5, dup_frac=0.3)
    p = pkg.default_params()
    p.min_gap = 4
    with pkg.Matcher(p) as m:
        search = pkg.sharding.ShardedLoopSearch(m)
        got = []
        for f in range(fs.n_frames):
            merged, ids, cands = search.process_frame(fs.frame(f), int(fs.ids[f]))
            got += cands
        op = oracle.default_params(min_gap=4)
        want = [t for c in range(fs.n_frames) for t in fast_detect_loops(oracle, fs, c, op)]
        assert got == want and len(want) > 0


def test_gap_by_position_with_sparse_frame_ids(pkg, oracle):
    """setGapByPosition: "at least min_loop_gap frames ago" counted on arrival positions — the tree's own loop,
    src/main.cpp:1375-1379 — vs on frame ids (default).  With ids 0, 3, 6, ... (frame_skip = 3 video numbers) the two
    readings select different frame sets; each must equal the oracle run with the matching key (ids, or positions)."""
    fs = pkg.synth.make_frames(36, 300, seed=5, ragged=True, dup_frac=0.3)
    fs.rows[20] = fs.rows[17]; fs.counts[20] = fs.counts[17]      # a revisit 3 frames later: a loop by id (9 >= 6), not by position
    gap, thr = 6, 0.05
    sparse_ids = (np.arange(fs.n_frames) * 3).astype(np.int32)
    results = {}
    for by_pos in (False, True):
        sys_ = pkg.LoopClosingSystem(thr, gap)
        try:
            if by_pos:
                sys_.setGapByPosition(True)
            for f in range(fs.n_frames):
                sys_.processFrame(fs.frame(f), int(sparse_ids[f]))
            if by_pos:
                with pytest.raises(pkg.capi.LcmError):
                    sys_.setGapByPosition(False)               # only before the first frame
            got = sys_.getLoopClosures()
            results[by_pos] = {(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"])) for r in got}
            # the oracle keyed the same way: ids (sparse) or positions; reported ids are always the caller's
            keys = np.arange(fs.n_frames, dtype=np.int32) if by_pos else sparse_ids
            p50 = oracle.default_params(min_gap=gap, sim_threshold=thr)
            want = set()
            fs_k = pkg.synth.FrameSet(rows=fs.rows, counts=fs.counts, ids=keys, seed=fs.seed)
            for c in range(fs.n_frames):
                for (_, matched_key, good, _sim) in fast_detect_loops(oracle, fs_k, c, p50):
                    matched_pos = int(np.searchsorted(keys, matched_key))
                    want.add((int(sparse_ids[c]), int(sparse_ids[matched_pos]), good))
            c0 = fs.n_frames - 1                                  # ... and the scalar restatement for one frame
            assert {(int(sparse_ids[c0]), int(sparse_ids[int(np.searchsorted(keys, int(r["matched_frame_id"])))]), int(r["num_matches"]))
                    for r in oracle.detect_loops(fs.rows, fs.counts, keys, c0, p50)} == {w for w in want if w[0] == int(sparse_ids[c0])}
            assert results[by_pos] == want
            if len(got):
                c = int(got["current_frame_id"][-1])
                lists = sys_.matchLoopClosures(c)
                mine = got[got["current_frame_id"] == c]
                for rec, lst in zip(mine, lists):
                    assert len(lst) == int(rec["num_matches"])
        finally:
            sys_.close()
    # by id: frame c sees ids <= 3c - 6 = positions <= c - 2; by position: positions <= c - 6 — a strict subset
    assert results[True] < results[False] and (60, 51, int(fs.counts[17])) in results[False] - results[True]


@pytest.mark.parametrize("by_pos,id_step", [(False, 1), (False, 3), (True, 3)])
def test_process_frames_in_micro_batches_equals_frame_by_frame(pkg, oracle, by_pos, id_step):
    """processFrames (micro-batches of up to 16 frames per launch, cut so that no launch holds frames min_loop_gap apart)
    leaves the same frames, loop closures and consecutive matches as processFrame called frame by frame — dense ids,
    sparse ids (the id span forces smaller batches), and the positional gap reading."""
    fs = pkg.synth.make_frames(70, 300, seed=9, ragged=True, dup_frac=0.3)
    fs.rows[33] = fs.rows[20]; fs.counts[33] = fs.counts[20]
    ids = (np.arange(fs.n_frames) * id_step).astype(np.int32)
    gap, thr = 10, 0.05
    one = pkg.LoopClosingSystem(thr, gap)
    many = pkg.LoopClosingSystem(thr, gap)
    try:
        for s_ in (one, many):
            if by_pos:
                s_.setGapByPosition(True)
        for f in range(fs.n_frames):
            one.processFrame(fs.frame(f), int(ids[f]))
        # uneven calls: 1 frame, 23 frames, an empty call, the rest
        many.processFrames([fs.frame(0)], [int(ids[0])])
        many.processFrames([fs.frame(f) for f in range(1, 24)], [int(ids[f]) for f in range(1, 24)])
        many.processFrames([], [])
        many.processFrames([fs.frame(f) for f in range(24, fs.n_frames)], [int(ids[f]) for f in range(24, fs.n_frames)])
        assert many.numFrames() == one.numFrames() == fs.n_frames
        a, b = one.getLoopClosures(), many.getLoopClosures()
        assert len(a) > 0 and cand_tuples(a) == cand_tuples(b)       # (field by field: the record has 4 padding bytes)
        np.testing.assert_array_equal(one.getConsecutiveMatches(), many.getConsecutiveMatches())
        c = int(a["current_frame_id"][-1])
        assert cand_tuples(one.detectLoops(c)) == cand_tuples(many.detectLoops(c))
        with pytest.raises(pkg.capi.LcmError):
            many.processFrames([fs.frame(3)], [int(ids[3])])          # ids must increase
        assert many.numFrames() == fs.n_frames
    finally:
        one.close(); many.close()
