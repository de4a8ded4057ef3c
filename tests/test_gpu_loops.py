"""Database / loop-search entry points vs the oracle: lcm_query_scores, lcm_detect_loops, lcm_all_vs_all,
streaming append, sharded == single."""
import numpy as np
import pytest

from conftest import fast_all_vs_all, fast_detect_loops

pytestmark = pytest.mark.gpu


def cand_tuples(c):
    return [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"])) for r in c]


def fill(m, fs, positions=None):
    m.clear()
    for f in (range(fs.n_frames) if positions is None else positions):
        m.append(int(fs.ids[f]), fs.frame(f))


def gpu_all_vs_all(m, fs=None, q_ids=None, d_rows=0, d_counts=0, stride=0):
    n, offs = m.all_vs_all_plan(d_rows, d_counts, q_ids, stride)
    d_scores = m.dev_alloc(max(n, 1) * 8)
    try:
        got = m.all_vs_all(d_scores, n, d_rows, d_counts, q_ids, stride)
        assert got == n
        out = np.zeros(max(n, 1), m_score_dtype(m))
        m.sync()
        m.dev_download(d_scores, out)
    finally:
        m.dev_free(d_scores)
    return out[:n], offs


def m_score_dtype(m):
    from conftest import load_package
    return load_package().capi.SCORE_DTYPE


@pytest.mark.parametrize("n_frames,max_desc,gap,ragged", [(100, 500, 30, False), (40, 300, 5, True), (12, 2000, 3, True),
                                                          (14, 1000, 2, True), (14, 1400, 2, True)])   # 128- and 192-thread workgroups
def test_all_vs_all_bit_exact(matcher, oracle, pkg, n_frames, max_desc, gap, ragged):
    """cfg1 (100 x 500, gap 30) is BASELINE.json's configs[0]; the others stress ragged / full-size frames."""
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=pkg.synth.BASE_SEED + 1, ragged=ragged, dup_frac=0.2)
    if ragged:
        fs.counts[2] = 0
    matcher.set_params(min_gap=gap)
    try:
        fill(matcher, fs)
        got, offs = gpu_all_vs_all(matcher)
        want, woffs = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap))
        np.testing.assert_array_equal(offs.astype(np.int64), woffs)
        np.testing.assert_array_equal(got, want)
        assert len(got) == pkg.synth.n_pairs_all_vs_all(n_frames, gap)
        info = matcher.launch_info()
        assert info.pairs == len(got) and info.kernel_ms > 0
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_query_scores_and_detect_loops(matcher, oracle, pkg):
    fs = pkg.synth.make_frames(64, 256, seed=4, ragged=True, dup_frac=0.3)
    matcher.set_params(min_gap=7)
    p = oracle.default_params(min_gap=7)
    try:
        fill(matcher, fs)
        for cur in (0, 6, 7, 8, 33, 63):
            scores, ids = matcher.query_scores(fs.frame(cur), int(fs.ids[cur]))
            want = [oracle.pair_score(fs.frame(cur), fs.frame(i), p) for i in range(fs.n_frames)
                    if fs.ids[cur] - fs.ids[i] >= 7]
            assert len(scores) == len(want)
            assert ids.tolist() == [int(fs.ids[i]) for i in range(fs.n_frames) if fs.ids[cur] - fs.ids[i] >= 7]
            for a, b in zip(scores, want):
                assert a == b
            # detectLoops on the stored frame and on an explicitly passed frame agree with the oracle
            want_c = oracle.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)
            for got_c in (matcher.detect_loops(int(fs.ids[cur])), matcher.detect_loops(int(fs.ids[cur]), fs.frame(cur))):
                assert len(got_c) == len(want_c)
                for f in ("current_frame_id", "matched_frame_id", "num_matches", "similarity_score"):
                    np.testing.assert_array_equal(got_c[f], want_c[f])
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_detect_loops_finds_revisits(matcher, oracle, pkg):
    """Place-structured data must yield real loop candidates, otherwise the test above compares empty lists."""
    fs = pkg.synth.make_frames(48, 600, seed=21)
    matcher.set_params(min_gap=10)
    try:
        fill(matcher, fs)
        total = 0
        p = oracle.default_params(min_gap=10)
        for cur in range(10, 48):
            got = matcher.detect_loops(int(fs.ids[cur]))
            assert cand_tuples(got) == fast_detect_loops(oracle, fs, cur, p)
            total += len(got)
        assert total > 0
        # and the scalar oracle's own detectLoops on two frames
        for cur in (20, 47):
            want = oracle.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)
            assert cand_tuples(matcher.detect_loops(int(fs.ids[cur]))) == cand_tuples(want)
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_streaming_append_equals_batch(matcher, oracle, pkg):
    """Online mode: append frame, query the next one, append it, ... == the all-vs-all batch result."""
    fs = pkg.synth.make_frames(50, 320, seed=13, ragged=True)
    gap = 4
    matcher.set_params(min_gap=gap)
    try:
        matcher.clear()
        online = []
        for f in range(fs.n_frames):
            s, _ = matcher.query_scores(fs.frame(f), int(fs.ids[f]))
            online.append(s.copy())
            matcher.append(int(fs.ids[f]), fs.frame(f))
        online = np.concatenate(online)
        want, _ = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap))
        np.testing.assert_array_equal(online, want)
        # and the stored rows read back unchanged
        for slot in (0, 17, 49):
            np.testing.assert_array_equal(matcher.read_frame(slot), fs.frame(slot))
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_sharded_equals_single(matcher, oracle, pkg):
    """K10: cyclic shards scored one after another on the one GPU, merged on the host == single-device result."""
    fs = pkg.synth.make_frames(45, 200, seed=17, ragged=True)
    gap, world = 5, 4
    matcher.set_params(min_gap=gap)
    d_rows = matcher.dev_alloc(fs.rows.nbytes)
    d_counts = matcher.dev_alloc(fs.counts.nbytes)
    try:
        matcher.dev_upload(d_rows, fs.rows)
        matcher.dev_upload(d_counts, fs.counts)
        fill(matcher, fs)
        single, offs = gpu_all_vs_all(matcher)
        shards = []
        for r in range(world):
            fill(matcher, fs, pkg.sharding.owned_positions(fs.n_frames, r, world))
            s, _ = gpu_all_vs_all(matcher, q_ids=fs.ids, d_rows=d_rows, d_counts=d_counts, stride=fs.stride_rows)
            shards.append(s)
            want_r, _ = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap), r, world)
            np.testing.assert_array_equal(s, want_r)
        merged, moffs = pkg.sharding.merge_shard_scores(shards, fs.ids, gap)
        np.testing.assert_array_equal(merged, single)
        np.testing.assert_array_equal(moffs, offs.astype(np.int64))
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_errors_are_loud(matcher, pkg):
    matcher.clear()
    matcher.append(5, np.zeros((3, 32), np.uint8))
    with pytest.raises(pkg.LcmError) as e:
        matcher.append(5, np.zeros((3, 32), np.uint8))
    assert e.value.code == -5                                   # LCM_ERR_ORDER
    with pytest.raises(pkg.LcmError) as e:
        matcher.detect_loops(1234)
    assert e.value.code == -6                                   # LCM_ERR_NOT_FOUND
    with pytest.raises(pkg.LcmError) as e:
        matcher.query_scores(np.zeros((65536, 32), np.uint8), 100)     # a frame holds at most 65535 rows
    assert e.value.code == -4                                   # LCM_ERR_CAPACITY
    matcher.set_params(cross_check=1)                           # cross_check keeps the 2048-row limit on query frames
    try:
        with pytest.raises(pkg.LcmError) as e:
            matcher.query_scores(np.zeros((2049, 32), np.uint8), 100)
        assert e.value.code == -4
    finally:
        matcher.set_params(cross_check=0)
    sc, _ = matcher.query_scores(np.zeros((2049, 32), np.uint8), 100)  # without it, 2049 rows are just a query frame
    assert len(sc) == 1 and int(sc[0]["n_train"]) == 3
    matcher.clear()


def test_match_stored_equals_match_features(matcher, oracle, pkg):
    """N1: the DMatch list of a detected loop, from device-resident rows (README.md:101 re-match)."""
    fs = pkg.synth.make_frames(20, 700, seed=23, ragged=True, dup_frac=0.4)
    fs.counts[6] = 0
    matcher.set_params(min_gap=3)
    try:
        fill(matcher, fs)
        for a, b in [(17, 1), (12, 12), (19, 4), (9, 6), (6, 2)]:
            got, md = matcher.match_stored(int(fs.ids[a]), int(fs.ids[b]))
            want, wmd = oracle.match_features(fs.frame(a), fs.frame(b), oracle.default_params(min_gap=3))
            assert md == wmd and len(got) == len(want)
            for f in ("query_idx", "train_idx", "img_idx", "distance"):
                np.testing.assert_array_equal(got[f], want[f])
        with pytest.raises(pkg.LcmError):
            matcher.match_stored(999, 1)
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_snapshot_restore_round_trip(matcher, pkg, tmp_path):
    fs = pkg.synth.make_frames(17, 260, seed=29, ragged=True)
    fs.counts[3] = 0
    matcher.set_params(min_gap=2)
    try:
        fill(matcher, fs)
        before, offs = gpu_all_vs_all(matcher)
        path = str(tmp_path / "db.lcm")
        matcher.save(path)
        matcher.clear()
        assert len(matcher) == 0
        matcher.load(path)
        assert len(matcher) == fs.n_frames
        for slot in (0, 3, 16):
            assert matcher.frame_info(slot) == (int(fs.ids[slot]), int(fs.counts[slot]), int(fs.counts[slot]))
            np.testing.assert_array_equal(matcher.read_frame(slot), fs.frame(slot))
        after, offs2 = gpu_all_vs_all(matcher)
        np.testing.assert_array_equal(before, after)
        np.testing.assert_array_equal(offs, offs2)
        blob = open(path, "rb").read()
        with open(path, "wb") as f:                             # truncated file: loud, and the handle stays usable
            f.write(blob[: len(blob) // 2])
        with pytest.raises(pkg.LcmError):
            matcher.load(path)
        with open(path, "wb") as f:
            f.write(b"garbage!" + blob[8:])
        with pytest.raises(pkg.LcmError):
            matcher.load(path)
        with open(path, "wb") as f:
            f.write(blob)
        matcher.load(path)
        again, _ = gpu_all_vs_all(matcher)
        np.testing.assert_array_equal(again, before)
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
def test_every_kernel_variant_is_bit_exact(matcher, oracle, pkg, variant):
    """0/1: row-per-lane (distances / argmin); 2/3: north_star's train-row-per-lane mapping with LDS-staged queries;
    4: the opt-in matrix-core variant (v_mfma_i32_32x32x32_i8 over +1/-1 operands)."""
    fs = pkg.synth.make_frames(26, 700, seed=41, ragged=True, dup_frac=0.4)
    fs.counts[4] = 0
    fs.counts[9] = 1
    gap = 3
    matcher.set_params(min_gap=gap)
    matcher.set_kernel_variant(variant)
    try:
        fill(matcher, fs)
        got, offs = gpu_all_vs_all(matcher)
        want, woffs = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap))
        np.testing.assert_array_equal(got, want)
        # pair mode (keys) through the same variant
        for a, b in [(20, 2), (11, 9), (9, 11), (25, 4)]:
            idx, d = matcher.match_pair(fs.frame(a), fs.frame(b))
            oi, od = oracle.bf_match(fs.frame(a), fs.frame(b))
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(d.astype(np.int32), od)
    finally:
        matcher.set_kernel_variant(0)
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_arena_grows_in_frames_and_rows(pkg, oracle):
    """No lcm_db_reserve: the arena must re-pitch (more rows per frame) and re-size (more frames) transparently."""
    rng = np.random.default_rng(6)
    sizes = [50, 120, 7, 2100, 300] + [40] * 80 + [2500, 3]
    frames = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in sizes]
    p = pkg.default_params()
    p.min_gap = 1
    with pkg.Matcher(p) as m:
        for i, f in enumerate(frames):
            m.append(i * 2, f)
        assert len(m) == len(frames)
        for slot in (0, 3, 4, 60, len(frames) - 2, len(frames) - 1):
            np.testing.assert_array_equal(m.read_frame(slot), frames[slot])
        q = frames[1]
        scores, ids = m.query_scores(q, 2 * len(frames))
        assert ids.tolist() == [2 * i for i in range(len(frames))]
        op = oracle.default_params(min_gap=1)
        for slot in (0, 3, 4, 60, len(frames) - 2, len(frames) - 1):
            assert scores[slot] == oracle.pair_score(q, frames[slot], op)
        # a stored frame wider than one workgroup's 2048 rows as the QUERY: the packed bulk route serves it
        cur = len(frames) - 2                                  # the 2500-row frame
        sc2, _ = m.query_scores(frames[cur], 2 * cur)
        for slot in (0, 3, 60, cur - 1):
            assert sc2[slot] == oracle.pair_score(frames[cur], frames[slot], op)
        m.set_params(min_matches=1, sim_threshold=0.0)
        c_stored = m.detect_loops(2 * cur)
        c_host = m.detect_loops(2 * cur, frames[cur])
        np.testing.assert_array_equal(c_stored, c_host)
        assert len(c_stored) == sum(1 for s in range(cur) if sizes[s] > 0)


def test_fused_on_device_loop_test(matcher, oracle, pkg):
    """configs[3] shape: scores stay on the device, the loop test runs there, only candidates come back."""
    fs = pkg.synth.make_frames(60, 500, seed=21, ragged=True, dup_frac=0.3)
    fs.counts[7] = 0
    gap = 10
    matcher.set_params(min_gap=gap)
    try:
        matcher.clear()
        for f in range(fs.n_frames):
            # keypoint counts that differ from the row counts exercise the similarity denominator
            matcher.append(int(fs.ids[f]), fs.frame(f), n_keypoints=int(fs.counts[f]) + (f % 3))
        got, n_pairs = matcher.all_vs_all_loops()
        assert n_pairs == pkg.synth.n_pairs_all_vs_all(fs.n_frames, gap)
        p = oracle.default_params(min_gap=gap)
        scores, offs = fast_all_vs_all(oracle, fs, p)
        want = []
        for c in range(fs.n_frames):
            for k, i in enumerate(i for i in range(fs.n_frames) if fs.ids[c] - fs.ids[i] >= gap):
                sc = scores[int(offs[c]) + k]
                ok, sim = oracle.loop_test(int(sc["good_count"]), int(fs.counts[c]) + c % 3, int(fs.counts[i]) + i % 3, p)
                if ok:
                    want.append((int(fs.ids[c]), int(fs.ids[i]), int(sc["good_count"]), sim))
        assert len(want) > 0
        assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"]))
                for r in got] == want
        with pytest.raises(pkg.LcmError) as e:
            matcher.all_vs_all_loops(cap=1)
        assert e.value.code == -4
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_short_database_split_mode_all_regimes(matcher, oracle, pkg):
    """lcm_query_scores cuts a pair's query rows over 8 / 4 / 2 / 1 workgroups depending on how many stored frames are
    eligible (< 256: 8 pieces, < 3072: 4, < 6144: 2, more: 1).  Every regime must give the single-workgroup answer
    (lcm_set_tuning(LCM_TUNE_ONLINE_SPLIT, 0) forces the unsplit path; the full-size sharded test covers it through lcm_all_vs_all)."""
    fs = pkg.synth.make_frames(700, 600, seed=57, ragged=True, dup_frac=0.2)
    fs.counts[50] = 0
    gap = 1
    matcher.set_params(min_gap=gap)
    try:
        fill(matcher, fs)
        p = oracle.default_params(min_gap=gap)
        for cur in (100, 255, 257, 500, 699):
            scores, ids = matcher.query_scores(fs.frame(cur), int(fs.ids[cur]))
            assert len(scores) == cur                                   # ids == positions, gap 1
            want, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [cur] * cur, list(range(cur)), p, n_threads=8)
            np.testing.assert_array_equal(scores, want)
            cands = matcher.detect_loops(int(fs.ids[cur]))              # stored-frame query through the same path
            keep = [i for i in range(cur) if oracle.loop_test(int(want[i]["good_count"]), int(fs.counts[cur]), int(fs.counts[i]), p)[0]]
            assert cands["matched_frame_id"].tolist() == [int(fs.ids[i]) for i in keep]
        # an empty query frame through the split path's guard (nq <= 512 never splits) and a full-size one
        big = pkg.synth.make_frames(4, 2000, seed=3)
        scores, _ = matcher.query_scores(big.frame(0), 120)
        want = [oracle.pair_score(big.frame(0), fs.frame(i), p) for i in (0, 50, 119)]
        assert [scores[0], scores[50], scores[119]] == want
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_async_submit_collect_pipeline(matcher, oracle, pkg):
    """Streaming with up to 4 queries in flight: results equal the synchronous path and the batch oracle."""
    fs = pkg.synth.make_frames(60, 700, seed=77, ragged=True, dup_frac=0.3)
    gap = 2
    matcher.set_params(min_gap=gap)
    try:
        matcher.clear()
        got = [None] * fs.n_frames
        pending = []
        for f in range(fs.n_frames):
            pending.append((matcher.query_submit(fs.frame(f), int(fs.ids[f])), f))
            matcher.append(int(fs.ids[f]), fs.frame(f))
            if len(pending) == 4:                               # the ring is full: a 5th submit must be refused
                with pytest.raises(pkg.LcmError) as e:
                    matcher.query_submit(fs.frame(f), int(fs.ids[f]))
                assert e.value.code == -4
                t, g = pending.pop(0)
                got[g], ids = matcher.query_collect(t)
                assert ids.tolist() == [int(fs.ids[i]) for i in range(fs.n_frames) if fs.ids[g] - fs.ids[i] >= gap]
        for t, g in pending:
            got[g], _ = matcher.query_collect(t)
        want, _ = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap))
        np.testing.assert_array_equal(np.concatenate(got), want)
        with pytest.raises(pkg.LcmError):
            matcher.query_collect(0)                            # nothing in flight on that ticket
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_create_use_destroy_cycles(pkg):
    """Handles come and go (every scratch buffer, pinned ring, stream and event is released) and two can coexist."""
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (700, 32), dtype=np.uint8) for _ in range(12)]
    ref = None
    for cycle in range(25):
        p = pkg.default_params()
        p.min_gap = 1
        a = pkg.Matcher(p)
        b = pkg.Matcher(p) if cycle % 5 == 0 else None
        for i, f in enumerate(frames):
            a.append(i, f)
            if b is not None:
                b.append(i, f)
        s, _ = a.query_scores(frames[3], 50)
        if b is not None:
            s2, _ = b.query_scores(frames[3], 50)
            np.testing.assert_array_equal(s, s2)
            b.close()
        idx, d = a.match_pair(frames[0], frames[1])
        t = a.query_submit(frames[5], 60)                     # destroyed with a query still in flight
        if ref is None:
            ref = (s.copy(), idx.copy(), d.copy())
        np.testing.assert_array_equal(s, ref[0])
        np.testing.assert_array_equal(idx, ref[1])
        a.close()


def gpu_all_vs_all_argmin(m, pkg):
    n, offs = m.all_vs_all_plan()
    d_scores = m.dev_alloc(max(n, 1) * 8)
    d_sums = m.dev_alloc(max(n, 1) * 4)
    try:
        assert m.all_vs_all_argmin(d_scores, n, d_sums) == n
        scores = np.zeros(max(n, 1), pkg.capi.SCORE_DTYPE)
        sums = np.zeros(max(n, 1), np.uint32)
        m.sync()
        m.dev_download(d_scores, scores)
        m.dev_download(d_sums, sums)
    finally:
        m.dev_free(d_scores); m.dev_free(d_sums)
    return scores[:n], sums[:n], offs


@pytest.mark.parametrize("n_frames,max_desc,gap,seed", [(26, 700, 3, 41), (12, 2000, 2, 8), (40, 130, 1, 17), (18, 1100, 2, 5)])
def test_argmin_kernel_index_checksum(matcher, oracle, pkg, n_frames, max_desc, gap, seed):
    """lcm_all_vs_all_argmin: the first-minimum TRAIN INDEX of every query row is found by the bulk kernel itself
    (16-row group keys + re-scan of the winning group); the per-pair checksum of the good matches' indices must equal
    the oracle's.  Heavy ties: duplicated rows inside a frame, across frames, across 16-row group boundaries."""
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=seed, ragged=True, dup_frac=0.5)
    fs.counts[4] = 0
    fs.counts[9] = 1
    n7 = int(fs.counts[7])
    fs.rows[7, : n7 // 2] = fs.rows[3, : n7 // 2]                     # frame 7 duplicates half of frame 3 ...
    fs.rows[3, 16:48] = fs.rows[3, 0:32]                              # ... which repeats its own rows 16 further on
    fs.rows[6, 15] = fs.rows[6, 16] = fs.rows[6, 31] = fs.rows[6, 32]  # ties straddling group boundaries
    matcher.set_params(min_gap=gap)
    try:
        fill(matcher, fs)
        scores, sums, offs = gpu_all_vs_all_argmin(matcher, pkg)
        pq, pt = [], []
        for c in range(n_frames):
            for t in range(n_frames):
                if fs.ids[c] - fs.ids[t] >= gap:
                    pq.append(c); pt.append(t)
        want, wsums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, oracle.default_params(min_gap=gap), n_threads=8)
        np.testing.assert_array_equal(scores, want)
        np.testing.assert_array_equal(sums, wsums)
        rng = np.random.default_rng(seed)
        for k in rng.choice(len(pq), 6, replace=False):               # and straight from the scalar oracle's match list
            assert int(sums[k]) == oracle.index_sum(fs.frame(pq[k]), fs.frame(pt[k]), oracle.default_params(min_gap=gap))
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("max_desc,n_frames,batch", [(700, 90, 8), (2000, 40, 16), (300, 60, 5), (1300, 50, 3)])
def test_micro_batched_online_queries_equal_one_by_one(matcher, oracle, pkg, max_desc, n_frames, batch):
    """lcm_query_submit_batch: B frames scored by one launch (split and unsplit regimes, ragged frames, an empty frame,
    queries with different eligibility) == the records of B single submissions == the batch oracle."""
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=88 + batch, ragged=True, dup_frac=0.3)
    fs.counts[11] = 0
    gap = batch                                                   # the exactness condition: id span of a batch <= gap
    matcher.set_params(min_gap=gap)
    try:
        matcher.clear()
        got = []
        for f0 in range(0, n_frames, batch):
            fr = list(range(f0, min(f0 + batch, n_frames)))
            t = matcher.query_submit_batch([fs.frame(f) for f in fr], [int(fs.ids[f]) for f in fr])
            with pytest.raises(pkg.LcmError):
                matcher.query_collect(t)                          # a batch ticket is not a single-query ticket
            scores, offs = matcher.query_collect_batch(t)
            assert len(offs) == len(fr) + 1 and int(offs[-1]) == len(scores)
            for k, f in enumerate(fr):
                part = scores[int(offs[k]): int(offs[k + 1])]
                e = sum(1 for i in range(f0) if fs.ids[f] - fs.ids[i] >= gap)
                assert len(part) == e, (f, len(part), e)
                got.append((f, part.copy()))
            for f in fr:
                matcher.append(int(fs.ids[f]), fs.frame(f))
        want, woffs = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=gap))
        for f, part in got:                                       # the empty frame 11 yields "empty pair" records too
            np.testing.assert_array_equal(part, want[int(woffs[f]): int(woffs[f + 1])])
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_batched_match_lists_equal_match_features_per_pair(matcher, oracle, pkg):
    """lcm_match_stored_batch / lcm_match_query_batch (one launch for all pairs, device-side segment fold) == the
    oracle's matchFeatures per pair: ragged frames, an empty frame, a 1-row frame, duplicated rows (ties across
    segment boundaries), a pair of a frame with itself."""
    fs = pkg.synth.make_frames(30, 1100, seed=61, ragged=True, dup_frac=0.5)
    fs.counts[4] = 0
    fs.counts[9] = 1
    fs.rows[7, 100:400] = fs.rows[7, 500:800]                       # ties 400 rows apart inside a train frame
    fs.rows[12, :300] = fs.rows[7, 500:800]
    p = oracle.default_params()
    try:
        fill(matcher, fs)
        pairs = [(20, 2), (12, 7), (7, 12), (25, 4), (4, 25), (9, 11), (11, 9), (7, 7), (29, 0), (12, 7)]
        lists, mins = matcher.match_stored_batch([(int(fs.ids[a]), int(fs.ids[b])) for a, b in pairs])
        assert len(lists) == len(pairs)
        for (a, b), got, md in zip(pairs, lists, mins):
            want, wmd = oracle.match_features(fs.frame(a), fs.frame(b), p)
            for f in ("query_idx", "train_idx", "img_idx", "distance"):
                np.testing.assert_array_equal(got[f], want[f], err_msg=f"pair {(a, b)} field {f}")
            assert int(md) == (wmd if len(want) else -1)
            one, omd = matcher.match_stored(int(fs.ids[a]), int(fs.ids[b]))      # the single-pair entry agrees
            np.testing.assert_array_equal(one, got)
        # one host query (not stored) against many stored frames
        q = pkg.synth.make_frames(1, 1500, seed=62).frame(0).copy()
        q[:200] = fs.rows[7, 500:700]
        trains = [7, 12, 4, 9, 0, 29]
        lists, mins = matcher.match_query_batch(q, [int(fs.ids[t]) for t in trains])
        for t, got, md in zip(trains, lists, mins):
            want, wmd = oracle.match_features(q, fs.frame(t), p)
            np.testing.assert_array_equal(got, want.astype(got.dtype))
            assert int(md) == (wmd if len(want) else -1)
        # empty batch, unknown id, too small a buffer
        assert matcher.match_stored_batch([])[0] == []
        with pytest.raises(pkg.LcmError) as e:
            matcher.match_stored_batch([(int(fs.ids[1]), 9999)])
        assert e.value.code == -6
        with pytest.raises(pkg.LcmError) as e:
            matcher.match_stored_batch([(int(fs.ids[20]), int(fs.ids[2]))], cap=3)
        assert e.value.code == -4
    finally:
        matcher.clear()


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("n_frames,max_desc,gap,seed", [(30, 2000, 2, 3), (50, 700, 1, 4), (64, 90, 3, 5), (20, 1300, 2, 6)])
def test_matrix_core_variant_is_bit_exact(matcher, oracle, pkg, n_frames, max_desc, gap, seed, variant):
    """lcm_set_kernel_variant(4 = int8 / 5 = fp4 matrix instruction): self and external query sets, ragged frames (row counts not multiples of 32 / 256),
    an empty and a 1-row frame, exact duplicates (distance 0), and the fused loop test on top of it."""
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=seed, ragged=True, dup_frac=0.4)
    fs.counts[4] = 0
    fs.counts[9] = 1
    fs.rows[7, :20] = fs.rows[3, :20]
    matcher.set_params(min_gap=gap)
    matcher.set_kernel_variant(variant)
    d_rows = matcher.dev_alloc(fs.rows.nbytes)
    d_counts = matcher.dev_alloc(fs.counts.nbytes)
    try:
        fill(matcher, fs)
        p = oracle.default_params(min_gap=gap)
        want, woffs = fast_all_vs_all(oracle, fs, p)
        got, offs = gpu_all_vs_all(matcher)
        np.testing.assert_array_equal(offs.astype(np.int64), woffs)
        np.testing.assert_array_equal(got, want)
        # the argmin form on the matrix cores: first tile that reaches the best dot product + exact re-scan of that tile
        n = len(want)
        pq = [c for c in range(n_frames) for _ in range(int(woffs[c + 1] - woffs[c]))]
        pt = [t for c in range(n_frames) for t in range(int(woffs[c + 1] - woffs[c]))]
        _, wsums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=8)
        d, ds = matcher.dev_alloc(n * 8), matcher.dev_alloc(n * 4)
        got_a, sums = np.zeros(n, want.dtype), np.zeros(n, np.uint32)
        matcher.all_vs_all_argmin(d, n, ds)
        assert matcher.launch_info().route == pkg.capi.ROUTE_MATRIX
        matcher.sync(); matcher.dev_download(d, got_a); matcher.dev_download(ds, sums)
        matcher.dev_free(d); matcher.dev_free(ds)
        np.testing.assert_array_equal(got_a, want)
        np.testing.assert_array_equal(sums, wsums)
        matcher.dev_upload(d_rows, fs.rows); matcher.dev_upload(d_counts, fs.counts)
        got2, _ = gpu_all_vs_all(matcher, q_ids=fs.ids, d_rows=d_rows, d_counts=d_counts, stride=fs.stride_rows)
        np.testing.assert_array_equal(got2, want)
        matcher.append(int(fs.ids[-1]) + 1, fs.frame(1))         # the operand image follows the database
        got3, offs3 = gpu_all_vs_all(matcher)
        np.testing.assert_array_equal(got3[: len(want)], want)
        cands, npairs = matcher.all_vs_all_loops(cap=len(got3))
        keep = []
        ids3 = list(fs.ids) + [int(fs.ids[-1]) + 1]
        cnt3 = list(fs.counts) + [int(fs.counts[1])]
        for c in range(len(ids3)):
            for k in range(int(offs3[c]), int(offs3[c + 1])):
                t = k - int(offs3[c])
                if oracle.loop_test(int(got3[k]["good_count"]), cnt3[c], cnt3[t], p)[0]:
                    keep.append((ids3[c], ids3[t], int(got3[k]["good_count"])))
        assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"])) for r in cands] == keep
        # online queries run on the matrix cores too (single query, stored-frame query, micro-batch); pair mode keeps
        # the vector-ALU kernels
        sc, _ = matcher.query_scores(fs.frame(n_frames - 1), int(fs.ids[-1]) + 50)
        assert len(sc) == n_frames + 1
        fs2_rows = np.concatenate([fs.rows, fs.rows[1:2]]); fs2_counts = np.concatenate([fs.counts, fs.counts[1:2]])
        wq, _, _ = oracle.fast_score_pairs(fs2_rows, fs2_counts, [n_frames - 1] * (n_frames + 1), list(range(n_frames + 1)), p, n_threads=8)
        np.testing.assert_array_equal(sc, wq)
        cur = n_frames // 2
        c_stored = matcher.detect_loops(int(fs.ids[cur]))
        c_host = matcher.detect_loops(int(fs.ids[cur]), fs.frame(cur))
        np.testing.assert_array_equal(c_stored, c_host)
        want_c = oracle.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)
        np.testing.assert_array_equal(c_stored["matched_frame_id"], want_c["matched_frame_id"])
        np.testing.assert_array_equal(c_stored["num_matches"], want_c["num_matches"])
        qb = [fs.frame(4), fs.frame(9), fs.frame(7), fs.frame(n_frames - 1)]          # empty, 1 row, duplicates, full
        t = matcher.query_submit_batch(qb, [int(fs.ids[-1]) + 60 + k for k in range(4)])
        bs, boffs = matcher.query_collect_batch(t)
        for k, f in enumerate((4, 9, 7, n_frames - 1)):
            wk, _, _ = oracle.fast_score_pairs(fs2_rows, fs2_counts, [f] * (n_frames + 1), list(range(n_frames + 1)), p, n_threads=8)
            np.testing.assert_array_equal(bs[int(boffs[k]): int(boffs[k + 1])], wk)
        idx, d = matcher.match_pair(fs.frame(2), fs.frame(3))
        oi, od = oracle.bf_match(fs.frame(2), fs.frame(3))
        np.testing.assert_array_equal(idx, oi)
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_kernel_variant(0)
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("own_streams", [1, 0])
def test_four_tickets_in_flight_with_appends_between_them(matcher, oracle, pkg, own_streams):
    """LCM_TUNE_ONLINE_STREAMS: every query slot enqueues on its own stream (default) so consecutive online launches
    overlap.  Four tickets in flight — single queries and micro-batches — with host appends AND a device append between
    the submits (each query must see exactly the frames stored before it), collected out of order: records == oracle,
    and the same with everything on the handle's stream."""
    fs = pkg.synth.make_frames(60, 1400, seed=77, ragged=True, dup_frac=0.3)
    gap = 3
    matcher.set_params(min_gap=gap)
    matcher.set_tuning(pkg.capi.TUNE_ONLINE_STREAMS, own_streams)
    p = oracle.default_params(min_gap=gap)
    d_frame = matcher.dev_alloc(fs.stride_rows * 32)
    try:
        matcher.clear()
        matcher.reserve(fs.n_frames, fs.stride_rows)
        for f in range(40):
            matcher.append(int(fs.ids[f]), fs.frame(f))
        want, woffs = fast_all_vs_all(oracle, fs, p)

        def rows_of(f):
            return want[int(woffs[f]): int(woffs[f + 1])]

        tickets = []
        tickets.append(("one", [40], matcher.query_submit(fs.frame(40), int(fs.ids[40]))))
        matcher.append(int(fs.ids[40]), fs.frame(40))                             # host append: copy stream
        tickets.append(("batch", [41, 42], matcher.query_submit_batch([fs.frame(41), fs.frame(42)], [int(fs.ids[41]), int(fs.ids[42])])))
        matcher.dev_upload(d_frame, np.ascontiguousarray(fs.rows[41]))
        matcher.append_device(int(fs.ids[41]), d_frame, int(fs.counts[41]))       # device append: the handle's stream
        matcher.append(int(fs.ids[42]), fs.frame(42))
        tickets.append(("one", [45], matcher.query_submit(fs.frame(45), int(fs.ids[45]))))    # sees 41 and 42 (gap 3)
        tickets.append(("batch", [46, 47, 48], matcher.query_submit_batch([fs.frame(f) for f in (46, 47, 48)], [int(fs.ids[f]) for f in (46, 47, 48)])))
        with pytest.raises(pkg.capi.LcmError):                                    # a fifth would need a fifth slot
            matcher.query_submit(fs.frame(49), int(fs.ids[49]))
        for kind, frames, t in reversed(tickets):
            if kind == "one":
                sc, _ = matcher.query_collect(t)
                np.testing.assert_array_equal(sc, rows_of(frames[0])[: len(sc)])
                assert len(sc) == min(len(rows_of(frames[0])), 43 if frames[0] == 45 else 40)
            else:
                sc, boffs = matcher.query_collect_batch(t)
                for j, f in enumerate(frames):
                    got = sc[int(boffs[j]): int(boffs[j + 1])]
                    np.testing.assert_array_equal(got, rows_of(f)[: len(got)])
                    assert len(got) > 0
        # frame 45 was eligible for 41 and 42 only if they had landed before its launch: ids 41, 42 <= 45 - 3
        matcher.sync()
    finally:
        matcher.dev_free(d_frame)
        matcher.set_tuning(pkg.capi.TUNE_ONLINE_STREAMS, 1)
        matcher.set_params(min_gap=30)
        matcher.clear()
