"""lcm_params.cross_check (BFMatcher crossCheck = true, both upstream behaviours) through every entry point, against the
oracle's restatement (tests/test_oracle_cross_check.py pins that one): ragged + tie-heavy inputs, empty frames."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(pkg, seed, n_frames=22, max_desc=900):
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=seed, ragged=True, dup_frac=0.6)
    fs.counts[4] = 0
    fs.counts[9] = 1
    n7 = int(fs.counts[7])
    fs.rows[7, : n7 // 3] = fs.rows[3, : n7 // 3]            # frame 7 repeats rows of frame 3 ...
    fs.rows[7, n7 // 3: 2 * (n7 // 3)] = fs.rows[3, : n7 // 3]   # ... twice: two query rows tie for one train row
    fs.rows[12, :50] = fs.rows[12, 50:100]                   # ties inside a frame
    return fs


@pytest.mark.parametrize("mode", [1, 2])
def test_pair_mode_cross_check(matcher, oracle, pkg, mode):
    fs = _frames(pkg, 40 + mode)
    matcher.set_params(cross_check=mode)
    p = oracle.default_params(cross_check=mode)
    try:
        for a, b in [(7, 3), (3, 7), (12, 12), (20, 2), (9, 11), (11, 9), (4, 5), (5, 4)]:
            idx, d = matcher.match_pair(fs.frame(a), fs.frame(b))
            oi, od = oracle.bf_match_cross(fs.frame(a), fs.frame(b), mode)
            # lcm_match_pair keeps per-query arrays; unmatched rows are -1 / 0xFFFF and n_matches counts the matched ones
            if len(oi) == 0 or len(fs.frame(b)) == 0:
                assert len(idx) == 0
                continue
            full_idx = np.full(len(oi), -1, np.int32)
            got, n_m = _match_pair_full(matcher, pkg, fs.frame(a), fs.frame(b))
            np.testing.assert_array_equal(got[0], oi)
            np.testing.assert_array_equal(got[1][oi >= 0].astype(np.int32), od[oi >= 0])
            assert n_m == int((oi >= 0).sum())
            m, md = matcher.match_features(fs.frame(a), fs.frame(b))
            om, omd = oracle.match_features(fs.frame(a), fs.frame(b), p)
            np.testing.assert_array_equal(m, om.astype(m.dtype))
            assert md == omd
        # stored frames, single and batched
        matcher.clear()
        for f in range(fs.n_frames):
            matcher.append(int(fs.ids[f]), fs.frame(f))
        pairs = [(7, 3), (3, 7), (12, 12), (20, 2), (9, 11), (4, 5)]
        lists, mins = matcher.match_stored_batch(pairs)
        for (a, b), got, md in zip(pairs, lists, mins):
            om, omd = oracle.match_features(fs.frame(a), fs.frame(b), p)
            np.testing.assert_array_equal(got, om.astype(got.dtype))
            assert int(md) == (omd if len(om) else -1)
            one, _ = matcher.match_stored(a, b)
            np.testing.assert_array_equal(one, got)
        q = fs.frame(7).copy()
        lists, _ = matcher.match_query_batch(q, [3, 12, 4, 9])
        for t, got in zip([3, 12, 4, 9], lists):
            om, _ = oracle.match_features(q, fs.frame(t), p)
            np.testing.assert_array_equal(got, om.astype(got.dtype))
    finally:
        matcher.set_params(cross_check=0)
        matcher.clear()


def _match_pair_full(m, pkg, q, t):
    """lcm_match_pair's raw per-query arrays (capi.match_pair slices them to n_matches, which cross_check makes < nq)."""
    import ctypes as C
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    idx = np.empty(q.shape[0], np.int32); dist = np.empty(q.shape[0], np.uint16)
    n = C.c_int32(0)
    rc = m._lib.lcm_match_pair(m._h, q.ctypes.data_as(C.c_void_p), q.shape[0], t.ctypes.data_as(C.c_void_p), t.shape[0],
                               idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p), C.byref(n))
    assert rc == 0
    return (idx, dist), n.value


@pytest.mark.parametrize("mode", [1, 2])
def test_loop_search_cross_check(matcher, oracle, pkg, mode):
    """lcm_all_vs_all (self and external query set), lcm_all_vs_all_argmin's index checksum, lcm_query_scores,
    lcm_query_submit_batch, lcm_detect_loops and the fused lcm_all_vs_all_loops under cross_check."""
    fs = _frames(pkg, 50 + mode)
    gap = 3
    matcher.set_params(cross_check=mode, min_gap=gap)
    p = oracle.default_params(cross_check=mode, min_gap=gap)
    d_rows = matcher.dev_alloc(fs.rows.nbytes)
    d_counts = matcher.dev_alloc(fs.counts.nbytes)
    try:
        matcher.clear()
        for f in range(fs.n_frames):
            matcher.append(int(fs.ids[f]), fs.frame(f))
        pq, pt = [], []
        for c in range(fs.n_frames):
            for t in range(fs.n_frames):
                if fs.ids[c] - fs.ids[t] >= gap:
                    pq.append(c); pt.append(t)
        want, wsums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=8)
        plain, _ = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, oracle.default_params(min_gap=gap), n_threads=8)
        assert (want["good_count"] != plain["good_count"]).sum() > len(pq) // 2       # the mode really changes the answer

        n, offs = matcher.all_vs_all_plan()
        assert n == len(pq)
        d_sc, d_su = matcher.dev_alloc(n * 8), matcher.dev_alloc(n * 4)
        got, sums = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
        matcher.all_vs_all_argmin(d_sc, n, d_su)
        matcher.sync()
        matcher.dev_download(d_sc, got); matcher.dev_download(d_su, sums)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(sums, wsums)
        # external query set (unpadded caller rows -> padded copy inside the library)
        matcher.dev_upload(d_rows, fs.rows); matcher.dev_upload(d_counts, fs.counts)
        got2 = np.zeros(n, pkg.capi.SCORE_DTYPE)
        matcher.all_vs_all(d_sc, n, d_rows, d_counts, fs.ids, fs.stride_rows)
        matcher.sync()
        matcher.dev_download(d_sc, got2)
        np.testing.assert_array_equal(got2, want)
        matcher.dev_free(d_sc); matcher.dev_free(d_su)
        # online: single query, micro-batch, detect_loops (host query and stored frame)
        for cur in (7, 12, 21, 4):
            sc, ids = matcher.query_scores(fs.frame(cur), int(fs.ids[cur]))
            np.testing.assert_array_equal(sc, want[int(offs[cur]): int(offs[cur + 1])])
        t = matcher.query_submit_batch([fs.frame(f) for f in (19, 20, 21)], [int(fs.ids[f]) for f in (19, 20, 21)])
        sc, boffs = matcher.query_collect_batch(t)
        for k, f in enumerate((19, 20, 21)):
            np.testing.assert_array_equal(sc[int(boffs[k]): int(boffs[k + 1])], want[int(offs[f]): int(offs[f + 1])])
        for cur in (12, 21):
            want_c = oracle.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)
            for got_c in (matcher.detect_loops(int(fs.ids[cur])), matcher.detect_loops(int(fs.ids[cur]), fs.frame(cur))):
                for f in ("current_frame_id", "matched_frame_id", "num_matches", "similarity_score"):
                    np.testing.assert_array_equal(got_c[f], want_c[f])
        cands, npairs = matcher.all_vs_all_loops(cap=n)
        assert npairs == n
        keep = [k for k in range(n) if oracle.loop_test(int(want[k]["good_count"]), int(fs.counts[pq[k]]), int(fs.counts[pt[k]]), p)[0]]
        assert cands["current_frame_id"].tolist() == [int(fs.ids[pq[k]]) for k in keep]
        assert cands["matched_frame_id"].tolist() == [int(fs.ids[pt[k]]) for k in keep]
        assert cands["num_matches"].tolist() == [int(want[k]["good_count"]) for k in keep]
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_params(cross_check=0, min_gap=30)
        matcher.clear()


def test_cross_check_parameter_is_validated(matcher, pkg):
    with pytest.raises(pkg.LcmError) as e:
        matcher.set_params(cross_check=3)
    assert e.value.code == -1
    assert matcher.params.cross_check == 0
