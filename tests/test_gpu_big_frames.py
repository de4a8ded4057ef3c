"""Query frames above 2048 rows (ORB with nfeatures > 2048): one workgroup's registers hold 2048 query rows, so these
go through the packed bulk route — a frame spans several 2048-row columns, k_finalize_bulk folds its rows in a loop —
from every entry point that takes a query frame: bulk (self / external, records + index checksums), online (single,
micro-batch, stored-frame detectLoops), the host class, the group.  cross_check keeps the 2048-row limit and says so."""
import numpy as np
import pytest

from conftest import fast_detect_loops

pytestmark = pytest.mark.gpu

ROWS = [3000, 2500, 0, 2049, 700, 4096, 1, 2048, 3333, 2600, 5000, 1500, 2100, 64]


def _frames(pkg, seed=41):
    n = len(ROWS)
    fs = pkg.synth.make_frames(n, max(ROWS), seed=seed, dup_frac=0.3)
    fs.counts[:] = np.array(ROWS, np.int32)
    fs.rows[5, :30] = fs.rows[0, :30]                       # exact duplicates across two big frames: distance 0, first-minimum ties
    fs.rows[8, 2500:2530] = fs.rows[0, :30]                 # ... and beyond the 2048-row seam of the query frame
    return fs


def _want(oracle, fs, gap, cross=0, check_scalar=True):
    p = oracle.default_params(min_gap=gap, cross_check=cross)
    pq, pt, offs = [], [], [0]
    for c in range(fs.n_frames):
        for i in range(fs.n_frames):
            if fs.ids[c] - fs.ids[i] >= max(gap, 1):
                pq.append(c); pt.append(i)
        offs.append(len(pq))
    sc, sums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=8)
    if check_scalar:                                       # the tuned path against the scalar restatement on these shapes
        for c, t in ((8, 0), (12, 5), (3, 1)):                 # (~1 s per 2000 x 2000 pair: three pairs only)
            k = int(offs[c]) + t
            assert (pq[k], pt[k]) == (c, t) and sc[k] == oracle.pair_score(fs.frame(c), fs.frame(t), p)
    return sc, sums, np.array(offs, np.int64), p


def test_bulk_and_online_with_query_frames_above_2048_rows(matcher, oracle, pkg):
    fs = _frames(pkg)
    gap = 2
    matcher.set_params(min_gap=gap)
    d_rows, d_counts = matcher.dev_alloc(fs.rows.nbytes), matcher.dev_alloc(fs.counts.nbytes)
    try:
        matcher.clear()
        for f in range(fs.n_frames):
            matcher.append(int(fs.ids[f]), fs.frame(f))
        want, wsums, woffs, p = _want(oracle, fs, gap)
        n, offs = matcher.all_vs_all_plan()
        assert n == len(want) and np.array_equal(offs.astype(np.int64), woffs)
        d, ds = matcher.dev_alloc(n * 8), matcher.dev_alloc(n * 4)
        got, sums = np.zeros(n, want.dtype), np.zeros(n, np.uint32)
        # self search: packed route whatever the size (the only one that holds such frames)
        matcher.all_vs_all(d, n)
        assert matcher.launch_info().route == pkg.capi.ROUTE_PACKED
        matcher.sync(); matcher.dev_download(d, got)
        np.testing.assert_array_equal(got, want)
        matcher.all_vs_all_argmin(d, n, ds)
        matcher.sync(); matcher.dev_download(d, got); matcher.dev_download(ds, sums)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(sums, wsums)
        # external query set (the frames again, from a caller's buffer), and the opt-in matrix-core variants falling
        # back to the vector-ALU packed route for these shapes
        matcher.dev_upload(d_rows, fs.rows); matcher.dev_upload(d_counts, fs.counts)
        for variant in (0, 5):
            matcher.set_kernel_variant(variant)
            got[:] = 0
            matcher.all_vs_all(d, n, d_query_rows=d_rows, d_query_counts=d_counts, q_ids=fs.ids, q_stride_rows=fs.stride_rows)
            matcher.sync(); matcher.dev_download(d, got)
            np.testing.assert_array_equal(got, want)
        matcher.set_kernel_variant(0)
        # the fused loop test sits on top
        cands, npairs = matcher.all_vs_all_loops(cap=n)
        keep = [(int(fs.ids[c]), int(fs.ids[k - int(woffs[c])]), int(want[k]["good_count"]))
                for c in range(fs.n_frames) for k in range(int(woffs[c]), int(woffs[c + 1]))
                if oracle.loop_test(int(want[k]["good_count"]), int(fs.counts[c]), int(fs.counts[k - int(woffs[c])]), p)[0]]
        assert npairs == n and [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"])) for r in cands] == keep
        matcher.dev_free(d); matcher.dev_free(ds)
        # lcm_all_vs_all with the packed route switched off cannot serve them, and says so
        matcher.set_tuning(pkg.capi.TUNE_PACKED, 0)
        with pytest.raises(pkg.capi.LcmError) as e:
            matcher.all_vs_all_plan()
        assert e.value.code == pkg.capi.ERR_CAPACITY
        matcher.set_tuning(pkg.capi.TUNE_PACKED, -1)

        # ---- online: single query (host rows), stored-frame detectLoops, micro-batch with a mix of sizes
        last = fs.n_frames - 1
        for c in (0, 5, 10, 3, 12):                            # 3000, 4096, 5000, 2049, 2100 rows
            qid = int(fs.ids[last]) + gap                       # eligible for every stored frame
            sc, sid = matcher.query_scores(fs.frame(c), qid)
            wq, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [c] * fs.n_frames, list(range(fs.n_frames)), p, n_threads=8)
            np.testing.assert_array_equal(sc, wq)
            np.testing.assert_array_equal(sid, fs.ids)
        for cur in (10, 8, 5):
            got_c = matcher.detect_loops(int(fs.ids[cur]))                         # the stored frame is the query
            got_h = matcher.detect_loops(int(fs.ids[cur]), fs.frame(cur))        # the same frame as host rows
            want_c = fast_detect_loops(oracle, fs, cur, p)
            for g in (got_c, got_h):
                assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"])) for r in g] == want_c
        batch = [0, 4, 3, 10, 2]                                # 3000, 700, 2049, 5000, 0 rows in one submit
        t = matcher.query_submit_batch([fs.frame(c) for c in batch], [int(fs.ids[last]) + gap + k for k in range(len(batch))])
        bs, boffs = matcher.query_collect_batch(t)
        for k, c in enumerate(batch):
            wq, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [c] * fs.n_frames, list(range(fs.n_frames)), p, n_threads=8)
            np.testing.assert_array_equal(bs[int(boffs[k]): int(boffs[k + 1])], wq)
        # pair mode never had the limit
        idx, dd = matcher.match_pair(fs.frame(10), fs.frame(5))
        oi, od = oracle.bf_match(fs.frame(10), fs.frame(5))
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(dd.astype(np.int32), od)
        # cross_check keeps the 2048-row limit on query frames: a clear error, not a wrong answer
        matcher.set_params(cross_check=1)
        with pytest.raises(pkg.capi.LcmError) as e:
            matcher.query_scores(fs.frame(0), int(fs.ids[last]) + gap)
        assert e.value.code == pkg.capi.ERR_CAPACITY
        matcher.set_params(cross_check=0)
        sc, _ = matcher.query_scores(fs.frame(4), int(fs.ids[last]) + gap)        # the handle is still usable
        assert len(sc) == fs.n_frames
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_tuning(pkg.capi.TUNE_PACKED, -1)
        matcher.set_kernel_variant(0)
        matcher.set_params(min_gap=30, cross_check=0)
        matcher.clear()


def test_host_class_and_group_with_big_frames(pkg, oracle):
    fs = _frames(pkg, seed=43)
    gap, thr = 2, 0.01
    p = oracle.default_params(min_gap=gap, sim_threshold=thr)
    sys_ = pkg.LoopClosingSystem(thr, gap)
    try:
        for f in range(fs.n_frames):
            sys_.processFrame(fs.frame(f), int(fs.ids[f]))
        got = sys_.getLoopClosures()
        want = [t for c in range(fs.n_frames) for t in fast_detect_loops(oracle, fs, c, p)]
        assert len(want) > 0
        assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"])) for r in got] == want
    finally:
        sys_.close()
    gp = pkg.default_params()
    gp.min_gap = gap
    with pkg.Group(gp, n_devices=1) as g:
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
        merged, offs = g.all_vs_all()
        wsc, _, woffs, _ = _want(oracle, fs, gap, check_scalar=False)
        np.testing.assert_array_equal(merged, wsc)
        np.testing.assert_array_equal(np.asarray(offs, np.int64), woffs)
        sc, _ = g.query_scores(fs.frame(10), int(fs.ids[-1]) + gap)
        wq, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [10] * fs.n_frames, list(range(fs.n_frames)), p, n_threads=8)
        np.testing.assert_array_equal(sc, wq)
