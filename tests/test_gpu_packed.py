"""Packed bulk mode (LCM_TUNE_PACKED; ScoreArgs::pk_* in csrc/lcm_kernels.h): the query rows of consecutive frames share
full 2048-row workgroups.  Forced on, forced off and automatic must give the same bytes, and those == the oracle —
records and per-pair index checksums, self and external query sets, ragged / empty / tiny frames, any gap."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fill(m, fs):
    m.clear()
    for f in range(fs.n_frames):
        m.append(int(fs.ids[f]), fs.frame(f))


def _want(oracle, fs, q_ids, q_rows, q_counts, params):
    """Oracle scores + index checksums of `query c against every stored frame i with q_ids[c] - ids[i] >= gap`."""
    gap = max(int(params.min_gap), 1)
    n_db = fs.n_frames
    stride = max(fs.rows.shape[1], q_rows.shape[1])
    rows = np.zeros((n_db + len(q_ids), stride, 32), np.uint8)
    rows[:n_db, : fs.rows.shape[1]] = fs.rows
    rows[n_db:, : q_rows.shape[1]] = q_rows
    counts = np.concatenate([fs.counts, q_counts]).astype(np.int32)
    pq, pt, offs = [], [], [0]
    for c, qid in enumerate(q_ids):
        for i in range(n_db):
            if int(qid) - int(fs.ids[i]) >= gap:
                pq.append(n_db + c); pt.append(i)
        offs.append(len(pq))
    sc, sums = oracle.fast_score_pairs_idx(rows, counts, pq, pt, params, n_threads=8)
    return sc, sums, np.array(offs, np.int64)


def _run(m, pkg, n, mode, ext=None):
    m.set_tuning(pkg.capi.TUNE_PACKED, mode)
    kw = {} if ext is None else ext
    d, ds = m.dev_alloc(max(n, 1) * 8), m.dev_alloc(max(n, 1) * 4)
    try:
        got, got2, sums = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
        m.all_vs_all(d, n, **kw)
        launches = m.launch_info().route if n else -1
        m.sync(); m.dev_download(d, got)
        m.all_vs_all_argmin(d, n, ds, **kw)
        m.sync(); m.dev_download(d, got2); m.dev_download(ds, sums)
        return got, got2, sums, launches
    finally:
        m.dev_free(d); m.dev_free(ds)
        m.set_tuning(pkg.capi.TUNE_PACKED, -1)


@pytest.mark.parametrize("n_frames,max_desc,gap,seed", [(40, 700, 1, 11), (24, 2000, 2, 12), (60, 90, 3, 13), (33, 1300, 0, 14), (9, 2048, 1, 15)])
def test_packed_self_search_is_bit_exact(matcher, oracle, pkg, n_frames, max_desc, gap, seed):
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=seed, ragged=True, dup_frac=0.4)
    fs.counts[min(4, n_frames - 1)] = 0                  # an empty frame: no rows in the packed space, records still due
    fs.counts[min(7, n_frames - 1)] = 1
    fs.rows[5, :20] = fs.rows[3, :20]                    # exact duplicates: distance 0, ties on the first minimum
    if max_desc == 2048:
        fs.counts[2] = 2048                              # a frame that fills a column exactly
    matcher.set_params(min_gap=gap)
    try:
        _fill(matcher, fs)
        p = oracle.default_params(min_gap=gap)
        want, wsums, woffs = _want(oracle, fs, fs.ids, fs.rows, fs.counts, p)
        n, offs = matcher.all_vs_all_plan()
        assert n == len(want) and np.array_equal(offs.astype(np.int64), woffs)
        for mode in (1, 0, -1, 2):                       # 2 = packed with 1536-row columns (6 rows per lane, 8 waves per SIMD)
            got, got2, sums, launches = _run(matcher, pkg, n, mode)
            np.testing.assert_array_equal(got, want, err_msg=f"packed={mode}")
            np.testing.assert_array_equal(got2, want, err_msg=f"packed={mode} (argmin kernel)")
            np.testing.assert_array_equal(sums, wsums, err_msg=f"packed={mode} (index checksums)")
            if mode in (1, 2):
                assert launches == pkg.capi.ROUTE_PACKED      # score + fold kernels: the packed route really ran
            if mode == 0:
                assert launches == pkg.capi.ROUTE_PLAIN
        # the fused loop test sits on top of whichever route the plan picked
        matcher.set_tuning(pkg.capi.TUNE_PACKED, 1)
        cands, npairs = matcher.all_vs_all_loops(cap=max(n, 1))
        matcher.set_tuning(pkg.capi.TUNE_PACKED, 0)
        cands0, _ = matcher.all_vs_all_loops(cap=max(n, 1))
        assert npairs == n
        np.testing.assert_array_equal(cands, cands0)
    finally:
        matcher.set_tuning(pkg.capi.TUNE_PACKED, -1)
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_packed_external_query_set_unsorted_ids(matcher, oracle, pkg, seed):
    """External query frames whose ids are in no order: eligibility is not monotonic along the query list, so the packed
    order (descending eligibility) differs from the caller's order and records must still land at the caller's offsets."""
    rng = np.random.default_rng(seed)
    fs = pkg.synth.make_frames(30, 600, seed=seed, ragged=True, dup_frac=0.3)
    n_q = 17
    q_counts = rng.integers(0, 900, n_q).astype(np.int32)
    q_counts[3] = 0
    q_rows = rng.integers(0, 256, (n_q, 900, 32), dtype=np.uint8)
    for c in range(n_q):                                              # plant copies of stored rows: low distances, ties
        src = int(rng.integers(0, fs.n_frames))
        k = min(int(q_counts[c]), int(fs.counts[src]), 40)
        q_rows[c, :k] = fs.rows[src, :k]
    q_ids = rng.integers(-5, int(fs.ids[-1]) + 40, n_q).astype(np.int32)
    gap = 2
    matcher.set_params(min_gap=gap)
    d_rows, d_counts = matcher.dev_alloc(q_rows.nbytes), matcher.dev_alloc(q_counts.nbytes)
    try:
        _fill(matcher, fs)
        matcher.dev_upload(d_rows, q_rows); matcher.dev_upload(d_counts, q_counts)
        p = oracle.default_params(min_gap=gap)
        want, wsums, woffs = _want(oracle, fs, q_ids, q_rows, q_counts, p)
        ext = dict(d_query_rows=d_rows, d_query_counts=d_counts, q_ids=q_ids, q_stride_rows=900)
        n, offs = matcher.all_vs_all_plan(**ext)
        assert n == len(want) and np.array_equal(offs.astype(np.int64), woffs)
        for mode in (1, 0):
            got, got2, sums, _ = _run(matcher, pkg, n, mode, ext)
            np.testing.assert_array_equal(got, want, err_msg=f"packed={mode}")
            np.testing.assert_array_equal(got2, want, err_msg=f"packed={mode} (argmin kernel)")
            np.testing.assert_array_equal(sums, wsums, err_msg=f"packed={mode} (index checksums)")
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_packed_is_the_automatic_choice_for_orb_sized_frames(matcher, oracle, pkg):
    """2000-row frames leave 48 of 2048 lane slots idle per workgroup: a search big enough to be throughput-bound
    (>= 8192 pairs) takes the packed route by itself; frames that already fill their workgroup shape do not."""
    fs = pkg.synth.make_frames(140, 2000, seed=31, dup_frac=0.2)
    matcher.set_params(min_gap=1)
    try:
        _fill(matcher, fs)
        n, _ = matcher.all_vs_all_plan()
        assert n >= 8192
        d = matcher.dev_alloc(n * 8)
        matcher.all_vs_all(d, n)
        li = matcher.launch_info()
        assert li.route == pkg.capi.ROUTE_PACKED and li.launches == 2
        got = np.zeros(n, pkg.capi.SCORE_DTYPE)
        matcher.sync(); matcher.dev_download(d, got)
        matcher.dev_free(d)
        p = oracle.default_params(min_gap=1)
        rng = np.random.default_rng(5)
        pq, pt = [], []
        for c in range(fs.n_frames):
            for i in range(c):
                pq.append(c); pt.append(i)
        pick = rng.choice(len(pq), 300, replace=False)
        want, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [pq[k] for k in pick], [pt[k] for k in pick], p, n_threads=8)
        np.testing.assert_array_equal(got[pick], want)
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()
    fs = pkg.synth.uniform_frames(140, 512, seed=32)                    # 512 rows on a 64 x 8 workgroup: nothing to gain
    matcher.set_params(min_gap=1)
    try:
        _fill(matcher, fs)
        n, _ = matcher.all_vs_all_plan()
        d = matcher.dev_alloc(n * 8)
        matcher.all_vs_all(d, n)
        assert matcher.launch_info().route == pkg.capi.ROUTE_PLAIN
        matcher.sync()
        matcher.dev_free(d)
    finally:
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("seed", list(range(8)))
def test_packed_equals_plain_on_random_shapes(matcher, oracle, pkg, seed):
    """Differential: the packed and the plain route (two different kernels + a fold kernel vs in-kernel reductions)
    must write the same bytes — records and index checksums — on random ragged databases of random size, gap, id spacing
    and item size; a sample of pairs pins both to the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_frames = int(rng.integers(20, 140))
    max_desc = int(rng.choice([40, 300, 777, 1500, 2048]))
    fs = pkg.synth.make_frames(n_frames, max_desc, seed=2000 + seed, ragged=True, dup_frac=float(rng.uniform(0.0, 0.5)))
    fs.counts[:] = rng.integers(0, max_desc + 1, n_frames)
    fs.counts[int(rng.integers(0, n_frames))] = max_desc
    ids = (np.cumsum(rng.integers(1, 4, n_frames)) - 1).astype(np.int32)
    gap = int(rng.integers(0, 6))
    matcher.set_params(min_gap=gap)
    matcher.set_tuning(pkg.capi.TUNE_ITEM_SLOTS, int(rng.choice([0, 1, 3, 8])))
    matcher.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, int(rng.choice([1, 2, 16, 8192])))     # 1 MiB: 128 pairs per chunk -> many chunks
    try:
        matcher.clear()
        for f in range(n_frames):
            matcher.append(int(ids[f]), fs.frame(f))
        n, offs = matcher.all_vs_all_plan()
        if n == 0:
            return
        res = {}
        for mode in (0, 1, 2):
            got, got2, sums, _ = _run(matcher, pkg, n, mode)
            np.testing.assert_array_equal(got, got2)
            res[mode] = (got, sums)
        for mode in (1, 2):
            np.testing.assert_array_equal(res[mode][0], res[0][0], err_msg=f"records, packed={mode}")
            np.testing.assert_array_equal(res[mode][1], res[0][1], err_msg=f"index checksums, packed={mode}")
        p = oracle.default_params(min_gap=gap)
        pq, pt = [], []
        for c in range(n_frames):
            e = int(offs[c + 1] - offs[c])
            for t in rng.choice(e, size=min(e, 2), replace=False) if e else []:
                pq.append(c); pt.append(int(t))
        want, wsums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=8)
        k = offs[pq].astype(np.int64) + np.array(pt, np.int64)
        np.testing.assert_array_equal(res[1][0][k], want)
        np.testing.assert_array_equal(res[1][1][k], wsums)
    finally:
        matcher.set_tuning(pkg.capi.TUNE_ITEM_SLOTS, 0)
        matcher.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, 1024)          # the default
        matcher.set_tuning(pkg.capi.TUNE_PACKED, -1)
        matcher.set_params(min_gap=30)
        matcher.clear()
