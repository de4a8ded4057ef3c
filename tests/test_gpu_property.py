"""K11: hypothesis-driven random + adversarial-tie inputs, HIP result == scalar oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


@st.composite
def descriptor_sets(draw):
    nq = draw(st.integers(1, 300))
    nt = draw(st.integers(1, 300))
    seed = draw(st.integers(0, 2**31 - 1))
    alphabet = draw(st.integers(1, 12))           # small alphabets force massive ties
    flips = draw(st.integers(0, 3))
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (alphabet, 32), dtype=np.uint8)
    q = base[rng.integers(0, alphabet, nq)].copy()
    t = base[rng.integers(0, alphabet, nt)].copy()
    for _ in range(flips):                          # a few single-bit perturbations -> near-ties at distance 1, 2
        q[rng.integers(0, nq), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        t[rng.integers(0, nt), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return q, t


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(descriptor_sets())
def test_match_pair_equals_oracle(matcher, oracle, qt):
    q, t = qt
    idx, d = matcher.match_pair(q, t)
    oi, od = oracle.bf_match(q, t)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(d.astype(np.int32), od)
    good, md = matcher.match_features(q, t)
    og, omd = oracle.match_features(q, t)
    assert md == omd
    np.testing.assert_array_equal(good["query_idx"], og["query_idx"])
    np.testing.assert_array_equal(good["train_idx"], og["train_idx"])


@st.composite
def small_databases(draw):
    n_frames = draw(st.integers(1, 14))
    max_rows = draw(st.sampled_from([1, 3, 4, 5, 63, 64, 65, 130]))
    seed = draw(st.integers(0, 2**31 - 1))
    gap = draw(st.integers(0, 4))
    id_step = draw(st.integers(1, 3))
    rng = np.random.default_rng(seed)
    alphabet = rng.integers(0, 256, (draw(st.integers(1, 6)), 32), dtype=np.uint8)
    counts = rng.integers(0, max_rows + 1, n_frames).astype(np.int32)
    rows = np.zeros((n_frames, max_rows, 32), np.uint8)
    for f in range(n_frames):
        rows[f, : counts[f]] = alphabet[rng.integers(0, len(alphabet), counts[f])]
        if counts[f] and rng.random() < 0.5:
            rows[f, rng.integers(0, counts[f]), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    ids = (np.cumsum(rng.integers(1, id_step + 1, n_frames)) - 1).astype(np.int32)
    return rows, counts, ids, gap


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(small_databases())
def test_bulk_and_online_paths_equal_oracle(matcher, oracle, db):
    """Random ragged databases (empty frames, tie-heavy rows, irregular ids, any gap): lcm_all_vs_all, lcm_query_scores and
    lcm_detect_loops == the scalar oracle."""
    rows, counts, ids, gap = db
    matcher.set_params(min_gap=gap, min_matches=1, sim_threshold=0.0)
    p = oracle.default_params(min_gap=gap, min_matches=1, sim_threshold=0.0)
    try:
        matcher.clear()
        for f in range(len(counts)):
            matcher.append(int(ids[f]), rows[f, : counts[f]])
        n, offs = matcher.all_vs_all_plan()
        want, woffs = oracle.all_vs_all(rows, counts, ids, p)
        assert n == len(want) and np.array_equal(offs.astype(np.int64), woffs.astype(np.int64))
        if n:
            d = matcher.dev_alloc(n * 8)
            matcher.all_vs_all(d, n)
            got = np.zeros(n, want.dtype)
            matcher.sync()
            matcher.dev_download(d, got)
            matcher.dev_free(d)
            np.testing.assert_array_equal(got, want)
        cur = len(counts) - 1
        s, sid = matcher.query_scores(rows[cur, : counts[cur]], int(ids[cur]))
        np.testing.assert_array_equal(s, want[int(woffs[cur]):int(woffs[cur + 1])])
        c = matcher.detect_loops(int(ids[cur]))
        wc = oracle.detect_loops(rows, counts, ids, cur, p)
        for f in ("matched_frame_id", "num_matches", "similarity_score"):
            np.testing.assert_array_equal(c[f], wc[f])
    finally:
        matcher.set_params(min_gap=30, min_matches=50, sim_threshold=0.15)
        matcher.clear()


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(small_databases(), st.sampled_from([0, 1, 2]), st.sampled_from([0, 1, 4, 5]), st.sampled_from([-1, 1]))
def test_every_bulk_route_equals_oracle(matcher, pkg, oracle, db, cross, variant, packed):
    """The round-2 routes through the same random ragged databases: lcm_all_vs_all_argmin (records + index checksum),
    cross_check 0 / 1 / 2, kernel variants 0 / 1 / 4 / 5, the packed bulk form (forced: all frames of these small
    databases share ONE 2048-row workgroup column), micro-batched online queries, batched match lists."""
    rows, counts, ids, gap = db
    matcher.set_params(min_gap=gap, min_matches=1, sim_threshold=0.0, cross_check=cross)
    matcher.set_kernel_variant(variant)
    matcher.set_tuning(pkg.capi.TUNE_PACKED, packed)
    p = oracle.default_params(min_gap=gap, min_matches=1, sim_threshold=0.0, cross_check=cross)
    n_frames = len(counts)
    try:
        matcher.clear()
        for f in range(n_frames):
            matcher.append(int(ids[f]), rows[f, : counts[f]])
        pq, pt = [], []
        for c in range(n_frames):
            for t in range(n_frames):
                if ids[c] - ids[t] >= max(gap, 1):
                    pq.append(c); pt.append(t)
        want, wsums = oracle.fast_score_pairs_idx(rows, counts, pq, pt, p, n_threads=2)
        n, offs = matcher.all_vs_all_plan()
        assert n == len(pq)
        if n:
            d, ds = matcher.dev_alloc(n * 8), matcher.dev_alloc(n * 4)
            got, sums = np.zeros(n, want.dtype), np.zeros(n, np.uint32)
            matcher.all_vs_all(d, n)
            matcher.sync(); matcher.dev_download(d, got)
            np.testing.assert_array_equal(got, want)
            matcher.all_vs_all_argmin(d, n, ds)
            matcher.sync(); matcher.dev_download(d, got); matcher.dev_download(ds, sums)
            matcher.dev_free(d); matcher.dev_free(ds)
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(sums, wsums)
        # the last three frames as one micro-batch against the frames before them
        k = min(3, n_frames)
        first = n_frames - k
        if ids[n_frames - 1] - ids[first] < max(gap, 1) or k == 1:       # exactness condition of a batch
            matcher.clear()
            for f in range(first):
                matcher.append(int(ids[f]), rows[f, : counts[f]])
            t = matcher.query_submit_batch([rows[f, : counts[f]] for f in range(first, n_frames)], [int(ids[f]) for f in range(first, n_frames)])
            sc, boffs = matcher.query_collect_batch(t)
            for j, f in enumerate(range(first, n_frames)):
                np.testing.assert_array_equal(sc[int(boffs[j]): int(boffs[j + 1])], want[int(offs[f]): int(offs[f + 1])])
            for f in range(first, n_frames):
                matcher.append(int(ids[f]), rows[f, : counts[f]])
        # match lists of a few pairs in one launch
        pairs = list(zip(pq, pt))[:5]
        if pairs:
            lists, mins = matcher.match_stored_batch([(int(ids[a]), int(ids[b])) for a, b in pairs])
            for (a, b), got_l in zip(pairs, lists):
                om, _ = oracle.match_features(rows[a, : counts[a]], rows[b, : counts[b]], p)
                np.testing.assert_array_equal(got_l, om.astype(got_l.dtype))
    finally:
        matcher.set_kernel_variant(0)
        matcher.set_tuning(pkg.capi.TUNE_PACKED, -1)
        matcher.set_params(min_gap=30, min_matches=50, sim_threshold=0.15, cross_check=0)
        matcher.clear()
