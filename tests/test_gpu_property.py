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
