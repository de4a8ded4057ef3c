"""Committed golden vectors (tests/golden/lcm_golden_v1.npz, made by tests/golden/make_golden.py with the scalar
oracle): the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lcm_golden_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(G, allow_pickle=False)


def frame(g, f):
    return g["rows"][f, : int(g["counts"][f])]


def test_oracle_reproduces_golden(oracle, gold):
    g = gold
    p = oracle.default_params(min_gap=int(g["gap"]), min_matches=int(g["min_matches"]), sim_threshold=float(g["sim_threshold"]))
    scores, offs = oracle.all_vs_all(g["rows"], g["counts"], g["ids"], p)
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(offs.astype(np.int64), g["offsets"])
    cands = np.concatenate([oracle.detect_loops(g["rows"], g["counts"], g["ids"], c, p) for c in range(len(g["counts"]))])
    for f in ("current_frame_id", "matched_frame_id", "num_matches", "similarity_score"):
        np.testing.assert_array_equal(cands[f], g["candidates"][f])
    for k, (a, b) in enumerate(g["pairs"]):
        idx, d = oracle.bf_match(frame(g, a), frame(g, b))
        np.testing.assert_array_equal(idx, g[f"pair{k}_idx"])
        np.testing.assert_array_equal(d, g[f"pair{k}_dist"])


def test_fast_cpu_path_reproduces_golden(oracle, gold):
    g = gold
    p = oracle.default_params(min_gap=int(g["gap"]))
    pq, pt = [], []
    for c in range(len(g["counts"])):
        for i in range(len(g["counts"])):
            if g["ids"][c] - g["ids"][i] >= int(g["gap"]):
                pq.append(c); pt.append(i)
    fast, _, _ = oracle.fast_score_pairs(g["rows"], g["counts"], pq, pt, p, n_threads=2)
    np.testing.assert_array_equal(fast, g["scores"])


@pytest.mark.gpu
def test_gpu_matches_golden(pkg, gold):
    g = gold
    p = pkg.default_params()
    p.min_gap, p.min_matches, p.sim_threshold = int(g["gap"]), int(g["min_matches"]), float(g["sim_threshold"])
    with pkg.Matcher(p) as m:
        for f in range(len(g["counts"])):
            m.append(int(g["ids"][f]), frame(g, f))
        n, offs = m.all_vs_all_plan()
        np.testing.assert_array_equal(offs.astype(np.int64), g["offsets"])
        d = m.dev_alloc(max(n, 1) * 8)
        m.all_vs_all(d, n)
        got = np.zeros(n, pkg.capi.SCORE_DTYPE)
        m.sync()
        m.dev_download(d, got)
        m.dev_free(d)
        np.testing.assert_array_equal(got, g["scores"])
        cands = np.concatenate([m.detect_loops(int(g["ids"][c])) for c in range(len(g["counts"]))])
        for f in ("current_frame_id", "matched_frame_id", "num_matches", "similarity_score"):
            np.testing.assert_array_equal(cands[f], g["candidates"][f])
        for k, (a, b) in enumerate(g["pairs"]):
            idx, dist = m.match_pair(frame(g, a), frame(g, b))
            np.testing.assert_array_equal(idx, g[f"pair{k}_idx"])
            np.testing.assert_array_equal(dist.astype(np.int32), g[f"pair{k}_dist"])
            good, md = m.match_features(frame(g, a), frame(g, b))
            assert md == int(g[f"pair{k}_min"])
            for f in ("query_idx", "train_idx", "img_idx", "distance"):
                np.testing.assert_array_equal(good[f], g[f"pair{k}_good"][f])


# ---- v2: the selective synthetic variant (30 shared pool descriptors per frame), default README parameters ----------
G2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lcm_golden_v2.npz")
CAND_FIELDS = ("current_frame_id", "matched_frame_id", "num_matches", "similarity_score")


@pytest.fixture(scope="module")
def gold2():
    return np.load(G2, allow_pickle=False)


def test_golden_v2_is_selective_and_reproduced_by_the_oracle(oracle, gold2):
    g = gold2
    p = oracle.default_params(min_gap=int(g["gap"]))
    assert p.min_matches == int(g["min_matches"]) == 50 and p.sim_threshold == float(g["sim_threshold"]) == 0.15
    scores, offs = oracle.all_vs_all(g["rows"], g["counts"], g["ids"], p)
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(offs.astype(np.int64), g["offsets"])
    cands = np.concatenate([oracle.detect_loops(g["rows"], g["counts"], g["ids"], c, p) for c in range(len(g["counts"]))])
    for f in CAND_FIELDS:
        np.testing.assert_array_equal(cands[f], g["candidates"][f])
    # what the variant is for: most pairs are NOT loops, some are, and there are good-match counts just under 50
    n = len(g["scores"])
    assert 0 < len(g["candidates"]) < n // 4
    good = g["scores"]["good_count"]
    assert ((good >= 40) & (good < 50)).any() and (good >= 50).any()
    # tuned CPU path: records and index checksums
    pq, pt = [], []
    for c in range(len(g["counts"])):
        for i in range(len(g["counts"])):
            if g["ids"][c] - g["ids"][i] >= int(g["gap"]):
                pq.append(c); pt.append(i)
    fast, sums = oracle.fast_score_pairs_idx(g["rows"], g["counts"], pq, pt, p, n_threads=2)
    np.testing.assert_array_equal(fast, g["scores"])
    np.testing.assert_array_equal(sums, g["index_sums"])


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [0, 3])
def test_gpu_matches_golden_v2(pkg, gold2, shards):
    """single handle (shards = 0) and a 3-shard loopback group: records, index checksums, fused loop search, detectLoops"""
    g = gold2
    p = pkg.default_params()
    p.min_gap = int(g["gap"])
    n_f = len(g["counts"])
    if shards:
        with pkg.Group(p, n_devices=shards, loopback_device=0) as grp:
            for f in range(n_f):
                grp.append(int(g["ids"][f]), frame(g, f))
            sc, ix, offs = grp.all_vs_all_argmin()
            np.testing.assert_array_equal(offs.astype(np.int64), g["offsets"])
            np.testing.assert_array_equal(sc, g["scores"])
            np.testing.assert_array_equal(ix, g["index_sums"])
            cands, pairs = grp.all_vs_all_loops(cap=len(sc))
            assert pairs == len(sc)
            for f in CAND_FIELDS:
                np.testing.assert_array_equal(cands[f], g["candidates"][f])
        return
    with pkg.Matcher(p) as m:
        for f in range(n_f):
            m.append(int(g["ids"][f]), frame(g, f))
        n, offs = m.all_vs_all_plan()
        np.testing.assert_array_equal(offs.astype(np.int64), g["offsets"])
        d, di = m.dev_alloc(n * 8), m.dev_alloc(n * 4)
        m.all_vs_all_argmin(d, n, di)
        got, ix = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
        m.sync(); m.dev_download(d, got); m.dev_download(di, ix)
        m.dev_free(d); m.dev_free(di)
        np.testing.assert_array_equal(got, g["scores"])
        np.testing.assert_array_equal(ix, g["index_sums"])
        cands, _ = m.all_vs_all_loops(cap=n)
        online = np.concatenate([m.detect_loops(int(g["ids"][c])) for c in range(n_f)])
        for f in CAND_FIELDS:
            np.testing.assert_array_equal(cands[f], g["candidates"][f])
            np.testing.assert_array_equal(online[f], g["candidates"][f])
