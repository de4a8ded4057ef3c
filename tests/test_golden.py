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
