"""Known-answer tests that PIN the oracle.  The reference ships no tests or golden vectors
for this path and OpenCV is absent, so these hand-derivable answers are the pin ("parity unpinned" otherwise)."""
import numpy as np
import pytest


def rows(*rs):
    return np.array(rs, np.uint8).reshape(-1, 32)


Z = np.zeros(32, np.uint8)
F = np.full(32, 0xFF, np.uint8)


def test_k1_distances(oracle):
    assert oracle.hamming(Z, F) == 256
    assert oracle.hamming(F, F) == 0
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, 32, dtype=np.uint8)
    assert oracle.hamming(x, x) == 0
    for byte in range(32):          # a single flipped bit in any byte counts 1: byte-order independent
        for bit in (0, 3, 7):
            y = x.copy()
            y[byte] ^= np.uint8(1 << bit)
            assert oracle.hamming(x, y) == 1
    a = Z.copy(); a[0] = 0b10110000; a[31] = 0b00000111
    assert oracle.hamming(a, Z) == 6


def test_k2_tie_break_first_minimum(oracle):
    rng = np.random.default_rng(2)
    A = rng.integers(0, 256, 32, dtype=np.uint8)
    B = A.copy(); B[5] ^= 0xFF
    idx, d = oracle.bf_match(rows(A), rows(A, B, A))
    assert idx.tolist() == [0] and d.tolist() == [0]          # exact duplicate at 0 and 2 -> 0
    # two train rows at the same non-zero distance -> the lower index
    C1 = A.copy(); C1[0] ^= 0b1
    C2 = A.copy(); C2[9] ^= 0b1000
    idx, d = oracle.bf_match(rows(A), rows(B, C1, C2))
    assert idx.tolist() == [1] and d.tolist() == [1]
    idx, d = oracle.bf_match(rows(A), rows(B, C2, C1))
    assert idx.tolist() == [1] and d.tolist() == [1]
    # a later STRICTLY smaller distance still wins
    idx, d = oracle.bf_match(rows(A), rows(C1, B, A))
    assert idx.tolist() == [2] and d.tolist() == [0]


def test_k3_single_train_row(oracle):
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (17, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    idx, d = oracle.bf_match(q, t)
    assert idx.tolist() == [0] * 17
    assert d.tolist() == [oracle.hamming(q[i], t[0]) for i in range(17)]


def test_k4_empty_inputs(oracle):
    q = np.zeros((0, 32), np.uint8)
    t = np.ones((5, 32), np.uint8)
    assert len(oracle.bf_match(q, t)[0]) == 0
    assert len(oracle.bf_match(t, q)[0]) == 0                   # no train rows -> no matches at all
    m, md = oracle.match_features(t, q)
    assert len(m) == 0 and md == -1
    s = oracle.pair_score(t, q)
    assert int(s["good_count"]) == 0 and int(s["min_dist"]) == 0xFFFF and int(s["n_train"]) == 0
    ok, sim = oracle.loop_test(0, 5, 0)
    assert not ok and sim == 0.0                                # no 0/0


def test_k6_filter_boundaries(oracle):
    g, keep, m = oracle.filter_good([0, 0, 1, 5], 2, 0)        # min 0 -> only distance-0 matches survive
    assert (g, m, keep.tolist()) == (2, 0, [True, True, False, False])
    g, keep, m = oracle.filter_good([7, 14, 15, 9], 2, 0)      # min 7 -> threshold 14 inclusive
    assert (g, m, keep.tolist()) == (3, 7, [True, True, False, True])
    g, keep, m = oracle.filter_good([7, 14, 15, 30, 31], 2, 30)  # a floor above 2*min takes over
    assert (g, m, keep.tolist()) == (4, 7, [True, True, True, True, False])
    g, keep, m = oracle.filter_good([], 2, 0)
    assert (g, m) == (0, -1)


def test_k7_loop_test_boundaries(oracle):
    assert oracle.loop_test(49, 100, 100)[0] is False           # 49 < 50 matches
    assert oracle.loop_test(50, 100, 100)[0] is True
    ok, sim = oracle.loop_test(300, 2000, 2000)                 # similarity exactly 0.15 must NOT pass ('>')
    assert sim == 300 / 2000 and ok is False
    assert oracle.loop_test(301, 2000, 2000)[0] is True
    # header default threshold 0.7 (include/loop_closing.hpp:31)
    p = oracle.default_params(sim_threshold=0.7)
    assert oracle.loop_test(1400, 2000, 2000, p)[0] is False
    assert oracle.loop_test(1401, 2000, 2000, p)[0] is True


def test_k9_denominator_is_smaller_frame(oracle):
    ok, sim = oracle.loop_test(60, 2000, 300)
    assert sim == 60 / 300 and ok
    ok, sim = oracle.loop_test(60, 300, 2000)
    assert sim == 60 / 300 and ok


def test_k8_gap_is_inclusive(oracle):
    rng = np.random.default_rng(8)
    n = 40
    base = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    frames = np.repeat(base[None], n, axis=0)                   # identical frames: every eligible pair is a loop
    counts = np.full(n, 60, np.int32)
    ids = np.arange(n, dtype=np.int32)
    c = oracle.detect_loops(frames, counts, ids, 35)
    assert c["matched_frame_id"].tolist() == [0, 1, 2, 3, 4, 5]  # 35-5 = 30 included, 35-6 = 29 skipped
    assert (c["num_matches"] == 60).all() and (c["similarity_score"] == 1.0).all()
    assert (c["current_frame_id"] == 35).all()
    assert len(oracle.detect_loops(frames, counts, ids, 29)) == 0
    # ids, not positions, carry the gap
    ids3 = ids * 3
    c = oracle.detect_loops(frames, counts, ids3, 12)            # id 36: ids <= 6 -> positions 0,1,2
    assert c["matched_frame_id"].tolist() == [0, 3, 6]


def test_match_features_record_layout(oracle):
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    q = t[[7, 3, 3, 20]].copy()
    q[3, 0] ^= 0x0F                                            # distance 4 -> filtered out (min 0 -> thr 0)
    m, md = oracle.match_features(q, t)
    assert md == 0
    assert m["query_idx"].tolist() == [0, 1, 2]
    assert m["train_idx"].tolist() == [7, 3, 3]
    assert m["img_idx"].tolist() == [0, 0, 0]
    assert m["distance"].tolist() == [0.0, 0.0, 0.0]
