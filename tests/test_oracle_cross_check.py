"""Cross-check (BFMatcher crossCheck = true) restatement: known answers for both upstream behaviours (mode 1: with the
`sidx` test = mutual nearest neighbours; mode 2: legacy), and the tuned path == the scalar path under both modes.
PARITY UNPINNED like the rest of the oracle (oracle/lcm_oracle.h): these KATs are the pin."""
import numpy as np
import pytest


def row(bits):
    """A 256-bit descriptor with exactly the given bit positions set."""
    r = np.zeros(32, np.uint8)
    for b in bits:
        r[b // 8] |= 1 << (b % 8)
    return r


def test_mutual_and_legacy_differ_exactly_where_upstream_differs(oracle):
    # distances: q0-t0 = 1, q0-t1 = 2, q1-t0 = 3, q1-t1 = 2  (q0 = {}, q1 = {0,1,2}, t0 = {0}, t1 = {0,1}... see below)
    q = np.stack([row([]), row([0, 1, 2, 3])])
    t = np.stack([row([0]), row([0, 1])])
    # d(q0,t0)=1 d(q0,t1)=2 | d(q1,t0)=3 d(q1,t1)=2
    # forward: q0->t0, q1->t1.  backward: t0->q0 (1 < 3), t1->q0 or q1? d(t1,q0)=2, d(t1,q1)=2: tie -> FIRST query = q0
    i1, d1 = oracle.bf_match_cross(q, t, 1)
    i2, d2 = oracle.bf_match_cross(q, t, 2)
    assert i1.tolist() == [0, -1]            # mutual: (q0,t0) only; q1's choice t1 prefers q0
    assert d1[0] == 1
    assert i2.tolist() == [0, -1]            # legacy: both trains chose q0; q0 keeps the closer one (t0, d=1); q1 unmatched
    # now make t1 strictly prefer q1: t1 = {0,1,2}
    t = np.stack([row([0]), row([0, 1, 2])])
    # d(q0,t1)=3, d(q1,t1)=1 -> forward q0->t0 (1), q1->t1 (1); backward t0->q0, t1->q1: all mutual
    for mode in (1, 2):
        i, d = oracle.bf_match_cross(q, t, mode)
        assert i.tolist() == [0, 1] and d.tolist() == [1, 1]


def test_legacy_keeps_a_non_mutual_match_that_mutual_drops(oracle):
    # q0 is closest to t0, but t1's nearest query is q0 as well and NO train chooses q1 ... and q1's own nearest train is t1
    q = np.stack([row([]), row(range(0, 40))])
    t = np.stack([row([0]), row(range(0, 10)), row(range(100, 130))])
    # d(q0,.) = 1, 10, 30 ; d(q1,.) = 39, 30, 70.  forward: q0->t0, q1->t1.
    # backward: t0->q0 (1 vs 39), t1->q0 (10 vs 30), t2->q0 (30 vs 70)
    i1, _ = oracle.bf_match_cross(q, t, 1)
    i2, d2 = oracle.bf_match_cross(q, t, 2)
    assert i1.tolist() == [0, -1]            # mutual: q1->t1 but t1->q0
    assert i2.tolist() == [0, -1] and d2[0] == 1   # legacy: q0 keeps the best of {t0,t1,t2}; nobody chose q1
    # a train that chooses q1 although q1 prefers another train: legacy keeps it, mutual does not
    q = np.stack([row([]), row(range(0, 40))])
    t = np.stack([row(range(0, 40)) ^ row([200]), row(range(0, 39)), row([0])])
    # d(q1,t0)=1, d(q1,t1)=1 (tie -> forward picks t0), d(q0,t2)=1.  backward: t0->q1, t1->q1, t2->q0
    i1, d1 = oracle.bf_match_cross(q, t, 1)
    i2, d2 = oracle.bf_match_cross(q, t, 2)
    assert i1.tolist() == [2, 0] and i2.tolist() == [2, 0]       # both: q1 gets t0 (legacy: first train on the tie too)
    t2 = t[[1, 0, 2]]                                           # swap t0 and t1: the tie now resolves to the other row
    i1, _ = oracle.bf_match_cross(q, t2, 1)
    assert i1.tolist() == [2, 0]


def test_empty_sides_and_single_rows(oracle):
    q = np.stack([row([1]), row([2])])
    for mode in (1, 2):
        i, d = oracle.bf_match_cross(q, np.zeros((0, 32), np.uint8), mode)
        assert i.tolist() == [-1, -1]
        i, d = oracle.bf_match_cross(q, q[:1], mode)            # one train row: only ITS nearest query can match
        assert i.tolist() == [0, -1] and d[0] == 0


@pytest.mark.parametrize("mode", [1, 2])
def test_match_features_and_scores_under_cross_check(oracle, pkg, mode):
    """matchFeatures / pair_score / the tuned path agree with a direct numpy statement of the rule."""
    fs = pkg.synth.make_frames(10, 150, seed=31 + mode, ragged=True, dup_frac=0.6)
    fs.counts[3] = 0
    fs.rows[5, :40] = fs.rows[2, :40]
    fs.rows[5, 40:80] = fs.rows[2, :40]                          # asymmetric ties: two query rows equal one train row
    p = oracle.default_params(min_gap=1, cross_check=mode)
    pq = [c for c in range(10) for t in range(10) if c != t]
    pt = [t for c in range(10) for t in range(10) if c != t]
    fast, sums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=3)
    n_dropped = 0
    for k, (c, t) in enumerate(zip(pq, pt)):
        Q, T = fs.frame(c), fs.frame(t)
        want = oracle.pair_score(Q, T, p)
        assert fast[k] == want, (c, t)
        m, md = oracle.match_features(Q, T, p)
        assert len(m) == int(want["good_count"])
        assert int(sums[k]) == int(m["train_idx"].astype(np.uint64).sum() % (1 << 32))
        if len(Q) and len(T):
            D = np.unpackbits(Q[:, None, :] ^ T[None, :, :], axis=2).sum(axis=2)
            f = D.argmin(axis=1); b = D.argmin(axis=0)           # numpy argmin = first minimum, both directions
            if mode == 1:
                keep = {i: int(f[i]) for i in range(len(Q)) if b[f[i]] == i}
            else:
                keep = {}
                for j in range(len(T)):
                    i = int(b[j])
                    if i not in keep or D[i, j] < D[i, keep[i]]:
                        keep[i] = j
            idx, dd = oracle.bf_match_cross(Q, T, mode)
            assert {i: int(idx[i]) for i in range(len(Q)) if idx[i] >= 0} == keep
            n_dropped += len(Q) - len(keep)
            if keep:
                dk = np.array([D[i, j] for i, j in keep.items()])
                thr = max(2 * dk.min(), 0)
                assert int(want["good_count"]) == int((dk <= thr).sum()) and int(want["min_dist"]) == int(dk.min())
    assert n_dropped > 100
