"""C oracle vs an independently written numpy restatement, and tuned CPU baseline vs golden oracle."""
import numpy as np
import pytest

from npref import bf_match, pair_score


@pytest.mark.parametrize("nq,nt,seed", [(1, 1, 0), (64, 65, 1), (37, 300, 2), (257, 129, 3), (5, 1, 4), (1, 200, 5)])
def test_bf_match_matches_numpy(oracle, nq, nt, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    # plant ties and duplicates
    if nt > 4:
        t[nt - 1] = t[0]
        t[nt // 2] = t[1]
        q[0] = t[0]
    i1, d1 = oracle.bf_match(q, t)
    i2, d2 = bf_match(q, t)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(d1, d2)
    s = oracle.pair_score(q, t)
    assert (int(s["good_count"]), int(s["min_dist"]), int(s["n_train"])) == pair_score(q, t)


def test_low_entropy_rows_many_ties(oracle):
    rng = np.random.default_rng(11)
    # rows drawn from a tiny alphabet: almost every minimum is tied several times
    alphabet = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    q = alphabet[rng.integers(0, 6, 90)]
    t = alphabet[rng.integers(0, 6, 140)]
    i1, d1 = oracle.bf_match(q, t)
    i2, d2 = bf_match(q, t)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(d1, d2)


def test_fast_baseline_equals_golden(oracle, pkg):
    fs = pkg.synth.make_frames(12, 150, seed=77, ragged=True, dup_frac=0.5)
    fs.counts[3] = 0                                            # an empty frame in the mix
    p = oracle.default_params(min_gap=2)
    golden, offs = oracle.all_vs_all(fs.rows, fs.counts, fs.ids, p)
    pq, pt = [], []
    for c in range(fs.n_frames):
        for i in range(fs.n_frames):
            if fs.ids[c] - fs.ids[i] >= 2:
                pq.append(c); pt.append(i)
    for threads in (1, 3):
        fast, secs, isa = oracle.fast_score_pairs(fs.rows, fs.counts, pq, pt, p, n_threads=threads)
        assert isa in ("avx512-vpopcntdq", "popcnt64")
        np.testing.assert_array_equal(fast, golden)


def test_all_vs_all_shards_partition_the_pairs(oracle, pkg):
    fs = pkg.synth.make_frames(20, 40, seed=5)
    p = oracle.default_params(min_gap=3)
    full, offs = oracle.all_vs_all(fs.rows, fs.counts, fs.ids, p)
    shards = [oracle.all_vs_all(fs.rows, fs.counts, fs.ids, p, r, 3)[0] for r in range(3)]
    assert sum(len(s) for s in shards) == len(full)
    merged, moffs = pkg.sharding.merge_shard_scores(shards, fs.ids, 3)
    np.testing.assert_array_equal(merged, full)
    np.testing.assert_array_equal(moffs, offs.astype(np.int64))


def test_fast_path_index_checksum_equals_scalar_match_features(oracle, pkg):
    """orc_fast_score_pairs_idx's per-pair checksum (sum of the good matches' trainIdx mod 2^32) == the same sum formed
    from the scalar oracle's matchFeatures list; ties, ragged sizes, an empty frame and min_dist == 0 included."""
    fs = pkg.synth.make_frames(14, 180, seed=99, ragged=True, dup_frac=0.5)
    fs.counts[5] = 0
    fs.rows[7, :40] = fs.rows[3, :40]                       # exact duplicates across frames: min_dist == 0, many ties
    fs.rows[7, 40:80] = fs.rows[3, :40]
    p = oracle.default_params(min_gap=1)
    pq = [c for c in range(14) for t in range(14) if c != t]
    pt = [t for c in range(14) for t in range(14) if c != t]
    scores, sums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, p, n_threads=3)
    plain, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, pq, pt, p, n_threads=3)
    np.testing.assert_array_equal(scores, plain)
    for k, (c, t) in enumerate(zip(pq, pt)):
        assert int(sums[k]) == oracle.index_sum(fs.frame(c), fs.frame(t), p), (c, t)
    assert len(set(sums.tolist())) > 50
