"""HIP path vs the CPU oracle, through the C ABI, bit-exact (integer / index work).  Needs a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


# (n_query, n_train): SURVEY.md §8c K5 ragged shapes + wave/workgroup/chunk edges (64 lanes, 256 threads,
# 2048-row query chunks, 4-row train padding)
SHAPES = [(1, 1), (1, 2), (2, 1), (3, 5), (64, 65), (63, 64), (65, 63), (1, 2000), (2000, 1), (500, 500),
          (513, 7), (1024, 1025), (1025, 9), (1500, 40), (1536, 5), (1537, 33), (1999, 1777), (2000, 2000), (2048, 2048), (2049, 11), (4100, 130),
          (7, 4097)]


@pytest.mark.parametrize("nq,nt", SHAPES)
def test_match_pair_bit_exact(matcher, oracle, nq, nt):
    rng = np.random.default_rng(nq * 100003 + nt)
    q, t = rnd(rng, nq), rnd(rng, nt)
    # planted ties: duplicate train rows, exact query copies
    if nt >= 4:
        t[nt - 1] = t[0]
        t[nt // 2] = t[1]
    if nq >= 2 and nt >= 2:
        q[0] = t[0]
        q[nq - 1] = t[1]
    idx, dist = matcher.match_pair(q, t)
    oi, od = oracle.bf_match(q, t)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist.astype(np.int32), od)


def test_empty_inputs(matcher):
    q0 = np.zeros((0, 32), np.uint8)
    t = np.ones((5, 32), np.uint8)
    assert len(matcher.match_pair(q0, t)[0]) == 0
    assert len(matcher.match_pair(t, q0)[0]) == 0
    m, md = matcher.match_features(t, q0)
    assert len(m) == 0 and md == -1


def test_kat_on_device(matcher):
    Z = np.zeros((1, 32), np.uint8)
    F = np.full((1, 32), 0xFF, np.uint8)
    idx, d = matcher.match_pair(Z, F)
    assert idx.tolist() == [0] and d.tolist() == [256]
    idx, d = matcher.match_pair(F, F)
    assert d.tolist() == [0]
    # single-bit flips in every byte / bit position
    q = np.zeros((256, 32), np.uint8)
    for b in range(256):
        q[b, b // 8] = 1 << (b % 8)
    idx, d = matcher.match_pair(q, Z)
    assert d.tolist() == [1] * 256 and idx.tolist() == [0] * 256
    # query row b is closest to train row b when the train set is the same single-bit rows
    idx, d = matcher.match_pair(q, q)
    assert idx.tolist() == list(range(256)) and d.tolist() == [0] * 256


def test_tie_break_lowest_index(matcher):
    rng = np.random.default_rng(2)
    A = rnd(rng, 1)
    B = A.copy(); B[0, 5] ^= 0xFF
    t = np.concatenate([A, B, A] + [B] * 300 + [A])
    idx, d = matcher.match_pair(A, t)
    assert idx.tolist() == [0] and d.tolist() == [0]
    C1 = A.copy(); C1[0, 0] ^= 1
    C2 = A.copy(); C2[0, 9] ^= 8
    idx, d = matcher.match_pair(A, np.concatenate([B, C1, C2]))
    assert idx.tolist() == [1] and d.tolist() == [1]
    # all train rows identical: index 0 for every query, whatever the padding does
    for nt in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        idx, d = matcher.match_pair(rnd(rng, 70), np.repeat(A, nt, axis=0))
        assert (idx == 0).all()


def test_low_entropy_many_ties(matcher, oracle):
    rng = np.random.default_rng(11)
    alphabet = rnd(rng, 6)
    q = alphabet[rng.integers(0, 6, 900)]
    t = alphabet[rng.integers(0, 6, 1400)]
    idx, dist = matcher.match_pair(q, t)
    oi, od = oracle.bf_match(q, t)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(dist.astype(np.int32), od)


@pytest.mark.parametrize("nq,nt", [(300, 280), (2000, 1500), (2100, 100)])
def test_match_features_bit_exact(matcher, oracle, pkg, nq, nt):
    fs = pkg.synth.make_frames(8, max(nq, nt), seed=31, dup_frac=0.5)
    q = fs.rows[5, :nq]
    t = fs.rows[1, :nt]              # same place (8 // 4 = 2 places; 5 % 2 == 1 % 2): real inliers
    m, md = matcher.match_features(q, t)
    om, omd = oracle.match_features(q, t)
    assert md == omd
    for f in ("query_idx", "train_idx", "img_idx", "distance"):
        np.testing.assert_array_equal(m[f], om[f])


def test_filter_params_take_effect(matcher, oracle, pkg):
    fs = pkg.synth.make_frames(8, 400, seed=9)
    q, t = fs.frame(6), fs.frame(2)
    old = matcher.params
    try:
        for ratio, floor in [(2, 0), (3, 0), (2, 64), (1, 0), (0, 0)]:
            matcher.set_params(ratio=ratio, dist_floor=floor)
            m, md = matcher.match_features(q, t)
            om, omd = oracle.match_features(q, t, oracle.default_params(ratio=ratio, dist_floor=floor))
            assert md == omd and len(m) == len(om)
            np.testing.assert_array_equal(m["train_idx"], om["train_idx"])
    finally:
        matcher.set_params(ratio=old.ratio, dist_floor=old.dist_floor)


@pytest.mark.parametrize("upload_kernel,host_fold", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_pair_mode_latency_paths_give_the_same_matches(pkg, oracle, upload_kernel, host_fold):
    """The latency shape of the pair mode (calls of <= 64 M distances) has two switches — staging block uploaded by a
    kernel or by hipMemcpyAsync, folded keys written into pinned host memory or into device memory + a copy: all four
    combinations, ragged sizes on both sides of the 512-row chunk and the segment seams, against the oracle; and a call
    above the 64 M-distance limit (throughput shape) between them."""
    rng = np.random.default_rng(77)
    with pkg.Matcher() as m:
        m.set_tuning(pkg.capi.TUNE_PAIR_UPLOAD_KERNEL, upload_kernel)
        m.set_tuning(pkg.capi.TUNE_PAIR_HOST_FOLD, host_fold)
        shapes = [(2000, 2000), (1, 1), (513, 31), (512, 33), (1025, 4097), (37, 20000), (3000, 700), (300, 300)]
        if upload_kernel and host_fold:
            shapes.insert(5, (9000, 8000))           # 72 M distances: the throughput shape, between two latency-shaped calls
        for nq, nt in shapes:
            q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            t[rng.integers(0, nt, max(nt // 7, 1))] = q[rng.integers(0, nq, max(nt // 7, 1))]      # exact copies: ties on distance 0
            idx, dist = m.match_pair(q, t)
            oi, od = oracle.bf_match(q, t)
            np.testing.assert_array_equal(idx, oi, err_msg=f"{nq} x {nt}")
            np.testing.assert_array_equal(dist.astype(np.int32), od, err_msg=f"{nq} x {nt}")
