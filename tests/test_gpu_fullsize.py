"""BASELINE.json's full sizes (2000-row frames, 1000-frame database): size-independent properties of the domain
plus sampled oracle checks, since the scalar oracle cannot score 470,935 full-size pairs in test time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def ham(a, b):
    return np.unpackbits(a ^ b, axis=-1).sum(axis=-1)


def test_self_match_is_identity_up_to_duplicates(matcher):
    rng = np.random.default_rng(1)
    a = rnd(rng, 2000)
    a[1500] = a[20]                                   # a duplicate: row 1500 must report index 20, not itself
    idx, d = matcher.match_pair(a, a)
    assert (d == 0).all()
    want = np.arange(2000); want[1500] = 20
    np.testing.assert_array_equal(idx, want)


def test_reported_distance_is_distance_to_reported_row(matcher):
    rng = np.random.default_rng(2)
    q, t = rnd(rng, 2000), rnd(rng, 2000)
    idx, d = matcher.match_pair(q, t)
    np.testing.assert_array_equal(ham(q, t[idx]), d)
    # and no row sampled at random beats it; ties must not have a lower index
    for _ in range(20):
        j = rng.integers(0, 2000, 2000)
        dj = ham(q, t[j])
        assert (dj >= d).all()
        assert ((dj > d) | (j >= idx)).all()


def test_concatenation_and_monotonicity(matcher):
    """best over [T1; T2] == lexicographic min of (best over T1) and (best over T2, index shifted)."""
    rng = np.random.default_rng(3)
    q, t1, t2 = rnd(rng, 2000), rnd(rng, 1203), rnd(rng, 797)
    t2[5] = t1[7]                                     # tie across the seam: T1's copy must win
    q[0] = t1[7]
    i1, d1 = matcher.match_pair(q, t1)
    i2, d2 = matcher.match_pair(q, t2)
    i12, d12 = matcher.match_pair(q, np.concatenate([t1, t2]))
    take2 = d2 < d1                                   # strict: ties go to the earlier block
    np.testing.assert_array_equal(d12, np.where(take2, d2, d1))
    np.testing.assert_array_equal(i12, np.where(take2, i2 + len(t1), i1))
    assert i12[0] == 7 and d12[0] == 0
    assert (d12 <= d1).all()                          # more train rows can only lower the best distance


def test_train_permutation_invariance(matcher):
    rng = np.random.default_rng(4)
    q, t = rnd(rng, 1999), rnd(rng, 1777)
    perm = rng.permutation(len(t))
    i0, d0 = matcher.match_pair(q, t)
    i1, d1 = matcher.match_pair(q, t[perm])
    np.testing.assert_array_equal(d0, d1)
    # the row found in the permuted set is one of the tied minima, and the unpermuted answer is the LOWEST of them
    back = perm[i1]
    np.testing.assert_array_equal(ham(q, t[back]), d0)
    assert (back >= i0).all()
    dm = np.unpackbits(q[:200, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    unique = (dm == dm.min(axis=1, keepdims=True)).sum(axis=1) == 1
    np.testing.assert_array_equal(back[:200][unique], i0[:200][unique])
    assert unique.sum() > 100


def test_query_order_equivariance_and_chunking(matcher):
    """Queries are independent: any subset / order gives the same per-row answers (also across the 2048-row
    chunk boundary of the pair kernel)."""
    rng = np.random.default_rng(5)
    q, t = rnd(rng, 5000), rnd(rng, 900)
    i0, d0 = matcher.match_pair(q, t)
    perm = rng.permutation(len(q))
    i1, d1 = matcher.match_pair(q[perm], t)
    np.testing.assert_array_equal(i1, i0[perm])
    np.testing.assert_array_equal(d1, d0[perm])
    i2, d2 = matcher.match_pair(q[2040:2060], t)
    np.testing.assert_array_equal(i2, i0[2040:2060])


def test_cfg2_full_database_sampled_vs_oracle_and_sharded(matcher, oracle, pkg):
    """configs[1]: 1000 frames x 2000 descriptors, min_gap 30 — all 470,935 pairs on the GPU; 200 random pairs checked
    against the oracle's tuned CPU path (itself checked against the scalar oracle in test_oracle_numpy.py) and 3
    against the scalar oracle; the 4-way cyclic-sharded run must merge to the byte-identical array (K10); both kernel
    variants (distance-only / full keys) must agree."""
    fs = pkg.synth.make_frames(1000, 2000, seed=pkg.synth.BASE_SEED + 2)
    gap = 30
    matcher.set_params(min_gap=gap)
    d_rows = matcher.dev_alloc(fs.rows.nbytes)
    d_counts = matcher.dev_alloc(fs.counts.nbytes)
    try:
        matcher.dev_upload(d_rows, fs.rows)
        matcher.dev_upload(d_counts, fs.counts)

        def run(positions, external):
            matcher.clear()
            fb = fs.stride_rows * 32
            for f in positions:
                matcher.append_device(int(fs.ids[f]), d_rows + int(f) * fb, int(fs.counts[f]))
            kw = dict(d_query_rows=d_rows, d_query_counts=d_counts, q_ids=fs.ids, q_stride_rows=fs.stride_rows) if external else {}
            n, offs = matcher.all_vs_all_plan(**kw)
            d = matcher.dev_alloc(max(n, 1) * 8)
            matcher.all_vs_all(d, n, **kw)
            out = np.zeros(n, pkg.capi.SCORE_DTYPE)
            matcher.sync()
            matcher.dev_download(d, out)
            matcher.dev_free(d)
            return out, offs

        full, offs = run(range(1000), False)
        assert len(full) == 470935 == pkg.synth.n_pairs_all_vs_all(1000, gap)
        info = matcher.launch_info()
        assert info.distances == 470935 * 2000 * 2000
        # canary, deliberately loose (measured 655 ms = 2.87e12 distances/s): a silent fallback or a de-optimised
        # build would be several times slower
        assert info.kernel_ms < 1300, f"cfg2 pass took {info.kernel_ms:.0f} ms"
        assert (full["n_train"] == 2000).all() and (full["good_count"] <= 2000).all() and (full["good_count"] >= 1).all()

        matcher.set_kernel_variant(1)
        keyed, _ = run(range(1000), False)
        argmin_ms = matcher.launch_info().kernel_ms
        matcher.set_kernel_variant(0)
        np.testing.assert_array_equal(keyed, full)
        # the argmin kernel may cost at most a few per cent over the distance-only one (measured: see DESIGN.md §5)
        assert argmin_ms < 1.10 * info.kernel_ms, (argmin_ms, info.kernel_ms)
        # ... and its indices, through the per-pair checksum of lcm_all_vs_all_argmin, against the oracle on a sample
        d_sc, d_su = matcher.dev_alloc(470935 * 8), matcher.dev_alloc(470935 * 4)
        assert matcher.all_vs_all_argmin(d_sc, 470935, d_su) == 470935
        sc2, sums = np.zeros(470935, pkg.capi.SCORE_DTYPE), np.zeros(470935, np.uint32)
        matcher.sync()
        matcher.dev_download(d_sc, sc2); matcher.dev_download(d_su, sums)
        matcher.dev_free(d_sc); matcher.dev_free(d_su)
        np.testing.assert_array_equal(sc2, full)

        # the opt-in matrix-core variants: same 470,935 records
        mfma_ms = {}
        for v in (4, 5):
            matcher.set_kernel_variant(v)
            mfma, _ = run(range(1000), False)
            mfma_ms[v] = matcher.launch_info().kernel_ms
            matcher.set_kernel_variant(0)
            np.testing.assert_array_equal(mfma, full)
        print(f"cfg2 kernels: distance-only {info.kernel_ms:.1f} ms, argmin {argmin_ms:.1f} ms, "
              f"matrix-core int8 {mfma_ms[4]:.1f} ms, fp4 {mfma_ms[5]:.1f} ms")

        rng = np.random.default_rng(7)
        qs = rng.integers(gap, 1000, 200)
        ts = np.array([rng.integers(0, q - gap + 1) for q in qs])
        p = oracle.default_params(min_gap=gap)
        cpu, cpu_sums = oracle.fast_score_pairs_idx(fs.rows, fs.counts, qs, ts, p, n_threads=8)
        np.testing.assert_array_equal(full[offs[qs].astype(np.int64) + ts], cpu)
        np.testing.assert_array_equal(sums[offs[qs].astype(np.int64) + ts], cpu_sums)
        for q, t in zip(qs[:3], ts[:3]):
            assert full[int(offs[q]) + int(t)] == oracle.pair_score(fs.frame(q), fs.frame(t), p)

        shards = [run(pkg.sharding.owned_positions(1000, r, 4), True)[0] for r in range(4)]
        merged, moffs = pkg.sharding.merge_shard_scores(shards, fs.ids, gap)
        np.testing.assert_array_equal(merged, full)
        np.testing.assert_array_equal(moffs, offs.astype(np.int64))
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_params(min_gap=30)
        matcher.clear()


def test_search_larger_than_one_launch(pkg, oracle):
    """More than 2^20 work items: the bulk search goes out as several launches; the records must not notice."""
    fs = pkg.synth.make_frames(1500, 256, seed=11)
    p = pkg.default_params()
    p.min_gap = 30

    def run(chunk, packed=0, scratch_mb=0):
        with pkg.Matcher(p) as m:
            m.set_tuning(pkg.capi.TUNE_ITEM_SLOTS, chunk)               # 0 = automatic
            m.set_tuning(pkg.capi.TUNE_PACKED, packed)
            if scratch_mb:
                m.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, scratch_mb)
            for f in range(fs.n_frames):
                m.append(int(fs.ids[f]), fs.frame(f))
            n, offs = m.all_vs_all_plan()
            d = m.dev_alloc(n * 8)
            m.all_vs_all(d, n)
            out = np.zeros(n, pkg.capi.SCORE_DTYPE)
            m.sync()
            m.dev_download(d, out)
            m.dev_free(d)
            return out, offs, m.launch_info().launches

    one, offs, l1 = run(0)
    many, offs2, l2 = run(1)                                   # 1,081,185 single-pair items -> 2 launches
    assert len(one) == pkg.synth.n_pairs_all_vs_all(1500, 30) == 1081185
    assert l1 == 1 and l2 == 2
    np.testing.assert_array_equal(one, many)
    np.testing.assert_array_equal(offs, offs2)
    # the automatic plan packs these 256-row frames eight to a 2048-row workgroup (half the lane slots of the 64 x 8
    # shape are idle otherwise); a chunk takes half of the per-row scratch (consecutive chunks alternate between the halves
    # and between two streams): with 16 GiB a chunk holds 2^20 pairs -> three chunks of (score, fold) launches (the frames
    # whose pairs fit one chunk form the first group, met in two slot ranges — the first one half-size, so that the two
    # streams stay out of step —, the remaining frames the second); with the default 1 GiB 2^16 pairs -> seventeen or
    # more; with 64 MiB 4096 pairs per chunk
    packed, offs3, l3 = run(0, -1, 16384)
    assert l3 == 6
    np.testing.assert_array_equal(one, packed)
    np.testing.assert_array_equal(offs, offs3)
    for mb, lo in ((0, 2 * 17), (64, 2 * 264)):
        packed, offs3, l3 = run(0, -1, mb)
        assert l3 >= lo and l3 % 2 == 0, (mb, l3)
        np.testing.assert_array_equal(one, packed)
        np.testing.assert_array_equal(offs, offs3)
    rng = np.random.default_rng(3)
    qs = rng.integers(30, 1500, 300)
    ts = np.array([rng.integers(0, q - 29) for q in qs])
    cpu, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, qs, ts, oracle.default_params(min_gap=30), n_threads=8)
    np.testing.assert_array_equal(one[offs[qs].astype(np.int64) + ts], cpu)
