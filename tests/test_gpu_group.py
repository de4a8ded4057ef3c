"""lcm_group_*: the multi-GPU loop search behind the C ABI (one process, one handle + host thread per device, RCCL).
The GPU box has ONE device, so the group runs with n_devices = 1 — every step is exercised (ncclCommInitAll, the
all-gather of the shard arena into the query buffer, the rank-major query addressing, the device merge, the single
download); the W > 1 merge kernel is driven on one GPU with shards produced one after the other."""
import numpy as np
import pytest

from conftest import fast_all_vs_all

pytestmark = pytest.mark.gpu


def test_group_of_one_equals_single_handle(pkg, oracle):
    fs = pkg.synth.make_frames(60, 700, seed=12, ragged=True, dup_frac=0.3)
    fs.counts[7] = 0
    p = pkg.default_params()
    p.min_gap = 4
    with pkg.Group(p, n_devices=1) as g, pkg.Matcher(p) as m:
        assert g.world == 1
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
            m.append(int(fs.ids[f]), fs.frame(f))
        assert len(g) == 60
        merged, offs = g.all_vs_all()
        n, moffs = m.all_vs_all_plan()
        d = m.dev_alloc(n * 8)
        m.all_vs_all(d, n)
        single = np.zeros(n, pkg.capi.SCORE_DTYPE)
        m.sync()
        m.dev_download(d, single)
        m.dev_free(d)
        np.testing.assert_array_equal(merged, single)
        np.testing.assert_array_equal(offs, moffs)
        want, _ = fast_all_vs_all(oracle, fs, oracle.default_params(min_gap=4))
        np.testing.assert_array_equal(merged, want)
        gi = g.info()
        assert gi.n_devices == 1 and gi.pairs == n and gi.kernel_ms_max > 0 and gi.gathered_query_bytes > 0
        # a second call re-uses the plan and the buffers; appending more frames invalidates both
        again, _ = g.all_vs_all()
        np.testing.assert_array_equal(again, single)
        extra = pkg.synth.make_frames(3, 700, seed=13)
        for k in range(3):
            g.append(1000 + k, extra.frame(k))
        more, moffs2 = g.all_vs_all()
        assert len(more) == n + 3 * 60 and int(moffs2[60]) == n
        np.testing.assert_array_equal(more[:n], single)
        # online path through the group
        q = extra.frame(1)
        scores, ids = g.query_scores(q, 5000)
        s2, ids2 = m.query_scores(q, 5000)
        np.testing.assert_array_equal(scores[:60], s2)
        assert ids.tolist()[:60] == ids2.tolist() and len(ids) == 63
        # micro-batched online queries through the group == the single handle's batch == one by one
        qb = [extra.frame(k) for k in range(3)]
        bs, boffs = g.query_scores_batch(qb, [5000, 5001, 5002])
        t = m.query_submit_batch(qb, [5000, 5001, 5002])
        ms, moffs_b = m.query_collect_batch(t)
        assert boffs.tolist() == [0, 63, 126, 189]
        for k in range(3):
            np.testing.assert_array_equal(bs[int(boffs[k]): int(boffs[k + 1])][:60], ms[int(moffs_b[k]): int(moffs_b[k + 1])])
        np.testing.assert_array_equal(bs[63:126], scores)
        c1 = g.detect_loops(5000, q)
        want_c = [i for i in range(63) if m.loop_test(scores[i], len(q), int(fs.counts[i]) if i < 60 else 700)[0]]
        assert c1["matched_frame_id"].tolist() == [int(ids[i]) for i in want_c]
        g.clear()
        assert len(g) == 0 and g.all_vs_all()[0].shape == (0,)


def test_group_rejects_bad_device_lists(pkg):
    p = pkg.default_params()
    n = pkg.load_library().lcm_device_count()
    for kw in (dict(n_devices=0), dict(n_devices=9), dict(n_devices=1, device_ids=[n]), dict(n_devices=2, device_ids=[0, 0])):
        with pytest.raises(pkg.LcmError) as e:
            pkg.Group(p, **kw)
        assert e.value.code == -1


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_merge_of_w_shards_equals_host_merge_and_single_device(matcher, pkg, world):
    """k_merge_shards with W > 1: the shards are scored one after the other on the one GPU (exactly what each device of
    a W-GPU group computes: external rank-major query set, owned frames only), laid back to back in device memory as
    the ncclSend / ncclRecv gather would leave them, and merged on the device."""
    fs = pkg.synth.make_frames(45, 400, seed=70 + world, ragged=True, dup_frac=0.2)
    gap = 3
    matcher.set_params(min_gap=gap)
    d_rows = matcher.dev_alloc(fs.rows.nbytes)
    d_counts = matcher.dev_alloc(fs.counts.nbytes)
    try:
        matcher.dev_upload(d_rows, fs.rows)
        matcher.dev_upload(d_counts, fs.counts)
        fb = fs.stride_rows * 32
        kw = dict(d_query_rows=d_rows, d_query_counts=d_counts, q_ids=fs.ids, q_stride_rows=fs.stride_rows)

        def run(positions):
            matcher.clear()
            for f in positions:
                matcher.append_device(int(fs.ids[f]), d_rows + int(f) * fb, int(fs.counts[f]))
            n, _ = matcher.all_vs_all_plan(**kw)
            out = np.zeros(max(n, 1), pkg.capi.SCORE_DTYPE)
            if n:
                d = matcher.dev_alloc(n * 8)
                matcher.all_vs_all(d, n, **kw)
                matcher.sync()
                matcher.dev_download(d, out)
                matcher.dev_free(d)
            return out[:n]

        single = run(range(fs.n_frames))
        shards = [run(pkg.sharding.owned_positions(fs.n_frames, r, world)) for r in range(world)]
        host, _ = pkg.capi.merge_shard_scores_host(shards, fs.ids, gap)
        np.testing.assert_array_equal(host, single)
        total = len(single)
        d_gath, d_merged = matcher.dev_alloc(total * 8), matcher.dev_alloc(total * 8)
        matcher.dev_upload(d_gath, np.concatenate(shards))
        n = matcher.merge_shards_device(d_gath, [len(s) for s in shards], fs.ids, gap, d_merged, total)
        assert n == total
        got = np.zeros(total, pkg.capi.SCORE_DTYPE)
        matcher.sync()
        matcher.dev_download(d_merged, got)
        matcher.dev_free(d_gath); matcher.dev_free(d_merged)
        np.testing.assert_array_equal(got, single)
    finally:
        matcher.dev_free(d_rows); matcher.dev_free(d_counts)
        matcher.set_params(min_gap=30)
        matcher.clear()


@pytest.mark.parametrize("mode", ["cross1", "cross2", "variant4", "variant5"])
def test_group_honours_cross_check_and_kernel_variants(pkg, oracle, mode):
    """The group's search reads its queries from the rank-major gathered buffer (an EXTERNAL query set with an index
    map): cross_check then works on a padded copy of it, the matrix-core variants on an expanded image of it."""
    fs = pkg.synth.make_frames(26, 600, seed=33, ragged=True, dup_frac=0.4)
    fs.counts[5] = 0
    fs.rows[7, :100] = fs.rows[3, :100]
    p = pkg.default_params()
    p.min_gap = 3
    op = oracle.default_params(min_gap=3)
    if mode.startswith("cross"):
        p.cross_check = int(mode[-1])
        op.cross_check = p.cross_check
    with pkg.Group(p, n_devices=1) as g:
        if mode.startswith("variant"):
            import ctypes as C
            h = C.c_void_p()
            assert g._lib.lcm_group_handle(g._g, 0, C.byref(h)) == 0
            assert g._lib.lcm_set_kernel_variant(h, int(mode[-1])) == 0
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
        merged, offs = g.all_vs_all()
        want, woffs = fast_all_vs_all(oracle, fs, op)
        np.testing.assert_array_equal(offs.astype(np.int64), woffs)
        np.testing.assert_array_equal(merged, want)
        sc, ids = g.query_scores(fs.frame(20), 500)
        pq = [20] * fs.n_frames
        wq, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, pq, list(range(fs.n_frames)), op, n_threads=8)
        np.testing.assert_array_equal(sc, wq)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_loopback_group_of_w_shards_equals_single_handle(pkg, oracle, world):
    """lcm_group_create_loopback: W shards on the ONE device, exchange steps as device-local copies.  Everything the
    multi-device path computes for W > 1 runs here on real kernels — cyclic ownership, shard arenas of equal geometry,
    the rank-major gathered query buffer and its q_frame_of addressing, W host threads planning and launching, the
    gatherv offsets, k_merge_shards — and must give the single handle's bytes (== oracle).  Only RCCL's transport is not
    in it.  Also the online group calls, the packed bulk route per shard (2000-row frames), and an uneven shard split."""
    n_frames = 8 * 23 + 5                                         # not a multiple of any world size used
    fs = pkg.synth.make_frames(n_frames, 2000, seed=70 + world, ragged=True, dup_frac=0.3)
    fs.counts[7] = 0
    fs.counts[11] = 1
    p = pkg.default_params()
    p.min_gap = 3
    with pkg.Group(p, n_devices=world, loopback_device=0) as g, pkg.Matcher(p) as m:
        assert g.world == world
        g.reserve(n_frames, 2000)
        for f in range(n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
            m.append(int(fs.ids[f]), fs.frame(f))
        merged, offs = g.all_vs_all()
        n, moffs = m.all_vs_all_plan()
        d = m.dev_alloc(n * 8)
        m.all_vs_all(d, n)
        single = np.zeros(n, pkg.capi.SCORE_DTYPE)
        m.sync(); m.dev_download(d, single); m.dev_free(d)
        np.testing.assert_array_equal(offs, moffs)
        np.testing.assert_array_equal(merged, single)
        rng = np.random.default_rng(world)
        pq, pt = [], []
        for _ in range(160):
            c = int(rng.integers(3, n_frames)); t = int(rng.integers(0, c - 2))
            pq.append(c); pt.append(t)
        want, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, pq, pt, oracle.default_params(min_gap=3), n_threads=8)
        np.testing.assert_array_equal(merged[offs[pq].astype(np.int64) + np.array(pt)], want)
        gi = g.info()
        assert gi.n_devices == world and gi.pairs == n and gi.gathered_score_bytes == (n - sum(1 for c in range(n_frames) for t in range(max(c - 2, 0)) if t % world == 0)) * 8
        # a second search (cached plans on every shard), then more frames
        again, _ = g.all_vs_all()
        np.testing.assert_array_equal(again, single)
        # online through the group: single query, micro-batch, detectLoops
        q = fs.frame(20)
        sc, ids = g.query_scores(q, int(fs.ids[-1]) + 3)
        s1, ids1 = m.query_scores(q, int(fs.ids[-1]) + 3)
        np.testing.assert_array_equal(sc, s1)
        np.testing.assert_array_equal(ids, ids1)
        qb = [fs.frame(5), fs.frame(7), fs.frame(40)]
        qids = [int(fs.ids[-1]) + 3, int(fs.ids[-1]) + 4, 60]    # the last one sees only a prefix of the database
        bs, boffs = g.query_scores_batch(qb, qids)
        t = m.query_submit_batch(qb, qids)
        ms, mo = m.query_collect_batch(t)
        np.testing.assert_array_equal(boffs, mo)
        np.testing.assert_array_equal(bs, ms)
        c1 = g.detect_loops(int(fs.ids[-1]) + 3, q)
        c2 = m.detect_loops(int(fs.ids[-1]) + 3, q)
        np.testing.assert_array_equal(c1["matched_frame_id"], c2["matched_frame_id"])
        np.testing.assert_array_equal(c1["num_matches"], c2["num_matches"])
