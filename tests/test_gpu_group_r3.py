"""Round-3 surface of lcm_group_*: the argmin search (index checksums gathered and merged like the records), the fused
loop search (loop test on every shard's device, candidates merged in order), asynchronous micro-batched group queries,
the skipped arena all-gather, truncate — each against the single handle (itself checked against the oracle elsewhere
and here by sample).  W > 1 runs as a loopback group on the box's one GPU; the same checks run over REAL devices
when the box has more than one (test_real_multi_device_group)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def single_argmin(pkg, m):
    n, offs = m.all_vs_all_plan()
    d, di = m.dev_alloc(max(n, 1) * 8), m.dev_alloc(max(n, 1) * 4)
    m.all_vs_all_argmin(d, n, di)
    sc, ix = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
    m.sync()
    if n:
        m.dev_download(d, sc); m.dev_download(di, ix)
    m.dev_free(d); m.dev_free(di)
    return sc, ix, offs


def fill(fs, *targets):
    for f in range(fs.n_frames):
        for t in targets:
            t.append(int(fs.ids[f]), fs.frame(f))


def check_group_against_single(pkg, oracle, g, m, fs, gap, world):
    """everything a group offers in bulk, against the single handle holding the same frames"""
    sc, ix, offs = single_argmin(pkg, m)
    gs, gi, goffs = g.all_vs_all_argmin()
    np.testing.assert_array_equal(goffs, offs)
    np.testing.assert_array_equal(gs, sc)
    np.testing.assert_array_equal(gi, ix)
    info = g.info()
    assert info.n_devices == world and info.pairs == len(sc) and info.arena_gather_skipped == 0
    assert sum(info.shard_pairs[r] for r in range(world)) == len(sc)
    assert all(info.kernel_ms[r] > 0 for r in range(world) if info.shard_pairs[r])
    # index checksums are right by value too: oracle sample
    rng = np.random.default_rng(world)
    n = fs.n_frames
    pq = rng.integers(gap, n, size=60)
    pt = np.array([rng.integers(0, q - gap + 1) for q in pq])
    want, wsum = oracle.fast_score_pairs_idx(fs.rows, fs.counts, pq, pt, oracle.default_params(min_gap=gap), 8)
    at = offs[pq].astype(np.int64) + pt
    np.testing.assert_array_equal(gs[at], want)
    np.testing.assert_array_equal(gi[at], wsum)
    # second search: nothing appended -> the arena all-gather is skipped, same bytes
    gs2, gi2, _ = g.all_vs_all_argmin()
    assert g.info().arena_gather_skipped == 1 and g.info().gathered_query_bytes == 0
    np.testing.assert_array_equal(gs2, sc)
    np.testing.assert_array_equal(gi2, ix)
    plain, _ = g.all_vs_all()
    np.testing.assert_array_equal(plain, sc)
    # fused loop search
    want_c, want_pairs = m.all_vs_all_loops(cap=len(sc) + 1)
    got_c, got_pairs = g.all_vs_all_loops(cap=len(sc) + 1)
    assert got_pairs == want_pairs == len(sc)
    np.testing.assert_array_equal(got_c, want_c)
    assert len(want_c) > 0
    if len(want_c) > 1:
        with pytest.raises(pkg.LcmError) as e:
            g.all_vs_all_loops(cap=len(want_c) - 1)
        assert e.value.code == pkg.capi.ERR_CAPACITY


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_group_argmin_and_loops_equal_single_handle(pkg, oracle, world):
    n_frames = 8 * 11 + 5
    fs = pkg.synth.make_frames(n_frames, 2000 if world in (2, 8) else 700, seed=300 + world, ragged=True, dup_frac=0.3)
    fs.counts[9] = 0
    fs.counts[13] = 1
    p = pkg.default_params()
    p.min_gap = 3
    p.min_matches = 20
    kw = dict(n_devices=1) if world == 1 else dict(n_devices=world, loopback_device=0)
    with pkg.Group(p, **kw) as g, pkg.Matcher(p) as m:
        fill(fs, g, m)
        check_group_against_single(pkg, oracle, g, m, fs, 3, world)
        assert g.info().loopback == (0 if world == 1 else 1)
        if world == 1:
            assert g.info().rccl_ranks == 1
        # appending invalidates the gathered buffers; truncating back gives the first result again
        sc, ix, _ = single_argmin(pkg, m)
        extra = pkg.synth.make_frames(5, 700, seed=9)
        for k in range(5):
            g.append(10000 + k, extra.frame(k)); m.append(10000 + k, extra.frame(k))
        gs, gi, _ = g.all_vs_all_argmin()
        assert g.info().arena_gather_skipped == 0
        s2, i2, _ = single_argmin(pkg, m)
        np.testing.assert_array_equal(gs, s2)
        np.testing.assert_array_equal(gi, i2)
        g.truncate(n_frames); m.truncate(n_frames)
        assert len(g) == n_frames and len(m) == n_frames
        gs, gi, _ = g.all_vs_all_argmin()
        np.testing.assert_array_equal(gs, sc)
        np.testing.assert_array_equal(gi, ix)
        s3, i3, _ = single_argmin(pkg, m)
        np.testing.assert_array_equal(s3, sc)
        np.testing.assert_array_equal(i3, ix)
        # ... and frames can be appended again behind the cut
        g.append(20000, extra.frame(0)); m.append(20000, extra.frame(0))
        np.testing.assert_array_equal(g.all_vs_all()[0], np.concatenate([sc, m.query_scores(extra.frame(0), 20000)[0]]))


def test_group_transports_of_one_device(pkg):
    """which exchange transport a group uses is observable; the peer-copy form (the fallback when no RCCL communicator can be
    created) gives the same bytes"""
    fs = pkg.synth.make_frames(40, 600, seed=21, ragged=True, dup_frac=0.2)
    p = pkg.default_params()
    p.min_gap = 2
    res = {}
    for name, kw in (("rccl", dict(n_devices=1)), ("peer", dict(n_devices=1, peer_copies=True)), ("loop", dict(n_devices=2, loopback_device=0))):
        with pkg.Group(p, **kw) as g:
            fill(fs, g)
            res[name] = (g.all_vs_all_argmin()[:2], g.transport, g.info().rccl_ranks)
    assert res["rccl"][1] == "rccl" and res["rccl"][2] == 1
    assert res["peer"][1] == "peer copies (requested)" and res["peer"][2] == 0
    assert res["loop"][1].startswith("loopback") and res["loop"][2] == 0
    for name in ("peer", "loop"):
        np.testing.assert_array_equal(res[name][0][0], res["rccl"][0][0])
        np.testing.assert_array_equal(res[name][0][1], res["rccl"][0][1])


def test_loops_capacity_is_reported_before_any_worst_case_allocation(pkg):
    fs = pkg.synth.make_frames(80, 600, seed=77, dup_frac=0.2)
    p = pkg.default_params()
    p.min_gap = 2
    p.min_matches = 10
    with pkg.Matcher(p) as m:
        fill(fs, m)
        full, pairs = m.all_vs_all_loops(cap=pairs_cap(80, 2))
        assert 1 < len(full) <= pairs
        import ctypes as C
        n, npairs = C.c_size_t(0), C.c_size_t(0)
        tiny = np.zeros(1, pkg.capi.CANDIDATE_DTYPE)
        rc = m._lib.lcm_all_vs_all_loops(m._h, None, None, None, None, 0, 0, tiny.ctypes.data_as(C.c_void_p), 1, C.byref(n), C.byref(npairs))
        assert rc == pkg.capi.ERR_CAPACITY and n.value == len(full) and npairs.value == pairs     # the needed count comes back
        exact = np.zeros(len(full), pkg.capi.CANDIDATE_DTYPE)
        got, _ = m.all_vs_all_loops(out=exact)
        np.testing.assert_array_equal(got, full)


def pairs_cap(n, gap):
    return (n - gap) * (n - gap + 1) // 2 + 1


@pytest.mark.parametrize("world", [1, 3, 8])
def test_async_group_queries_pipeline_equals_single_handle(pkg, world):
    """Streaming through the group the way bench.py --mode stream --gpus N does it: submit batch k + 1, append its frames,
    collect batch k — up to 4 group tickets in flight — against the single handle fed the same way."""
    n_frames, B, gap = 150, 4, 6
    fs = pkg.synth.make_frames(n_frames, 900, seed=400 + world, ragged=True, dup_frac=0.2)
    fs.counts[17] = 0
    p = pkg.default_params()
    p.min_gap = gap
    kw = dict(n_devices=1) if world == 1 else dict(n_devices=world, loopback_device=0)
    with pkg.Group(p, **kw) as g, pkg.Matcher(p) as m:
        got, want, pending = [], [], []
        for depth in (1, 3, 4):
            g.clear(); m.clear(); got.clear(); want.clear()
            for f0 in range(0, n_frames, B):
                fr = list(range(f0, min(f0 + B, n_frames)))
                qs, ids = [fs.frame(f) for f in fr], [int(fs.ids[f]) for f in fr]
                pending.append((g.query_submit_batch(qs, ids), len(fr)))
                t = m.query_submit_batch(qs, ids)
                want.append(m.query_collect_batch(t)[0])
                for f in fr:
                    g.append(int(fs.ids[f]), fs.frame(f)); m.append(int(fs.ids[f]), fs.frame(f))
                if len(pending) == depth:
                    tk, nb = pending.pop(0)
                    sc, offs = g.query_collect_batch(tk, n_frames * nb, nb)
                    assert int(offs[nb]) == len(sc)
                    got.append(sc)
            while pending:
                tk, nb = pending.pop(0)
                got.append(g.query_collect_batch(tk, n_frames * nb, nb)[0])
            np.testing.assert_array_equal(np.concatenate(got), np.concatenate(want))
        st = g.online_stats()
        assert st.pairs == 3 * (n_frames - gap) * (n_frames - gap + 1) // 2 and st.kernel_ms > 0
        # a fifth ticket is refused; a too-small buffer keeps the ticket; clear voids tickets in flight
        qs, ids = [fs.frame(3)], [5000]
        tks = [g.query_submit_batch(qs, ids) for _ in range(4)]
        with pytest.raises(pkg.LcmError) as e:
            g.query_submit_batch(qs, ids)
        assert e.value.code == pkg.capi.ERR_CAPACITY
        with pytest.raises(pkg.LcmError) as e:
            g.query_collect_batch(tks[0], 3, 1)
        assert e.value.code == pkg.capi.ERR_CAPACITY
        a = g.query_collect_batch(tks[0], n_frames, 1)[0]
        np.testing.assert_array_equal(a, m.query_scores(fs.frame(3), 5000)[0])
        g.clear()
        for tk in tks[1:]:
            with pytest.raises(pkg.LcmError) as e:
                g.query_collect_batch(tk, n_frames, 1)
            assert e.value.code == pkg.capi.ERR_NOT_FOUND
        assert g.query_collect_batch(g.query_submit_batch(qs, ids), 4, 1)[0].shape == (0,)       # usable again


@pytest.mark.parametrize("shards", [2, 8])
def test_host_class_over_a_group_equals_one_device(pkg, shards):
    """LoopClosingSystem over a (loopback) group: processFrame and the pipelined processFrames leave exactly what the
    one-device system leaves (frames, loop closures in order, consecutive matches)."""
    fs = pkg.synth.make_frames(90, 500, seed=91, dup_frac=0.3)
    one = pkg.capi.LoopClosingSystem(0.15, 4)
    grp = pkg.capi.LoopClosingSystem(0.15, 4, loopback_shards=shards)
    bat = pkg.capi.LoopClosingSystem(0.15, 4, loopback_shards=shards)
    try:
        for f in range(40):
            one.processFrame(fs.frame(f), int(fs.ids[f]))
            grp.processFrame(fs.frame(f), int(fs.ids[f]))
        one.processFrames([fs.frame(f) for f in range(40, 90)], [int(fs.ids[f]) for f in range(40, 90)])
        grp.processFrames([fs.frame(f) for f in range(40, 90)], [int(fs.ids[f]) for f in range(40, 90)])
        bat.processFrames([fs.frame(f) for f in range(90)], [int(fs.ids[f]) for f in range(90)])
        fields = ["current_frame_id", "matched_frame_id", "num_matches", "similarity_score"]     # (_pad is padding)
        want = one.getLoopClosures()[fields]
        assert len(want) > 0
        np.testing.assert_array_equal(grp.getLoopClosures()[fields], want)
        np.testing.assert_array_equal(bat.getLoopClosures()[fields], want)
        np.testing.assert_array_equal(grp.getConsecutiveMatches(), one.getConsecutiveMatches())
        np.testing.assert_array_equal(bat.getConsecutiveMatches(), one.getConsecutiveMatches())
        np.testing.assert_array_equal(grp.detectLoops(int(fs.ids[80]))[fields], one.detectLoops(int(fs.ids[80]))[fields])
        for a, b in zip(grp.matchLoopClosures(int(want[-1]["current_frame_id"])), one.matchLoopClosures(int(want[-1]["current_frame_id"]))):
            np.testing.assert_array_equal(a, b)
        # a rejected frame (ids must increase) leaves all three unchanged
        with pytest.raises(pkg.LcmError):
            grp.processFrames([fs.frame(1)], [3])
        assert grp.numFrames() == 90
    finally:
        one.close(); grp.close(); bat.close()


def test_truncate_single_handle(pkg, oracle):
    fs = pkg.synth.make_frames(40, 300, seed=5, ragged=True)
    p = pkg.default_params()
    p.min_gap = 2
    with pkg.Matcher(p) as m:
        fill(fs, m)
        t = m.query_submit(fs.frame(3), 900)
        m.truncate(25)
        with pytest.raises(pkg.LcmError) as e:          # submitted against the longer database: void
            m.query_collect(t)
        assert e.value.code == pkg.capi.ERR_NOT_FOUND
        assert len(m) == 25
        m.truncate(30)                                   # more than stored: no-op
        assert len(m) == 25
        sc, ids = m.query_scores(fs.frame(30), 900)
        assert ids.tolist() == [int(x) for x in fs.ids[:25]]
        want, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [30] * 25, list(range(25)), oracle.default_params(min_gap=2), n_threads=4)
        np.testing.assert_array_equal(sc, want)
        m.append(int(fs.ids[25]), fs.frame(25))          # the slot behind the cut is reused
        assert len(m) == 26 and np.array_equal(m.read_frame(25), fs.frame(25))


def test_group_snapshot_is_the_single_handle_format(pkg, tmp_path):
    """lcm_group_save / _load use lcm_db_save's own file format with frames in arrival order: group of 3 -> single handle
    -> group of 5 -> group of 3, every hop with identical frames and identical search records; a corrupt file is
    refused BEFORE the group's contents are touched."""
    fs = pkg.synth.make_frames(41, 500, seed=77, ragged=True, dup_frac=0.3)
    p = pkg.default_params()
    p.min_gap = 2
    a, b, c = str(tmp_path / "a.lcmdb"), str(tmp_path / "b.lcmdb"), str(tmp_path / "c.lcmdb")
    with pkg.Group(p, n_devices=3, loopback_device=0) as g3, pkg.Matcher(p) as m, pkg.Group(p, n_devices=5, loopback_device=0) as g5:
        fill(fs, g3)
        g3.save(a)
        m.load(a)
        assert len(m) == 41
        for s in (0, 17, 40):
            assert m.frame_info(s) == (int(fs.ids[s]), int(fs.counts[s]), int(fs.counts[s]))
            np.testing.assert_array_equal(m.read_frame(s), fs.frame(s))
        m.save(b)
        assert open(a, "rb").read() == open(b, "rb").read()          # one format, byte for byte
        g5.load(b)
        assert len(g5) == 41
        for x, y in zip(g5.all_vs_all_argmin(), single_argmin(pkg, m)):
            np.testing.assert_array_equal(x, y)
        g5.save(c)
        assert open(c, "rb").read() == open(a, "rb").read()
        # loading replaces what the group held; appends continue behind the loaded frames
        g3.load(c)
        g3.append(int(fs.ids[-1]) + 1, fs.frame(0))
        assert len(g3) == 42
        # a truncated file: refused, the group keeps its 42 frames
        bad = str(tmp_path / "bad.lcmdb")
        open(bad, "wb").write(open(a, "rb").read()[:-100])
        with pytest.raises(pkg.LcmError):
            g3.load(bad)
        assert len(g3) == 42
        with pytest.raises(pkg.LcmError) as e:
            g3.load(str(tmp_path / "missing.lcmdb"))
        assert e.value.code == pkg.capi.ERR_NOT_FOUND and len(g3) == 42


def test_real_multi_device_group(pkg, oracle):
    """RCCL's own transport with more than one rank: ncclAllGather of the shard arenas, grouped ncclSend / ncclRecv of the
    records and index checksums, per-device host threads.  Needs a box with >= 2 GPUs (the driver's multi-GPU node);
    skipped on the one-GPU box, where the loopback group rehearses everything but the transport."""
    n_dev = pkg.load_library().lcm_device_count()
    if n_dev < 2:
        pytest.skip("one HIP device: RCCL with > 1 rank cannot run here (loopback group tests cover the arithmetic)")
    fs = pkg.synth.make_frames(8 * 9 + 3, 2000, seed=55, ragged=True, dup_frac=0.3)
    p = pkg.default_params()
    p.min_gap = 3
    p.min_matches = 20
    for world, peer in [(w, t) for w in sorted({2, min(n_dev, 4), min(n_dev, 8)}) for t in (False, True)]:
        with pkg.Group(p, n_devices=world, peer_copies=peer) as g, pkg.Matcher(p) as m:
            fill(fs, g, m)
            check_group_against_single(pkg, oracle, g, m, fs, 3, world)
            assert g.info().loopback == 0 and g.info().gathered_score_bytes > 0
            if peer:
                assert g.transport.startswith("peer copies") and g.info().rccl_ranks == 0
            elif g.transport == "rccl":
                assert g.info().rccl_ranks == world
            else:
                print("NOTE: RCCL communicator not available on this box, the group fell back to:", g.transport)
            q = fs.frame(20)
            np.testing.assert_array_equal(g.query_scores(q, 9000)[0], m.query_scores(q, 9000)[0])
            qb, qids = [fs.frame(5), fs.frame(7), fs.frame(40)], [9000, 9001, 60]
            t = g.query_submit_batch(qb, qids)
            t2 = g.query_submit_batch(qb[:1], qids[:1])
            bs = g.query_collect_batch(t, 3 * len(g), 3)[0]
            ms = m.query_collect_batch(m.query_submit_batch(qb, qids))[0]
            np.testing.assert_array_equal(bs, ms)
            np.testing.assert_array_equal(g.query_collect_batch(t2, len(g), 1)[0], m.query_scores(qb[0], 9000)[0])
            c1, c2 = g.detect_loops(9000, q), m.detect_loops(9000, q)
            np.testing.assert_array_equal(c1, c2)
