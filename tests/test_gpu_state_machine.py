"""Randomised API sequences against a Python model of the database (K10 / K11 in the time dimension): appends (host
and device rows), clear, snapshot round trips, parameter / kernel-variant / tuning changes, bulk searches (records,
index checksums, fused loop test), online queries with several tickets in flight and appends between them, stored-frame
detectLoops and match lists — every result compared with the oracle on the model's frames.  What this hunts: state that
outlives its validity (cached plans, operand images, tickets, staging buffers, stream ordering between the handle's
stream, the copy stream and the query slots' streams)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAX_ROWS = 140


class Model:
    def __init__(self):
        self.ids, self.frames = [], []

    def arrays(self, extra=()):
        fr = self.frames + list(extra)
        rows = np.zeros((max(len(fr), 1), MAX_ROWS, 32), np.uint8)
        counts = np.zeros(max(len(fr), 1), np.int32)
        for i, f in enumerate(fr):
            rows[i, : len(f)] = f
            counts[i] = len(f)
        return rows, counts

    def elig(self, qid, gap):
        return [i for i, s in enumerate(self.ids) if qid - s >= max(gap, 1)]


def _frame(rng, alphabet):
    n = int(rng.choice([0, 1, 3, 17, 64, 65, 100, MAX_ROWS]))
    f = alphabet[rng.integers(0, len(alphabet), n)].copy()
    if n and rng.random() < 0.5:
        f[rng.integers(0, n), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return f


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_api_sequences_equal_oracle(pkg, oracle, tmp_path, seed):
    rng = np.random.default_rng(9000 + seed)
    alphabet = rng.integers(0, 256, (int(rng.integers(2, 9)), 32), dtype=np.uint8)
    model = Model()
    st = dict(gap=2, cross=0, variant=0)
    next_id = 0
    pending = []                                  # (ticket, kind, query frames, query ids, model size at submit)
    with pkg.Matcher() as m:
        m.set_params(min_gap=st["gap"], min_matches=1, sim_threshold=0.0)
        d_frame = m.dev_alloc(MAX_ROWS * 32)

        def params():
            return oracle.default_params(min_gap=st["gap"], min_matches=1, sim_threshold=0.0, cross_check=st["cross"])

        def want_query(q, qid, n_stored):
            rows, counts = model.arrays([q])
            el = [i for i in model.elig(qid, st["gap"]) if i < n_stored]
            sc, _ = oracle.fast_score_pairs_idx(rows, counts, [len(model.frames)] * len(el), el, params(), n_threads=2)
            return sc

        def collect_all():
            while pending:
                t, kind, qs, qids, n_stored, p_at = pending.pop(int(rng.integers(0, len(pending))))
                saved = (st["gap"], st["cross"])
                st["gap"], st["cross"] = p_at                        # a ticket's records follow the parameters at submit
                if kind == "one":
                    sc, ids = m.query_collect(t)
                    np.testing.assert_array_equal(sc, want_query(qs[0], qids[0], n_stored))
                    assert ids.tolist() == [model.ids[i] for i in model.elig(qids[0], st["gap"]) if i < n_stored]
                else:
                    sc, offs = m.query_collect_batch(t)
                    for k, (q, qid) in enumerate(zip(qs, qids)):
                        np.testing.assert_array_equal(sc[int(offs[k]): int(offs[k + 1])], want_query(q, qid, n_stored))
                st["gap"], st["cross"] = saved

        ran = {}
        for step in range(300):
            op = rng.choice(["append", "append", "append", "append_dev", "bulk", "bulk", "loops", "query", "submit", "batch",
                             "collect", "detect", "match", "params", "variant", "tuning", "snapshot", "clear", "truncate"])
            op = str(op)
            ran[op] = ran.get(op, 0) + 1
            if op in ("append", "append_dev"):
                f = _frame(rng, alphabet)
                next_id += int(rng.integers(1, 4))
                if op == "append_dev" and len(f):
                    m.dev_upload(d_frame, np.ascontiguousarray(f))
                    m.append_device(next_id, d_frame, len(f))
                    m.sync()                                          # d_frame is reused by the next device append
                else:
                    m.append(next_id, f)
                model.ids.append(next_id); model.frames.append(f)
            elif op == "clear" and rng.random() < 0.3:
                collect_all()
                m.clear(); model.ids.clear(); model.frames.clear()
            elif op == "truncate" and len(model.frames) >= 2:
                collect_all()                                         # (tickets submitted before a truncate are void)
                keep = int(rng.integers(1, len(model.frames)))
                m.truncate(keep)
                del model.ids[keep:]; del model.frames[keep:]
                next_id = max(next_id, model.ids[-1])
                assert len(m) == keep
            elif op == "snapshot" and model.frames:
                collect_all()
                path = str(tmp_path / f"db_{seed}.bin")
                m.save(path)
                if rng.random() < 0.5:
                    m.clear()
                m.load(path)
                assert len(m) == len(model.frames)
            elif op == "params":
                collect_all()                                         # (a ticket is compared under the parameters at submit)
                st["gap"] = int(rng.integers(0, 5))
                st["cross"] = int(rng.choice([0, 0, 0, 1, 2]))
                m.set_params(min_gap=st["gap"], cross_check=st["cross"])
            elif op == "variant":
                st["variant"] = int(rng.choice([0, 0, 1, 4, 5]))
                m.set_kernel_variant(st["variant"])
            elif op == "tuning":
                m.set_tuning(pkg.capi.TUNE_PACKED, int(rng.choice([-1, 0, 1, 2])))
                m.set_tuning(pkg.capi.TUNE_ITEM_SLOTS, int(rng.choice([0, 1, 2, 5])))
                m.set_tuning(pkg.capi.TUNE_ONLINE_SPLIT, int(rng.choice([-1, 0, 1, 2, 4, 16, 32])))
                m.set_tuning(pkg.capi.TUNE_ONLINE_STREAMS, int(rng.choice([0, 1])))
                m.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, int(rng.choice([1, 2, 1024])))     # 1 MiB: 64 pairs per chunk -> 2-D chunks
                m.set_tuning(pkg.capi.TUNE_PAIR_UPLOAD_KERNEL, int(rng.choice([0, 1])))
                m.set_tuning(pkg.capi.TUNE_PAIR_HOST_FOLD, int(rng.choice([0, 1])))
            elif op in ("bulk", "loops") and model.frames:
                rows, counts = model.arrays()
                pq, pt, offs = [], [], [0]
                for c, cid in enumerate(model.ids):
                    for i in model.elig(cid, st["gap"]):
                        pq.append(c); pt.append(i)
                    offs.append(len(pq))
                want, wsums = oracle.fast_score_pairs_idx(rows, counts, pq, pt, params(), n_threads=2)
                n, goffs = m.all_vs_all_plan()
                assert n == len(pq) and goffs.astype(np.int64).tolist() == offs
                if n == 0:
                    continue
                if op == "bulk":
                    d, ds = m.dev_alloc(n * 8), m.dev_alloc(n * 4)
                    got, sums = np.zeros(n, want.dtype), np.zeros(n, np.uint32)
                    m.all_vs_all(d, n); m.sync(); m.dev_download(d, got)
                    np.testing.assert_array_equal(got, want, err_msg=f"step {step} {st}")
                    m.all_vs_all_argmin(d, n, ds); m.sync(); m.dev_download(d, got); m.dev_download(ds, sums)
                    m.dev_free(d); m.dev_free(ds)
                    np.testing.assert_array_equal(got, want, err_msg=f"step {step} argmin {st}")
                    np.testing.assert_array_equal(sums, wsums, err_msg=f"step {step} checksums {st}")
                else:
                    cands, npairs = m.all_vs_all_loops(cap=n)
                    keep = [(model.ids[pq[k]], model.ids[pt[k]], int(want[k]["good_count"])) for k in range(n)
                            if oracle.loop_test(int(want[k]["good_count"]), int(counts[pq[k]]), int(counts[pt[k]]), params())[0]]
                    assert npairs == n
                    assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"])) for r in cands] == keep
            elif op == "query":
                q = _frame(rng, alphabet)
                qid = next_id + int(rng.integers(-3, 6))
                sc, ids = m.query_scores(q, qid)
                np.testing.assert_array_equal(sc, want_query(q, qid, len(model.frames)), err_msg=f"step {step} {st}")
            elif op in ("submit", "batch") and len(pending) < 3:        # the synchronous calls below need the fourth slot
                k = 1 if op == "submit" else int(rng.integers(1, 5))
                qs = [_frame(rng, alphabet) for _ in range(k)]
                qids = [next_id + 1 + j for j in range(k)]            # ascending; may or may not span min_gap: each query only
                if op == "submit":                                    # ever sees the STORED frames, never its batch mates
                    t = m.query_submit(qs[0], qids[0])
                    pending.append((t, "one", qs, qids, len(model.frames), (st["gap"], st["cross"])))
                else:
                    t = m.query_submit_batch(qs, qids)
                    pending.append((t, "batch", qs, qids, len(model.frames), (st["gap"], st["cross"])))
            elif op == "collect":
                collect_all()
            elif op == "detect" and model.frames:
                cur = int(rng.integers(0, len(model.frames)))
                got = m.detect_loops(model.ids[cur])
                rows, counts = model.arrays()
                wc = oracle.detect_loops(rows, counts, np.array(model.ids, np.int32), cur, params())
                for f in ("matched_frame_id", "num_matches", "similarity_score"):
                    np.testing.assert_array_equal(got[f], wc[f], err_msg=f"step {step} {st}")
            elif op == "match" and len(model.frames) >= 2:
                a, b = (int(x) for x in rng.integers(0, len(model.frames), 2))
                got, md = m.match_stored(model.ids[a], model.ids[b])
                om, omd = oracle.match_features(model.frames[a], model.frames[b], params())
                assert md == omd or len(om) == 0
                np.testing.assert_array_equal(got, om.astype(got.dtype), err_msg=f"step {step} {st}")
        collect_all()
        m.dev_free(d_frame)
        print(f"seed {seed}: {sorted(ran.items())}, {len(model.frames)} frames at the end")
        assert all(ran.get(k, 0) > 0 for k in ("append", "bulk", "loops", "query", "submit", "batch", "detect", "match", "params", "variant", "tuning"))


@pytest.mark.parametrize("world,seed", [(2, 0), (3, 1), (8, 2), (3, 3)])
def test_random_group_sequences_equal_single_handle(pkg, oracle, world, seed):
    """The same idea for the multi-device path: a loopback group of W shards (W matchers on the one device, exchange steps
    as device-local copies) and ONE matcher receive the same random calls — appends of frames of changing size (the
    shard arenas must keep one geometry), clear, parameter changes, bulk searches, online single / micro-batch queries,
    detectLoops — and must answer with the same bytes; a sample of bulk records is pinned to the oracle."""
    rng = np.random.default_rng(7000 + seed)
    alphabet = rng.integers(0, 256, (int(rng.integers(2, 9)), 32), dtype=np.uint8)
    model = Model()
    gap, next_id = 2, 0
    p0 = pkg.default_params()
    p0.min_gap, p0.min_matches, p0.sim_threshold = gap, 1, 0.0
    with pkg.Group(p0, n_devices=world, loopback_device=0) as g, pkg.Matcher(p0) as m:
        ran = {}
        for step in range(150):
            op = str(rng.choice(["append", "append", "append", "append", "bulk", "bulk", "argmin", "loops", "query", "batch", "async", "detect",
                                 "params", "clear", "truncate"]))
            ran[op] = ran.get(op, 0) + 1
            if op == "append":
                f = _frame(rng, alphabet)
                next_id += int(rng.integers(1, 4))
                g.append(next_id, f); m.append(next_id, f)
                model.ids.append(next_id); model.frames.append(f)
            elif op == "clear" and rng.random() < 0.25:
                g.clear(); m.clear(); model.ids.clear(); model.frames.clear()
            elif op == "truncate" and len(model.frames) >= 2:
                keep = int(rng.integers(1, len(model.frames)))
                g.truncate(keep); m.truncate(keep)
                del model.ids[keep:]; del model.frames[keep:]
                next_id = max(next_id, model.ids[-1])
                assert len(g) == keep
            elif op in ("argmin", "loops") and model.frames:
                n, moffs = m.all_vs_all_plan()
                if op == "argmin":
                    gs, gi, goffs = g.all_vs_all_argmin()
                    assert len(gs) == n
                    if n:
                        d, ds = m.dev_alloc(n * 8), m.dev_alloc(n * 4)
                        ss, si = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
                        m.all_vs_all_argmin(d, n, ds); m.sync(); m.dev_download(d, ss); m.dev_download(ds, si)
                        m.dev_free(d); m.dev_free(ds)
                        np.testing.assert_array_equal(gs, ss, err_msg=f"step {step} world {world}")
                        np.testing.assert_array_equal(gi, si, err_msg=f"step {step} world {world} (index checksums)")
                elif n:
                    a, na = g.all_vs_all_loops(cap=n)
                    b, nb = m.all_vs_all_loops(cap=n)
                    assert na == nb == n
                    np.testing.assert_array_equal(a, b, err_msg=f"step {step} world {world}")
            elif op == "async":
                # several group tickets in flight with appends between submit and collect, collected in submit order
                tickets = []
                for _ in range(int(rng.integers(1, 4))):
                    k = int(rng.integers(1, 5))
                    qs = [_frame(rng, alphabet) for _ in range(k)]
                    qids = [next_id + 1 + j for j in range(k)]
                    tg = g.query_submit_batch(qs, qids)
                    tm = m.query_submit_batch(qs, qids)
                    tickets.append((tg, tm, k))
                    if rng.random() < 0.6:
                        f = _frame(rng, alphabet)
                        next_id += int(rng.integers(1, 4))
                        g.append(next_id, f); m.append(next_id, f)
                        model.ids.append(next_id); model.frames.append(f)
                for tg, tm, k in tickets:
                    b, ob = m.query_collect_batch(tm)
                    a, oa = g.query_collect_batch(tg, max(len(model.frames), 1) * k, k)
                    np.testing.assert_array_equal(oa[: k + 1], ob)
                    np.testing.assert_array_equal(a, b, err_msg=f"step {step} world {world}")
            elif op == "params":
                gap = int(rng.integers(0, 5))
                p = pkg.default_params()
                p.min_gap, p.min_matches, p.sim_threshold = gap, 1, 0.0
                g.set_params(p); m.set_params(min_gap=gap)
            elif op == "bulk":
                merged, offs = g.all_vs_all()
                n, moffs = m.all_vs_all_plan()
                assert len(merged) == n and np.array_equal(np.asarray(offs, np.int64), moffs.astype(np.int64)), f"step {step}"
                if n:
                    d = m.dev_alloc(n * 8)
                    single = np.zeros(n, pkg.capi.SCORE_DTYPE)
                    m.all_vs_all(d, n); m.sync(); m.dev_download(d, single); m.dev_free(d)
                    np.testing.assert_array_equal(merged, single, err_msg=f"step {step} world {world}")
                    rows, counts = model.arrays()
                    pq, pt = [], []
                    for c, cid in enumerate(model.ids):
                        el = model.elig(cid, gap)
                        if el:
                            pq.append(c); pt.append(el[int(rng.integers(0, len(el)))])
                    want, _ = oracle.fast_score_pairs_idx(rows, counts, pq, pt, oracle.default_params(min_gap=gap), n_threads=2)
                    np.testing.assert_array_equal(merged[moffs[pq].astype(np.int64) + np.array(pt, np.int64)], want)
            elif op == "query":
                q = _frame(rng, alphabet)
                qid = next_id + int(rng.integers(-3, 6))
                a, ia = g.query_scores(q, qid)
                b, ib = m.query_scores(q, qid)
                np.testing.assert_array_equal(a, b, err_msg=f"step {step}")
                np.testing.assert_array_equal(ia, ib)
            elif op == "batch":
                k = int(rng.integers(1, 6))
                qs = [_frame(rng, alphabet) for _ in range(k)]
                qids = [next_id + 1 + j for j in range(k)]
                a, oa = g.query_scores_batch(qs, qids)
                t = m.query_submit_batch(qs, qids)
                b, ob = m.query_collect_batch(t)
                np.testing.assert_array_equal(oa, ob)
                np.testing.assert_array_equal(a, b, err_msg=f"step {step}")
            elif op == "detect" and model.frames:
                q = _frame(rng, alphabet)
                qid = next_id + 2
                a = g.detect_loops(qid, q)
                b = m.detect_loops(qid, q)
                for f in ("matched_frame_id", "num_matches", "similarity_score"):
                    np.testing.assert_array_equal(a[f], b[f], err_msg=f"step {step}")
        assert all(ran.get(k, 0) > 0 for k in ("append", "bulk", "argmin", "loops", "query", "batch", "async", "detect", "params"))
