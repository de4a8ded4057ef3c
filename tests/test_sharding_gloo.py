"""N > 1 host logic on CPU: world_size-2 gloo processes, each scoring its cyclic shard (the CPU oracle stands in for
the HIP scorer — this test covers ownership, offsets, the all-gather of variable-length score arrays and the merge,
not the kernel)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import load_oracle, load_package

    pkg, orc = load_package(), load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gap = 3
        fs = pkg.synth.make_frames(23, 48, seed=99, ragged=True, dup_frac=0.4)
        fs.counts[5] = 0
        p = orc.default_params(min_gap=gap)
        # this rank's shard scores, in (query asc, owned stored asc) order — what lcm_all_vs_all writes on a GPU rank
        local, _ = orc.all_vs_all(fs.rows, fs.counts, fs.ids, p, rank, world)
        expect_n = int(pkg.sharding.shard_eligible_counts(fs.ids, gap, rank, world).sum())
        assert len(local) == expect_n
        t = torch.from_numpy(local.view(np.int64).copy()) if len(local) else torch.zeros(0, dtype=torch.int64)
        shards = pkg.sharding.all_gather_scores(t, len(local))
        merged, offs = pkg.sharding.merge_shard_scores(shards, fs.ids, gap)
        full, foffs = orc.all_vs_all(fs.rows, fs.counts, fs.ids, p)
        assert np.array_equal(merged, full)
        assert np.array_equal(offs, foffs.astype(np.int64))
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([len(merged)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_merge_equals_single(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    vals = [int(np.load(tmp_path / f"ok{r}.npy")[0]) for r in range(world)]
    assert len(set(vals)) == 1 and vals[0] > 0


def test_ownership_and_offsets(pkg):
    sh = pkg.sharding
    ids = np.arange(0, 50, 2)                       # ids 0,2,...,48 ; gap 7 -> id difference >= 7
    e = sh.eligible_counts(ids, 7)
    assert e.tolist() == [int(np.sum(ids[c] - ids >= 7)) for c in range(len(ids))]
    for world in (1, 2, 3, 8):
        tot = np.zeros_like(e)
        for r in range(world):
            er = sh.shard_eligible_counts(ids, 7, r, world)
            assert er.tolist() == [int(np.sum((ids[c] - ids >= 7) & (np.arange(len(ids)) % world == r))) for c in range(len(ids))]
            tot += er
            assert sh.owned_positions(len(ids), r, world).tolist() == list(range(r, len(ids), world))
        assert tot.tolist() == e.tolist()
    assert sh.eligible_counts(ids, 0).tolist() == list(range(len(ids)))      # gap 0 still never pairs a frame with itself


class _OracleScorer:
    """CPU stand-in for Matcher in the host-logic tests: same interface, distances from the oracle."""

    def __init__(self, oracle, params):
        self.oracle, self.params, self.frames = oracle, params, []

    def __len__(self):
        return len(self.frames)

    def append(self, frame_id, rows, n_keypoints=-1):
        self.frames.append((int(frame_id), np.ascontiguousarray(rows)))

    def query_scores(self, rows, frame_id):
        gap = max(int(self.params.min_gap), 1)
        el = [(i, r) for i, r in self.frames if frame_id - i >= gap]
        out = np.zeros(len(el), dtype=[("good_count", "<u4"), ("min_dist", "<u2"), ("n_train", "<u2")])
        for k, (_, r) in enumerate(el):
            out[k] = self.oracle.pair_score(rows, r, self.params)
        return out, np.array([i for i, _ in el], np.int32)


def _online_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_oracle, load_package

    pkg, orc = load_package(), load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fs = pkg.synth.make_frames(18, 40, seed=7, ragged=True, dup_frac=0.5)
        p = orc.default_params(min_gap=2, min_matches=3, sim_threshold=0.05)
        search = pkg.sharding.ShardedLoopSearch(_OracleScorer(orc, p), rank, world)
        all_scores, all_cands = [], []
        for f in range(fs.n_frames):
            merged, ids, cands = search.process_frame(fs.frame(f), int(fs.ids[f]))
            assert ids.tolist() == [int(i) for i in fs.ids if fs.ids[f] - i >= 2]
            all_scores.append(merged)
            all_cands += cands
        full, _ = orc.all_vs_all(fs.rows, fs.counts, fs.ids, p)
        assert np.array_equal(np.concatenate(all_scores), full)
        want = [tuple(map(lambda x: x.item() if hasattr(x, "item") else x, (c["current_frame_id"], c["matched_frame_id"], c["num_matches"], c["similarity_score"])))
                for cur in range(fs.n_frames) for c in orc.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)]
        assert all_cands == want and len(want) > 0
        assert len(search.scorer) == len(range(rank, fs.n_frames, world))          # only owned frames were stored
        np.save(os.path.join(out_dir, f"online{rank}.npy"), np.array([len(want)]))
    finally:
        dist.destroy_process_group()


def test_gloo_online_sharded_search(tmp_path):
    port = _free_port()
    mp.spawn(_online_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "online0.npy")[0]) == int(np.load(tmp_path / "online1.npy")[0]) > 0
