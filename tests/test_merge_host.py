"""lcm_merge_shard_scores — the host-only C function that un-permutes W cyclic shards' score arrays (what
lcm_group_all_vs_all does on the device with k_merge_shards) — against the numpy merge the process-per-GPU path uses
(sharding.merge_shard_scores).  No GPU involved: this is index arithmetic."""
import numpy as np
import pytest


def _shards(pkg, ids, gap, world, rng):
    """Synthetic per-shard arrays whose records encode (query, stored) so that any misplacement is visible."""
    e = pkg.sharding.eligible_counts(ids, gap)
    full = np.zeros(int(e.sum()), pkg.capi.SCORE_DTYPE)
    k = 0
    for c in range(len(ids)):
        for t in range(int(e[c])):
            full[k] = (c * 100003 + t, t % 65521, c % 65521)
            k += 1
    offs = pkg.sharding.offsets_from_counts(e)
    shards = []
    for r in range(world):
        parts = [full[int(offs[c]) + r: int(offs[c + 1]): world] for c in range(len(ids))]
        shards.append(np.concatenate(parts) if parts else np.zeros(0, pkg.capi.SCORE_DTYPE))
    return full, offs, shards


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("n,gap,step", [(0, 3, 1), (1, 1, 1), (7, 30, 1), (40, 3, 1), (57, 1, 1), (33, 4, 3), (64, 0, 1)])
def test_host_merge_equals_numpy_merge(pkg, world, n, gap, step):
    rng = np.random.default_rng(n * 31 + world)
    ids = (np.cumsum(rng.integers(1, step + 1, n)) if n else np.zeros(0)).astype(np.int32)
    full, offs, shards = _shards(pkg, ids, gap, world, rng)
    got, goffs = pkg.capi.merge_shard_scores_host(shards, ids, gap)
    np.testing.assert_array_equal(got, full)
    np.testing.assert_array_equal(goffs.astype(np.int64), offs)
    want, woffs = pkg.sharding.merge_shard_scores(shards, ids, gap)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(goffs.astype(np.int64), woffs)


def test_host_merge_rejects_wrong_shard_sizes_and_unordered_ids(pkg):
    ids = np.arange(20, dtype=np.int32)
    _, _, shards = _shards(pkg, ids, 2, 3, None)
    bad = [shards[0], shards[1][:-1], shards[2]]
    with pytest.raises(pkg.LcmError) as e:
        pkg.capi.merge_shard_scores_host(bad, ids, 2)
    assert e.value.code == -1 and "shard 1" in str(e.value)
    with pytest.raises(pkg.LcmError) as e:
        pkg.capi.merge_shard_scores_host(shards, ids[::-1].copy(), 2)
    assert e.value.code == -5
