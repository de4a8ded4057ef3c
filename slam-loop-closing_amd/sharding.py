"""Frame-sharded multi-GPU loop search: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) for the one real exchange step — gathering per-shard score records.

Partitioning: the stored-frame database is sharded cyclically by frame — stored frame with
position i (ascending id) is OWNED by rank i mod W.  Cyclic rather than blocked because the all-vs-all workload is
triangular (a query only meets frames >= min_gap older) and, in streaming mode, the database grows: blocks would
leave the first rank idle early and the last overloaded.  Every rank sees every QUERY frame (each frame is a query
exactly once; 64 KB), scores it against the frames it owns with no communication, and the 8-byte score records
(good_count, min_dist, n_train) are then all-gathered and un-permuted into ascending stored-frame order, so the
merged result is byte-identical to the single-GPU result.

This module holds only the host logic (ownership, offsets, merge, gather).  The scoring itself is whatever
`Matcher`-like object the caller passes in: the HIP library in production, the CPU oracle in the world_size-2
gloo tests that cover this logic without a GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

SCORE_DTYPE = np.dtype([("good_count", "<u4"), ("min_dist", "<u2"), ("n_train", "<u2")])


def owner_of(position: int, world: int) -> int:
    return position % world


def owned_positions(n_frames: int, rank: int, world: int) -> np.ndarray:
    return np.arange(rank, n_frames, world, dtype=np.int64)


def eligible_counts(ids: Sequence[int], gap: int) -> np.ndarray:
    """e[c] = number of stored frames i with ids[c] - ids[i] >= max(gap, 1); ids strictly increasing."""
    ids = np.asarray(ids, np.int64)
    return np.searchsorted(ids, ids - max(int(gap), 1), side="right").astype(np.int64)


def shard_eligible_counts(ids: Sequence[int], gap: int, rank: int, world: int) -> np.ndarray:
    """Same, counting only the frames owned by `rank`: positions rank, rank + W, ... below e[c]."""
    e = eligible_counts(ids, gap)
    return np.maximum(0, (e - rank + world - 1) // world).astype(np.int64)


def offsets_from_counts(counts: np.ndarray) -> np.ndarray:
    offs = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=offs[1:])
    return offs


def shard_destinations(ids: Sequence[int], gap: int, rank: int, world: int) -> np.ndarray:
    """For every record of rank `rank`'s (query asc, owned stored asc) array: its index in the single-device
    (query asc, stored asc) array.  Query c's k-th owned frame is stored position rank + k*world."""
    e = eligible_counts(ids, gap)
    offs = offsets_from_counts(e)
    er = np.maximum(0, (e - rank + world - 1) // world)
    offr = offsets_from_counts(er)
    c_of = np.repeat(np.arange(len(e)), er)
    k_of = np.arange(int(offr[-1])) - offr[c_of]
    return offs[c_of] + rank + k_of * world


def merge_shard_scores(shard_scores: List[np.ndarray], ids: Sequence[int], gap: int) -> Tuple[np.ndarray, np.ndarray]:
    """Un-permute per-rank score arrays (each in (query asc, owned stored asc) order) into the single-device
    (query asc, stored asc) order.  Returns (scores, offsets[n_frames + 1])."""
    world = len(shard_scores)
    e = eligible_counts(ids, gap)
    offs = offsets_from_counts(e)
    out = np.zeros(int(offs[-1]), SCORE_DTYPE)
    for r in range(world):
        dst = shard_destinations(ids, gap, r, world)
        src = np.asarray(shard_scores[r])
        if dst.shape[0] != src.shape[0]:
            raise ValueError(f"rank {r}: expected {dst.shape[0]} score records, got {src.shape[0]}")
        if src.shape[0]:
            out[dst] = src
    return out, offs


def all_gather_scores(local: "torch.Tensor", n_local: int, group=None) -> List[np.ndarray]:  # noqa: F821
    """All-gather variable-length score arrays.  `local` is an int64 tensor viewing 8-byte records (device tensor
    for nccl/RCCL, CPU tensor for gloo) holding at least n_local records.  One fixed-size all_gather of the padded
    payload (shard sizes differ by at most one frame's worth, so padding is negligible) plus one of the lengths."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_t = torch.tensor([n_local], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n_t) for _ in range(world)]
    dist.all_gather(lens, n_t, group=group)
    lens = [int(x.item()) for x in lens]
    cap = max(max(lens), 1)
    if local.numel() < cap:
        pad = torch.zeros(cap, dtype=torch.int64, device=local.device)
        pad[:n_local] = local[:n_local]
        send = pad
    else:
        send = local[:cap].contiguous()
    recv = torch.empty(world * cap, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group) if hasattr(dist, "all_gather_into_tensor") and local.is_cuda \
        else dist.all_gather(list(recv.view(world, cap).unbind(0)), send, group=group)
    host = recv.view(world, cap).cpu().numpy()
    return [host[r, : lens[r]].view(SCORE_DTYPE).copy() for r in range(world)]


class ShardedLoopSearch:
    """One process per GPU: this rank's slice of a frame-sharded loop search.

    `scorer` is a Matcher-like object (the HIP library in production; tests inject a CPU stand-in) exposing
    append(frame_id, rows), query_scores(rows, frame_id) -> (scores, ids), __len__.  `group` is a torch.distributed
    process group (None = default group; pass world == 1 to run without torch.distributed).

    Online use mirrors LoopClosingSystem::processFrame (include/loop_closing.hpp:34): every rank calls
    process_frame(rows, frame_id) for EVERY frame, in the same order; the frame is scored against the frames this
    rank owns, the per-shard records are all-gathered, and every rank gets the same merged, ascending-id answer."""

    def __init__(self, scorer, rank: int = 0, world: int = 1, group=None, device=None, params=None):
        self.scorer, self.rank, self.world, self.group, self.device = scorer, int(rank), int(world), group, device
        self.params = params            # object with min_matches / sim_threshold (loop test); None -> scorer.params
        self.ids: List[int] = []        # ids of ALL frames seen, arrival order == ascending
        self.kp: List[int] = []         # keypoint counts of all frames (similarity denominator)

    # -- helpers -----------------------------------------------------------------------------------------
    def _gather(self, local: np.ndarray) -> List[np.ndarray]:
        if self.world == 1:
            return [local]
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local).view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        return all_gather_scores(t, len(local), self.group)

    def _loop_params(self):
        p = self.params if self.params is not None else self.scorer.params
        return int(p.min_matches), float(p.sim_threshold), int(p.min_gap)

    # -- online -------------------------------------------------------------------------------------------
    def process_frame(self, rows: np.ndarray, frame_id: int, n_keypoints: int = -1):
        """Returns (scores, stored_ids, candidates): merged records of this frame against every eligible stored
        frame of ALL shards, in ascending stored-frame order, and the loop candidates among them."""
        if self.ids and frame_id <= self.ids[-1]:
            raise ValueError("frame ids must increase")
        local, _ = self.scorer.query_scores(rows, frame_id)
        shards = self._gather(np.asarray(local))
        min_matches, thr, gap = self._loop_params()
        ids = np.asarray(self.ids, np.int64)
        e = int(np.searchsorted(ids, frame_id - max(gap, 1), side="right")) if len(ids) else 0
        merged = np.zeros(e, SCORE_DTYPE)
        for r, s in enumerate(shards):
            want = max(0, (e - r + self.world - 1) // self.world)
            if len(s) != want:
                raise ValueError(f"rank {r}: expected {want} records, got {len(s)}")
            merged[r::self.world][:want] = s
        nq = int(len(rows)) if n_keypoints < 0 else int(n_keypoints)
        cands = []
        for i in range(e):
            den = min(nq, self.kp[i])
            good = int(merged[i]["good_count"])
            if den > 0 and good >= min_matches and (good / den) > thr:       # README.md:123-126, IEEE double
                cands.append((int(frame_id), int(self.ids[i]), good, good / den))
        pos = len(self.ids)
        if pos % self.world == self.rank:
            self.scorer.append(frame_id, rows) if n_keypoints < 0 else self.scorer.append(frame_id, rows, n_keypoints)
        self.ids.append(int(frame_id))
        self.kp.append(nq)
        return merged, np.asarray(self.ids[:e], np.int32), cands
