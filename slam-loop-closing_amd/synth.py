"""Seeded synthetic ORB-like descriptor sets.

The reference's input videos are absent (.MISSING_LARGE_BLOBS) and ORB extraction is out of scope, so every
benchmark and parity input is synthetic.  Uniform-random descriptors alone are useless for the README filter
(best-of-2000 distances are 88..107, so 2 x min >= all and every match passes), therefore frames are
place-structured:

  * P = max(1, n_frames // 4) "places", each a base set of `max_desc` uniform-random 256-bit descriptors;
  * frame f looks at place f mod P, so revisits are P frames apart (>> min_loop_gap for the BASELINE configs);
  * a fraction `inlier_frac` of a frame's rows are base rows of its place with every bit flipped with
    probability flip_p(place) in [0.02, 0.12] (so min_dist spans "tiny" to "moderate" and the ratio filter is
    selective in different ways); the rest are fresh uniform rows (outliers);
  * rows are shuffled per frame; `dup_frac` of the frames carry exact copies of a place's base rows and of their
    own rows (ties and min_dist == 0);
  * with ragged=True the row count is uniform in [0.75 * max_desc, max_desc].

  * with pool_k > 0 ("selective" variant) every frame also carries noisy copies of the SAME pool_k global descriptors
    (each bit flipped with probability pool_flip) in place of as many outlier rows.  Without them two UNRELATED frames
    have no close rows at all: their best distances are 88..107, 2 x min keeps every match, the similarity is 1.0 and
    every pair "is a loop" (99.96 % of cfg4's pairs) — the README filter is vacuous on exactly the pairs that dominate.
    With them every pair of frames shares pool_k true correspondences at distance ~46 +- 6 (pool_flip 0.10 on both
    sides): min_dist ~ 34, the filter keeps those ~pool_k matches and rejects the 88+ ones, good_count ~ 30 < 50 and the
    pair is NOT a loop; a revisit of a moderately noisy place (flip_p >= ~0.07: inlier distances concentrated well
    above zero, so 2 x min covers most of them) keeps several hundred matches and IS one; a revisit of a very clean
    place is not (its minimum is ~1, so "2 x min" keeps a handful of matches: the README rule's own blind spot).

Everything is a pure function of (seed, n_frames, max_desc, ...): any rank can regenerate any frame.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DESC_BYTES = 32
BASE_SEED = 20260116   # + config index, as SURVEY.md §8d prescribes


@dataclass
class FrameSet:
    rows: np.ndarray      # (n_frames, stride_rows, 32) uint8; rows beyond counts[f] are zero
    counts: np.ndarray    # (n_frames,) int32
    ids: np.ndarray       # (n_frames,) int32, strictly increasing
    seed: int

    @property
    def n_frames(self) -> int:
        return int(self.rows.shape[0])

    @property
    def stride_rows(self) -> int:
        return int(self.rows.shape[1])

    def frame(self, f: int) -> np.ndarray:
        return self.rows[f, : int(self.counts[f])]


def _flip(rng: np.random.Generator, rows: np.ndarray, p: float) -> np.ndarray:
    """Flip every bit of `rows` independently with probability p."""
    n = rows.shape[0]
    # Bernoulli(p) bits packed into bytes: compare uniform uint16 draws against a threshold
    thr = int(round(p * 65536))
    bits = (rng.integers(0, 65536, size=(n, DESC_BYTES * 8), dtype=np.uint16) < thr)
    return rows ^ np.packbits(bits, axis=1)


def make_frames(n_frames: int, max_desc: int, seed: int = BASE_SEED, *, inlier_frac: float = 0.4,
                dup_frac: float = 0.01, ragged: bool = False, id_step: int = 1, pool_k: int = 0,
                pool_flip: float = 0.10) -> FrameSet:
    n_places = max(1, n_frames // 4)
    pool = np.random.default_rng([seed, 4]).integers(0, 256, size=(pool_k, DESC_BYTES), dtype=np.uint8) if pool_k > 0 else None
    rows = np.zeros((n_frames, max_desc, DESC_BYTES), np.uint8)
    counts = np.zeros(n_frames, np.int32)
    place_rng = [np.random.default_rng([seed, 1, p]) for p in range(n_places)]
    bases = {}

    def base(p):
        if p not in bases:
            bases[p] = place_rng[p].integers(0, 256, size=(max_desc, DESC_BYTES), dtype=np.uint8)
        return bases[p]

    for f in range(n_frames):
        rng = np.random.default_rng([seed, 2, f])
        n = max_desc
        if ragged:
            n = int(rng.integers(int(0.75 * max_desc), max_desc + 1))
        p = f % n_places
        flip_p = 0.02 + 0.10 * ((p * 7919) % 11) / 10.0
        n_in = int(round(inlier_frac * n))
        pick = rng.permutation(max_desc)[:n_in]
        inl = _flip(rng, base(p)[pick], flip_p)
        out = rng.integers(0, 256, size=(n - n_in, DESC_BYTES), dtype=np.uint8)
        if pool is not None and n - n_in > 0:
            k = min(pool_k, n - n_in)
            out[:k] = _flip(rng, pool[:k], pool_flip)
        fr = np.concatenate([inl, out], axis=0)
        if rng.random() < dup_frac and n >= 8:
            # planted exact duplicates: copies of base rows (distance 0 against the place) and of own rows (ties)
            k = max(1, n // 50)
            fr[:k] = base(p)[pick[:k]] if n_in >= k else fr[:k]
            fr[n - k:] = fr[:k]
        fr = fr[rng.permutation(n)]
        rows[f, :n] = fr
        counts[f] = n
    ids = (np.arange(n_frames, dtype=np.int32) * id_step).astype(np.int32)
    return FrameSet(rows=rows, counts=counts, ids=ids, seed=seed)


def make_frames_selective(n_frames: int, max_desc: int, seed: int = BASE_SEED, **kw) -> FrameSet:
    """The variant on which the README filter is selective for unrelated pairs too (module docstring): 30 shared pool
    descriptors per frame.  Loop candidates are then (a part of) the revisit pairs only."""
    kw.setdefault("pool_k", 30)
    return make_frames(n_frames, max_desc, seed, **kw)


def uniform_frames(n_frames: int, max_desc: int, seed: int = BASE_SEED) -> FrameSet:
    """Plain uniform-random rows: fine for raw distances/s, useless for the filter (see module docstring)."""
    rng = np.random.default_rng([seed, 3])
    rows = rng.integers(0, 256, size=(n_frames, max_desc, DESC_BYTES), dtype=np.uint8)
    return FrameSet(rows=rows, counts=np.full(n_frames, max_desc, np.int32),
                    ids=np.arange(n_frames, dtype=np.int32), seed=seed)


def n_pairs_all_vs_all(n_frames: int, gap: int) -> int:
    """pairs = sum_c max(0, c - gap + 1) for ids == indices."""
    m = n_frames - gap
    return m * (m + 1) // 2 if m > 0 else 0


def frames_for_pairs(target_pairs: int, gap: int) -> int:
    """Smallest frame count whose all-vs-all has at least target_pairs pairs."""
    n = gap
    while n_pairs_all_vs_all(n, gap) < target_pairs:
        n += 1
    return n
