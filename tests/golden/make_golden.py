#!/usr/bin/env python3
"""Regenerates tests/golden/lcm_golden_v1.npz and lcm_golden_v2.npz.

Provenance: the reference holds no golden vectors for this path and OpenCV is absent from the image, so these
vectors are produced by the scalar CPU oracle (oracle/lcm_oracle.c, Part 1) from seeded synthetic frames
(slam-loop-closing_amd/synth.py).  They pin the oracle against regressions and give the GPU tests a fixed target;
they are NOT outputs of cv::BFMatcher ("parity unpinned", see oracle/lcm_oracle.h).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_oracle, load_package  # noqa: E402

pkg = load_package()
orc = load_oracle()

GAP = 2
fs = pkg.synth.make_frames(10, 96, seed=pkg.synth.BASE_SEED, ragged=True, dup_frac=0.6)
fs.counts[4] = 0                      # an empty frame
fs.rows[4] = 0
fs.rows[7, :5] = fs.rows[3, :5]       # planted exact copies across frames (min_dist == 0)
fs.rows[7, 9] = fs.rows[7, 8]         # duplicate rows inside a frame (tie on train index)
p = orc.default_params(min_gap=GAP, min_matches=5, sim_threshold=0.05)

scores, offs = orc.all_vs_all(fs.rows, fs.counts, fs.ids, p)
pairs = [(7, 3), (9, 1), (8, 0), (5, 4), (4, 2), (6, 6)]
out = dict(rows=fs.rows, counts=fs.counts, ids=fs.ids, gap=np.int32(GAP), min_matches=np.int32(5),
           sim_threshold=np.float64(0.05), scores=scores, offsets=offs.astype(np.int64),
           pairs=np.array(pairs, np.int32))
cands = []
for c in range(fs.n_frames):
    cands.append(orc.detect_loops(fs.rows, fs.counts, fs.ids, c, p))
out["candidates"] = np.concatenate(cands)
for k, (a, b) in enumerate(pairs):
    idx, d = orc.bf_match(fs.frame(a), fs.frame(b))
    m, md = orc.match_features(fs.frame(a), fs.frame(b), p)
    out[f"pair{k}_idx"] = idx
    out[f"pair{k}_dist"] = d
    out[f"pair{k}_good"] = m
    out[f"pair{k}_min"] = np.int32(md)
np.savez_compressed(os.path.join(HERE, "lcm_golden_v1.npz"), **out)
print("wrote lcm_golden_v1.npz:", len(scores), "pair scores,", len(out["candidates"]), "loop candidates")


# ---- v2: the SELECTIVE synthetic variant (synth.make_frames_selective: 30 shared pool descriptors per frame), default
# README parameters (>= 50 good matches, similarity > 0.15): unrelated pairs keep ~30 matches and are NOT loops, revisits
# of moderately noisy places are — both branches of the loop test occur, with realistic counts on either side of 50.
GAP2 = 2
fs2 = pkg.synth.make_frames_selective(24, 224, seed=pkg.synth.BASE_SEED + 7, dup_frac=0.1)
p2 = orc.default_params(min_gap=GAP2)
scores2, offs2 = orc.all_vs_all(fs2.rows, fs2.counts, fs2.ids, p2)
cands2 = np.concatenate([orc.detect_loops(fs2.rows, fs2.counts, fs2.ids, c, p2) for c in range(fs2.n_frames)])
isums2 = []
for c in range(fs2.n_frames):
    for t in range(fs2.n_frames):
        if fs2.ids[c] - fs2.ids[t] >= GAP2:
            isums2.append(orc.index_sum(fs2.frame(c), fs2.frame(t), p2))
out2 = dict(rows=fs2.rows, counts=fs2.counts, ids=fs2.ids, gap=np.int32(GAP2), min_matches=np.int32(p2.min_matches),
            sim_threshold=np.float64(p2.sim_threshold), scores=scores2, offsets=offs2.astype(np.int64), candidates=cands2,
            index_sums=np.array(isums2, np.uint32))
np.savez_compressed(os.path.join(HERE, "lcm_golden_v2.npz"), **out2)
good2 = scores2["good_count"]
print("wrote lcm_golden_v2.npz:", len(scores2), "pair scores,", len(cands2), "loop candidates; good counts",
      int(good2.min()), "..", int(good2.max()), "; pairs with 40 <= good < 50:", int(((good2 >= 40) & (good2 < 50)).sum()))
