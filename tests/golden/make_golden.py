#!/usr/bin/env python3
"""Regenerates tests/golden/lcm_golden_v1.npz.

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
