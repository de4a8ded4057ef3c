"""A third, pure-Python restatement (big-int XOR + int.bit_count, explicit strict-'<' scan) of the matcher and of
detectLoops, written without looking at the C code's structure: small cases only.  Three independently written
implementations (C scalar, numpy, this one) agreeing is the strongest pin available without OpenCV."""
import numpy as np


def py_match(q, t):
    Q = [int.from_bytes(r.tobytes(), "little") for r in q]
    T = [int.from_bytes(r.tobytes(), "little") for r in t]
    out = []
    if not Q or not T:
        return out
    for i, a in enumerate(Q):
        best, bj = 1 << 30, -1
        for j, b in enumerate(T):
            d = (a ^ b).bit_count()
            if d < best:
                best, bj = d, j
        out.append((i, bj, best))
    return out


def py_good(matches, ratio=2, floor=0):
    if not matches:
        return []
    m = min(d for _, _, d in matches)
    thr = max(ratio * m, floor)
    return [x for x in matches if x[2] <= thr]


def py_detect_loops(frames, ids, cur, gap, thr, min_matches):
    out = []
    for i in range(len(frames)):
        if ids[cur] - ids[i] < gap or i == cur:
            continue
        good = len(py_good(py_match(frames[cur], frames[i])))
        den = min(len(frames[cur]), len(frames[i]))
        if den > 0 and good / den > thr and good >= min_matches:
            out.append((int(ids[cur]), int(ids[i]), good, good / den))
    return out


def test_three_implementations_agree(oracle, pkg):
    fs = pkg.synth.make_frames(9, 40, seed=123, ragged=True, dup_frac=0.6)
    fs.counts[2] = 0
    frames = [fs.frame(f) for f in range(fs.n_frames)]
    p = oracle.default_params(min_gap=2, min_matches=3, sim_threshold=0.05)
    for a in range(fs.n_frames):
        for b in range(fs.n_frames):
            pm = py_match(frames[a], frames[b])
            idx, d = oracle.bf_match(frames[a], frames[b])
            assert [(i, int(idx[i]), int(d[i])) for i in range(len(idx))] == pm
            g, md = oracle.match_features(frames[a], frames[b], p)
            assert [(int(r["query_idx"]), int(r["train_idx"]), int(r["distance"])) for r in g] == py_good(pm)
    for cur in range(fs.n_frames):
        want = py_detect_loops(frames, fs.ids, cur, 2, 0.05, 3)
        got = oracle.detect_loops(fs.rows, fs.counts, fs.ids, cur, p)
        assert [(int(r["current_frame_id"]), int(r["matched_frame_id"]), int(r["num_matches"]), float(r["similarity_score"]))
                for r in got] == want
    assert any(py_detect_loops(frames, fs.ids, c, 2, 0.05, 3) for c in range(fs.n_frames))
