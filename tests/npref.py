"""Independent numpy restatement of BFMatcher(NORM_HAMMING).match + README filter, used ONLY to cross-check the C
oracle (two implementations written separately must agree).  np.unpackbits / argmin semantics: argmin returns the
FIRST minimum, which is exactly OpenCV's strict-'<' ascending scan."""
import numpy as np


def dist_matrix(q, t):
    q = np.asarray(q, np.uint8)
    t = np.asarray(t, np.uint8)
    x = q[:, None, :] ^ t[None, :, :]
    return np.unpackbits(x, axis=2).sum(axis=2, dtype=np.int32)


def bf_match(q, t):
    if len(q) == 0 or len(t) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    d = dist_matrix(q, t)
    idx = d.argmin(axis=1).astype(np.int32)
    return idx, d[np.arange(len(q)), idx].astype(np.int32)


def pair_score(q, t, ratio=2, floor=0):
    idx, d = bf_match(q, t)
    if len(d) == 0:
        return 0, 0xFFFF, len(t)
    m = int(d.min())
    thr = max(ratio * m, floor)
    return int((d <= thr).sum()), m, len(t)
