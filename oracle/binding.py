"""ctypes binding of oracle/liblcm_oracle.so — the CPU restatement (TEST INFRASTRUCTURE, see lcm_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never from the product
package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcm_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [("ratio", C.c_int32), ("dist_floor", C.c_int32), ("min_matches", C.c_int32),
                ("min_gap", C.c_int32), ("sim_threshold", C.c_double), ("cross_check", C.c_int32),
                ("reserved", C.c_int32)]


SCORE_DTYPE = np.dtype([("good_count", "<u4"), ("min_dist", "<u2"), ("n_train", "<u2")])
DMATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"), ("distance", "<f4")])
CANDIDATE_DTYPE = np.dtype([("current_frame_id", "<i4"), ("matched_frame_id", "<i4"), ("num_matches", "<i4"),
                            ("_pad", "<i4"), ("similarity_score", "<f8")])

_vp = C.c_void_p
_lib = None


def build(force: bool = False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "lcm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        L.orc_hamming256.restype = C.c_int
        L.orc_hamming256.argtypes = [_vp, _vp]
        L.orc_bf_match.restype = C.c_int
        L.orc_bf_match.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, _vp]
        L.orc_bf_match_cross.restype = C.c_int
        L.orc_bf_match_cross.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]
        L.orc_filter_good.restype = C.c_int
        L.orc_filter_good.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)]
        L.orc_match_features.restype = C.c_int
        L.orc_match_features.argtypes = [_vp, C.c_int, _vp, C.c_int, C.POINTER(OrcParams), _vp, C.POINTER(C.c_int)]
        L.orc_pair_score.restype = None
        L.orc_pair_score.argtypes = [_vp, C.c_int, _vp, C.c_int, C.POINTER(OrcParams), _vp]
        L.orc_loop_test.restype = C.c_int
        L.orc_loop_test.argtypes = [C.POINTER(OrcParams), C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_detect_loops.restype = C.c_int
        L.orc_detect_loops.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(OrcParams), _vp, C.c_int]
        L.orc_all_vs_all.restype = C.c_size_t
        L.orc_all_vs_all.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(OrcParams), C.c_int, C.c_int, _vp, _vp]
        L.orc_fast_score_pairs.restype = C.c_double
        L.orc_fast_score_pairs.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(OrcParams), C.c_int,
                                           _vp, C.c_char_p]
        L.orc_fast_score_pairs_idx.restype = C.c_double
        L.orc_fast_score_pairs_idx.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(OrcParams), C.c_int,
                                               _vp, _vp, C.c_char_p]
        _lib = L
    return _lib


def default_params(**kw) -> OrcParams:
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _rows(a):
    a = np.ascontiguousarray(a, np.uint8)
    assert a.ndim == 2 and a.shape[1] == 32, a.shape
    return a


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_hamming256(_p(a), _p(b))


def bf_match(q, t):
    q, t = _rows(q), _rows(t)
    idx = np.full(q.shape[0], -1, np.int32)
    d = np.full(q.shape[0], -1, np.int32)
    n = lib().orc_bf_match(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(d))
    return idx[:n], d[:n]


def bf_match_cross(q, t, mode):
    """BFMatcher(NORM_HAMMING, crossCheck=True).match, mode 1 (mutual, recent OpenCV) / 2 (legacy):
    (train_idx int32[nq] with -1 = unmatched, dist int32[nq])."""
    q, t = _rows(q), _rows(t)
    idx = np.full(max(q.shape[0], 1), -1, np.int32)
    d = np.full(max(q.shape[0], 1), -1, np.int32)
    lib().orc_bf_match_cross(_p(q), q.shape[0], _p(t), t.shape[0], mode, idx.ctypes.data_as(_vp), d.ctypes.data_as(_vp))
    return idx[: q.shape[0]], d[: q.shape[0]]


def filter_good(dist, ratio=2, dist_floor=0):
    d = np.ascontiguousarray(dist, np.int32)
    keep = np.zeros(max(d.shape[0], 1), np.uint8)
    m = C.c_int(0)
    g = lib().orc_filter_good(_p(d), d.shape[0], ratio, dist_floor, keep.ctypes.data_as(_vp), C.byref(m))
    return g, keep[: d.shape[0]].astype(bool), m.value


def match_features(q, t, params=None):
    q, t = _rows(q), _rows(t)
    p = params or default_params()
    out = np.zeros(max(q.shape[0], 1), DMATCH_DTYPE)
    m = C.c_int(0)
    n = lib().orc_match_features(_p(q), q.shape[0], _p(t), t.shape[0], C.byref(p), out.ctypes.data_as(_vp), C.byref(m))
    return out[:n], m.value


def pair_score(q, t, params=None):
    q, t = _rows(q), _rows(t)
    p = params or default_params()
    out = np.zeros(1, SCORE_DTYPE)
    lib().orc_pair_score(_p(q), q.shape[0], _p(t), t.shape[0], C.byref(p), out.ctypes.data_as(_vp))
    return out[0]


def loop_test(good_count, n_q, n_t, params=None):
    p = params or default_params()
    sim = C.c_double(0)
    r = lib().orc_loop_test(C.byref(p), int(good_count), int(n_q), int(n_t), C.byref(sim))
    return bool(r), sim.value


def detect_loops(rows, counts, ids, cur, params=None):
    rows = np.ascontiguousarray(rows, np.uint8)
    counts = np.ascontiguousarray(counts, np.int32)
    ids = np.ascontiguousarray(ids, np.int32)
    p = params or default_params()
    n_frames, stride = rows.shape[0], rows.shape[1]
    out = np.zeros(max(n_frames, 1), CANDIDATE_DTYPE)
    n = lib().orc_detect_loops(_p(rows), _p(counts), _p(ids), n_frames, stride, cur, C.byref(p),
                               out.ctypes.data_as(_vp), out.shape[0])
    return out[:n]


def all_vs_all(rows, counts, ids, params=None, shard_rank=0, shard_world=1):
    rows = np.ascontiguousarray(rows, np.uint8)
    counts = np.ascontiguousarray(counts, np.int32)
    ids = np.ascontiguousarray(ids, np.int32)
    p = params or default_params()
    n_frames, stride = rows.shape[0], rows.shape[1]
    offs = np.zeros(n_frames + 1, np.uintp)
    n = lib().orc_all_vs_all(_p(rows), _p(counts), _p(ids), n_frames, stride, C.byref(p), shard_rank, shard_world,
                             None, offs.ctypes.data_as(_vp))
    scores = np.zeros(max(n, 1), SCORE_DTYPE)
    lib().orc_all_vs_all(_p(rows), _p(counts), _p(ids), n_frames, stride, C.byref(p), shard_rank, shard_world,
                         scores.ctypes.data_as(_vp), offs.ctypes.data_as(_vp))
    return scores[:n], offs


def fast_score_pairs(rows, counts, pair_q, pair_t, params=None, n_threads=1):
    """Tuned CPU baseline: returns (scores, seconds, isa)."""
    rows = np.ascontiguousarray(rows, np.uint8)
    counts = np.ascontiguousarray(counts, np.int32)
    pq = np.ascontiguousarray(pair_q, np.int32)
    pt = np.ascontiguousarray(pair_t, np.int32)
    p = params or default_params()
    scores = np.zeros(max(pq.shape[0], 1), SCORE_DTYPE)
    isa = C.create_string_buffer(64)
    secs = lib().orc_fast_score_pairs(_p(rows), _p(counts), rows.shape[1], _p(pq), _p(pt), pq.shape[0], C.byref(p),
                                      n_threads, scores.ctypes.data_as(_vp), isa)
    return scores[: pq.shape[0]], secs, isa.value.decode()


def fast_score_pairs_idx(rows, counts, pair_q, pair_t, params=None, n_threads=1):
    """Tuned path + the per-pair checksum of the good matches' train indices: (scores, idx_sums uint32)."""
    rows = np.ascontiguousarray(rows, np.uint8)
    counts = np.ascontiguousarray(counts, np.int32)
    pq = np.ascontiguousarray(pair_q, np.int32)
    pt = np.ascontiguousarray(pair_t, np.int32)
    p = params or default_params()
    scores = np.zeros(max(pq.shape[0], 1), SCORE_DTYPE)
    sums = np.zeros(max(pq.shape[0], 1), np.uint32)
    isa = C.create_string_buffer(64)
    lib().orc_fast_score_pairs_idx(_p(rows), _p(counts), rows.shape[1], _p(pq), _p(pt), pq.shape[0], C.byref(p),
                                   n_threads, scores.ctypes.data_as(_vp), sums.ctypes.data_as(_vp), isa)
    return scores[: pq.shape[0]], sums[: pq.shape[0]]


def index_sum(q, t, params=None) -> int:
    """The same checksum from the SCALAR oracle's matchFeatures: sum of trainIdx over the good matches mod 2^32."""
    m, _ = match_features(q, t, params)
    return int(m["train_idx"].astype(np.uint64).sum() % (1 << 32))
