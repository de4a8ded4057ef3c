"""ctypes binding of include/lcm.h — the C ABI of the MI355X loop-closure matcher.

This module is plumbing for tests, bench.py and __graft_entry__: the product is the shared library
`lib/liblcm_hip.so` (hand-written gfx950 kernels behind a C ABI).  There is no Python or CPU compute path here;
if the library is missing or no HIP device is present, everything raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCM_LIB_PATH: test plumbing only (e.g. the host-sanitizer build of the same sources, `make -C csrc asan`)
LIB_PATH = os.environ.get("LCM_LIB_PATH") or os.path.join(_HERE, "lib", "liblcm_hip.so")
DESC_BYTES = 32
KEY_SHIFT = 22
TUNE_ITEM_SLOTS, TUNE_ONLINE_SPLIT, TUNE_PACKED, TUNE_ONLINE_STREAMS, TUNE_PACKED_SCRATCH_MB = 0, 1, 2, 3, 4      # lcm_tuning
TUNE_PAIR_UPLOAD_KERNEL, TUNE_PAIR_HOST_FOLD = 5, 6


OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_ORDER, ERR_NOT_FOUND, ERR_OOM = 0, -1, -2, -3, -4, -5, -6, -7   # lcm_status


class LcmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"lcm error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("ratio", C.c_int32), ("dist_floor", C.c_int32), ("min_matches", C.c_int32),
                ("min_gap", C.c_int32), ("sim_threshold", C.c_double), ("cross_check", C.c_int32),
                ("reserved", C.c_int32)]


class Score(C.Structure):
    _fields_ = [("good_count", C.c_uint32), ("min_dist", C.c_uint16), ("n_train", C.c_uint16)]


class DMatch(C.Structure):
    _fields_ = [("query_idx", C.c_int32), ("train_idx", C.c_int32), ("img_idx", C.c_int32), ("distance", C.c_float)]


class LoopCandidate(C.Structure):
    _fields_ = [("current_frame_id", C.c_int32), ("matched_frame_id", C.c_int32), ("num_matches", C.c_int32),
                ("similarity_score", C.c_double)]


class LaunchInfo(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("pairs", C.c_uint64), ("distances", C.c_uint64),
                ("algo_bytes", C.c_uint64), ("launches", C.c_uint32), ("workgroups", C.c_uint32),
                ("aux_kernel_ms", C.c_double), ("route", C.c_uint32), ("launches_in_flight", C.c_uint32),
                ("score_ms_sum", C.c_double), ("score_launches", C.c_uint32), ("reserved_", C.c_uint32)]


ROUTE_PLAIN, ROUTE_PACKED, ROUTE_SPLIT, ROUTE_MATRIX, ROUTE_CROSS = range(5)      # lcm_route


class OnlineStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("launches", C.c_uint64), ("queries", C.c_uint64), ("pairs", C.c_uint64),
                ("distances", C.c_uint64), ("algo_bytes", C.c_uint64)]


class GroupInfo(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("rccl_ranks", C.c_int32), ("pairs", C.c_uint64), ("distances", C.c_uint64),
                ("algo_bytes", C.c_uint64), ("kernel_ms_max", C.c_double), ("gather_merge_ms", C.c_double),
                ("download_ms", C.c_double), ("gathered_query_bytes", C.c_uint64), ("gathered_score_bytes", C.c_uint64),
                ("allgather_ms", C.c_double), ("kernel_ms", C.c_double * 8), ("shard_pairs", C.c_uint64 * 8),
                ("arena_gather_skipped", C.c_int32), ("loopback", C.c_int32)]


SCORE_DTYPE = np.dtype([("good_count", "<u4"), ("min_dist", "<u2"), ("n_train", "<u2")])
DMATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"), ("distance", "<f4")])
CANDIDATE_DTYPE = np.dtype([("current_frame_id", "<i4"), ("matched_frame_id", "<i4"), ("num_matches", "<i4"),
                            ("_pad", "<i4"), ("similarity_score", "<f8")])
assert SCORE_DTYPE.itemsize == C.sizeof(Score) == 8
assert DMATCH_DTYPE.itemsize == C.sizeof(DMatch) == 16
assert CANDIDATE_DTYPE.itemsize == C.sizeof(LoopCandidate) == 24

_lib = None

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/lcm.h one to one (tests/test_abi.py checks the symbol list)
_SIGNATURES = {
    "lcm_params_default": (None, [C.POINTER(Params)]),
    "lcm_last_error": (C.c_char_p, []),
    "lcm_backend_name": (C.c_char_p, []),
    "lcm_device_count": (C.c_int, []),
    "lcm_create": (C.c_int, [C.POINTER(Params), C.c_int, _vp, C.POINTER(_vp)]),
    "lcm_destroy": (None, [_vp]),
    "lcm_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "lcm_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "lcm_sync": (C.c_int, [_vp]),
    "lcm_db_reserve": (C.c_int, [_vp, C.c_int, C.c_int]),
    "lcm_db_append": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    "lcm_db_append_device": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    "lcm_db_size": (C.c_int, [_vp]),
    "lcm_db_clear": (C.c_int, [_vp]),
    "lcm_db_truncate": (C.c_int, [_vp, C.c_int]),
    "lcm_db_frame_info": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p]),
    "lcm_db_read": (C.c_int, [_vp, C.c_int, _vp, C.c_int]),
    "lcm_db_save": (C.c_int, [_vp, C.c_char_p]),
    "lcm_db_load": (C.c_int, [_vp, C.c_char_p]),
    "lcm_match_pair": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _i32p]),
    "lcm_match_features": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _i32p, _i32p]),
    "lcm_match_stored": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _i32p, _i32p]),
    "lcm_match_stored_batch": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, _vp, _vp]),
    "lcm_match_query_batch": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_size_t, _vp, _vp]),
    "lcm_query_scores": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _i32p]),
    "lcm_query_submit": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i32p]),
    "lcm_query_collect": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i32p]),
    "lcm_query_submit_batch": (C.c_int, [_vp, _vp, _i32p, _i32p, C.c_int, _i32p]),
    "lcm_query_collect_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "lcm_online_stats_read": (C.c_int, [_vp, C.POINTER(OnlineStats), C.c_int]),
    "lcm_detect_loops": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _i32p]),
    "lcm_loop_test": (C.c_int, [C.POINTER(Params), C.POINTER(Score), C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "lcm_all_vs_all": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t,
                                  C.POINTER(C.c_size_t), _vp]),
    "lcm_all_vs_all_argmin": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp,
                                         C.POINTER(C.c_size_t), _vp]),
    "lcm_all_vs_all_loops": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lcm_last_launch_info": (C.c_int, [_vp, C.POINTER(LaunchInfo)]),
    "lcm_last_bulk_scores": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "lcm_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "lcm_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "lcm_group_create": (C.c_int, [C.POINTER(Params), C.c_int, _i32p, C.POINTER(_vp)]),
    "lcm_group_create_loopback": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.POINTER(_vp)]),
    "lcm_group_create_peer": (C.c_int, [C.POINTER(Params), C.c_int, _i32p, C.POINTER(_vp)]),
    "lcm_group_transport": (C.c_char_p, [_vp]),
    "lcm_group_destroy": (None, [_vp]),
    "lcm_group_size": (C.c_int, [_vp]),
    "lcm_group_db_size": (C.c_int, [_vp]),
    "lcm_group_handle": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "lcm_group_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "lcm_group_reserve": (C.c_int, [_vp, C.c_int, C.c_int]),
    "lcm_group_append": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    "lcm_group_clear": (C.c_int, [_vp]),
    "lcm_group_truncate": (C.c_int, [_vp, C.c_int]),
    "lcm_group_save": (C.c_int, [_vp, C.c_char_p]),
    "lcm_group_load": (C.c_int, [_vp, C.c_char_p]),
    "lcm_group_sync": (C.c_int, [_vp]),
    "lcm_group_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "lcm_group_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "lcm_group_all_vs_all": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "lcm_group_all_vs_all_argmin": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "lcm_group_all_vs_all_loops": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lcm_group_query_submit_batch": (C.c_int, [_vp, _vp, _i32p, _i32p, C.c_int, _i32p]),
    "lcm_group_query_collect_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "lcm_group_online_stats_read": (C.c_int, [_vp, C.POINTER(OnlineStats), C.c_int]),
    "lcm_group_last_info": (C.c_int, [_vp, C.POINTER(GroupInfo)]),
    "lcm_group_query_scores": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _i32p]),
    "lcm_group_query_scores_batch": (C.c_int, [_vp, _vp, _i32p, _i32p, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "lcm_group_detect_loops": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _i32p]),
    "lcm_merge_shard_scores": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_size_t,
                                          C.POINTER(C.c_size_t), _vp]),
    "lcm_merge_shard_scores_device": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_size_t,
                                                 C.POINTER(C.c_size_t)]),
    "lcm_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "lcm_dev_free": (C.c_int, [_vp, _vp]),
    "lcm_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "lcm_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
}


def load_library(path: Optional[str] = None):
    """dlopen the C-ABI library.  Raises OSError if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise LcmError(rc, load_library().lcm_last_error().decode("utf-8", "replace"))


def default_params() -> Params:
    p = Params()
    load_library().lcm_params_default(C.byref(p))
    return p


def _rows(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2 or a.shape[1] != DESC_BYTES:
        raise ValueError(f"descriptor matrix must be (n, {DESC_BYTES}) uint8, got {a.shape}")
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data_as(_vp)


class Matcher:
    """One matcher handle bound to one HIP device (one process per GPU).

    `stream` is a hipStream_t handle as an int (e.g. `torch.cuda.Stream(...).cuda_stream`); None or 0 (the legacy default
    stream's handle) makes the library create its own non-blocking stream.  When results are consumed by another
    framework on the device (torch.distributed collectives on the buffers lcm_all_vs_all filled), pass that framework's
    CURRENT stream — an explicit one — so its work is ordered after the kernels; see bench.py."""

    def __init__(self, params: Optional[Params] = None, device: int = 0, stream: Optional[int] = None):
        self._lib = load_library()
        self._h = _vp()
        p = params if params is not None else default_params()
        _check(self._lib.lcm_create(C.byref(p), device, _vp(stream) if stream else None, C.byref(self._h)))

    # -- lifetime ----------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.lcm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- parameters --------------------------------------------------------------------------------
    @property
    def params(self) -> Params:
        p = Params()
        _check(self._lib.lcm_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.params
        for k, v in kw.items():
            setattr(p, k, v)
        _check(self._lib.lcm_set_params(self._h, C.byref(p)))

    def set_kernel_variant(self, v: int):
        _check(self._lib.lcm_set_kernel_variant(self._h, v))

    def set_tuning(self, knob: int, value: int):
        _check(self._lib.lcm_set_tuning(self._h, knob, value))

    def sync(self):
        _check(self._lib.lcm_sync(self._h))

    # -- database ----------------------------------------------------------------------------------
    def reserve(self, n_frames: int, max_desc: int):
        _check(self._lib.lcm_db_reserve(self._h, n_frames, max_desc))

    def append(self, frame_id: int, desc, n_keypoints: int = -1):
        d = _rows(desc)
        _check(self._lib.lcm_db_append(self._h, frame_id, _ptr(d), d.shape[0], n_keypoints))

    def append_device(self, frame_id: int, d_ptr: int, n: int, n_keypoints: int = -1):
        _check(self._lib.lcm_db_append_device(self._h, frame_id, _vp(d_ptr), n, n_keypoints))

    def __len__(self):
        return self._lib.lcm_db_size(self._h)

    def clear(self):
        _check(self._lib.lcm_db_clear(self._h))

    def truncate(self, n_frames: int):
        _check(self._lib.lcm_db_truncate(self._h, n_frames))

    def frame_info(self, slot: int) -> Tuple[int, int, int]:
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self._lib.lcm_db_frame_info(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def read_frame(self, slot: int) -> np.ndarray:
        _, n, _ = self.frame_info(slot)
        out = np.empty((n, DESC_BYTES), np.uint8)
        _check(self._lib.lcm_db_read(self._h, slot, _ptr(out) if n else None, n))
        return out

    def save(self, path: str):
        _check(self._lib.lcm_db_save(self._h, path.encode()))

    def load(self, path: str):
        _check(self._lib.lcm_db_load(self._h, path.encode()))

    # -- pair mode ---------------------------------------------------------------------------------
    def match_pair(self, query, train) -> Tuple[np.ndarray, np.ndarray]:
        """BFMatcher(NORM_HAMMING).match: (train_idx int32[n], dist uint16[n]); n = 0 if either side is empty."""
        q, t = _rows(query), _rows(train)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.uint16)
        n = C.c_int32(0)
        _check(self._lib.lcm_match_pair(self._h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(idx), _ptr(dist),
                                        C.byref(n)))
        return idx[: n.value], dist[: n.value]

    def match_features(self, query, train) -> Tuple[np.ndarray, int]:
        """matchFeatures: structured array of DMatch records that survive the ratio*min filter, and min_dist."""
        q, t = _rows(query), _rows(train)
        out = np.zeros(max(q.shape[0], 1), DMATCH_DTYPE)
        n, m = C.c_int32(0), C.c_int32(0)
        _check(self._lib.lcm_match_features(self._h, _ptr(q), q.shape[0], _ptr(t), t.shape[0],
                                            out.ctypes.data_as(_vp), C.byref(n), C.byref(m)))
        return out[: n.value], m.value

    def match_stored(self, query_frame_id: int, train_frame_id: int, cap: int = 65536) -> Tuple[np.ndarray, int]:
        """matchFeatures between two stored frames (device-resident rows)."""
        out = np.zeros(cap, DMATCH_DTYPE)
        n, m = C.c_int32(0), C.c_int32(0)
        _check(self._lib.lcm_match_stored(self._h, query_frame_id, train_frame_id, out.ctypes.data_as(_vp), cap,
                                          C.byref(n), C.byref(m)))
        return out[: n.value], m.value

    def match_stored_batch(self, pairs: Sequence[Tuple[int, int]], cap: Optional[int] = None):
        """matchFeatures for many stored (query id, train id) pairs in one launch: (list of DMatch arrays, min_dists)."""
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = pr.shape[0]
        cap = 2048 * max(n, 1) if cap is None else cap
        out = np.zeros(max(cap, 1), DMATCH_DTYPE)
        offs = np.zeros(n + 1, np.uintp)
        md = np.zeros(max(n, 1), np.int32)
        _check(self._lib.lcm_match_stored_batch(self._h, _ptr(pr), n, out.ctypes.data_as(_vp), cap,
                                                offs.ctypes.data_as(_vp), md.ctypes.data_as(_vp)))
        return [out[int(offs[i]): int(offs[i + 1])] for i in range(n)], md[:n]

    def match_query_batch(self, query, train_ids: Sequence[int], cap: Optional[int] = None):
        """One host query frame against many stored frames in one launch: (list of DMatch arrays, min_dists)."""
        q = _rows(query)
        ids = np.ascontiguousarray(train_ids, np.int32)
        n = ids.shape[0]
        cap = max(q.shape[0], 1) * max(n, 1) if cap is None else cap
        out = np.zeros(max(cap, 1), DMATCH_DTYPE)
        offs = np.zeros(n + 1, np.uintp)
        md = np.zeros(max(n, 1), np.int32)
        _check(self._lib.lcm_match_query_batch(self._h, _ptr(q), q.shape[0], _ptr(ids), n, out.ctypes.data_as(_vp), cap,
                                               offs.ctypes.data_as(_vp), md.ctypes.data_as(_vp)))
        return [out[int(offs[i]): int(offs[i + 1])] for i in range(n)], md[:n]

    # -- loop search -------------------------------------------------------------------------------
    def query_scores(self, query, query_frame_id: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _rows(query)
        cap = max(len(self), 1)
        scores = np.zeros(cap, SCORE_DTYPE)
        ids = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        _check(self._lib.lcm_query_scores(self._h, _ptr(q), q.shape[0], query_frame_id, scores.ctypes.data_as(_vp),
                                          ids.ctypes.data_as(_vp), C.byref(n)))
        return scores[: n.value], ids[: n.value]

    def query_submit(self, query, query_frame_id: int) -> int:
        """Asynchronous query: returns a ticket at once (the rows are copied before returning)."""
        q = _rows(query)
        t = C.c_int32(-1)
        _check(self._lib.lcm_query_submit(self._h, _ptr(q), q.shape[0], query_frame_id, C.byref(t)))
        return t.value

    def query_collect(self, ticket: int, cap: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        cap = max(len(self), 1) if cap is None else cap
        scores = np.zeros(cap, SCORE_DTYPE)
        ids = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        _check(self._lib.lcm_query_collect(self._h, ticket, scores.ctypes.data_as(_vp), ids.ctypes.data_as(_vp), cap,
                                           C.byref(n)))
        return scores[: n.value], ids[: n.value]

    def query_submit_batch(self, queries: Sequence[np.ndarray], query_frame_ids: Sequence[int]) -> int:
        """Up to 16 query frames scored by ONE launch against the database as it stands now; returns one ticket."""
        qs = [_rows(q) for q in queries]
        B = len(qs)
        ptrs = (_vp * B)(*[q.ctypes.data if q.shape[0] else None for q in qs])
        nq = np.array([q.shape[0] for q in qs], np.int32)
        ids = np.ascontiguousarray(query_frame_ids, np.int32)
        assert ids.shape[0] == B
        t = C.c_int32(-1)
        _check(self._lib.lcm_query_submit_batch(self._h, ptrs, nq.ctypes.data_as(_i32p), ids.ctypes.data_as(_i32p), B,
                                                C.byref(t)))
        self._batch_sizes = getattr(self, "_batch_sizes", {})
        self._batch_sizes[t.value] = B
        return t.value

    def query_collect_batch(self, ticket: int, cap: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """(records of all the batch's queries back to back, offsets[B + 1])."""
        B = self._batch_sizes[ticket]
        cap = max(len(self), 1) * B if cap is None else cap
        scores = np.zeros(max(cap, 1), SCORE_DTYPE)
        offs = np.zeros(B + 1, np.uintp)
        n = C.c_size_t(0)
        _check(self._lib.lcm_query_collect_batch(self._h, ticket, scores.ctypes.data_as(_vp), cap, C.byref(n),
                                                 offs.ctypes.data_as(_vp)))
        return scores[: n.value], offs

    def online_stats(self, reset: bool = False) -> OnlineStats:
        st = OnlineStats()
        _check(self._lib.lcm_online_stats_read(self._h, C.byref(st), 1 if reset else 0))
        return st

    def detect_loops(self, current_frame_id: int, query=None, n_keypoints: int = -1) -> np.ndarray:
        cap = max(len(self), 1)
        out = np.zeros(cap, CANDIDATE_DTYPE)
        n = C.c_int32(0)
        if query is None:
            qp, nq = None, 0
        else:
            q = _rows(query)
            nq = q.shape[0]
            # an explicit empty frame still needs a non-NULL pointer to be told apart from "use the stored frame"
            qp = q.ctypes.data_as(_vp) if nq else np.zeros((1, DESC_BYTES), np.uint8).ctypes.data_as(_vp)
        _check(self._lib.lcm_detect_loops(self._h, current_frame_id, qp, nq, n_keypoints, out.ctypes.data_as(_vp), cap,
                                          C.byref(n)))
        return out[: n.value]

    def loop_test(self, score, n_query_kp: int, n_train_kp: int) -> Tuple[bool, float]:
        s = Score(int(score["good_count"]), int(score["min_dist"]), int(score["n_train"]))
        p = self.params
        sim = C.c_double(0)
        r = self._lib.lcm_loop_test(C.byref(p), C.byref(s), n_query_kp, n_train_kp, C.byref(sim))
        return bool(r), sim.value

    # -- bulk --------------------------------------------------------------------------------------
    def all_vs_all_plan(self, d_query_rows: int = 0, d_query_counts: int = 0, q_ids: Optional[Sequence[int]] = None,
                        q_stride_rows: int = 0) -> Tuple[int, np.ndarray]:
        """Sizing call: returns (n_pairs, offsets[n_q_frames + 1])."""
        ids = None if q_ids is None else np.ascontiguousarray(q_ids, np.int32)
        nq = len(self) if ids is None else ids.shape[0]
        offs = np.zeros(nq + 1, np.uintp)
        n = C.c_size_t(0)
        _check(self._lib.lcm_all_vs_all(self._h, _vp(d_query_rows) if d_query_rows else None,
                                        _vp(d_query_counts) if d_query_counts else None, _ptr(ids), nq, q_stride_rows,
                                        None, 0, C.byref(n), offs.ctypes.data_as(_vp)))
        return n.value, offs

    def all_vs_all(self, d_scores: int, scores_cap: int, d_query_rows: int = 0, d_query_counts: int = 0,
                   q_ids: Optional[Sequence[int]] = None, q_stride_rows: int = 0) -> int:
        """Enqueue the bulk scoring on the handle's stream; scores land in device memory at d_scores."""
        ids = None if q_ids is None else np.ascontiguousarray(q_ids, np.int32)
        nq = len(self) if ids is None else ids.shape[0]
        n = C.c_size_t(0)
        _check(self._lib.lcm_all_vs_all(self._h, _vp(d_query_rows) if d_query_rows else None,
                                        _vp(d_query_counts) if d_query_counts else None, _ptr(ids), nq, q_stride_rows,
                                        _vp(d_scores), scores_cap, C.byref(n), None))
        return n.value

    def all_vs_all_argmin(self, d_scores: int, scores_cap: int, d_index_sums: int, d_query_rows: int = 0,
                          d_query_counts: int = 0, q_ids: Optional[Sequence[int]] = None, q_stride_rows: int = 0) -> int:
        """all_vs_all through the argmin kernel; d_index_sums receives one uint32 per pair (sum of the good matches'
        train indices mod 2^32)."""
        ids = None if q_ids is None else np.ascontiguousarray(q_ids, np.int32)
        nq = len(self) if ids is None else ids.shape[0]
        n = C.c_size_t(0)
        _check(self._lib.lcm_all_vs_all_argmin(self._h, _vp(d_query_rows) if d_query_rows else None,
                                               _vp(d_query_counts) if d_query_counts else None, _ptr(ids), nq,
                                               q_stride_rows, _vp(d_scores), scores_cap, _vp(d_index_sums),
                                               C.byref(n), None))
        return n.value

    def all_vs_all_loops(self, cap: int = 1 << 20, d_query_rows: int = 0, d_query_counts: int = 0,
                         q_ids: Optional[Sequence[int]] = None, q_keypoints: Optional[Sequence[int]] = None,
                         q_stride_rows: int = 0, out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
        """Bulk loop search with the loop test fused on the device: (candidates sorted by (current, matched), n_pairs)."""
        ids = None if q_ids is None else np.ascontiguousarray(q_ids, np.int32)
        kps = None if q_keypoints is None else np.ascontiguousarray(q_keypoints, np.int32)
        nq = len(self) if ids is None else ids.shape[0]
        if out is None:
            out = np.zeros(max(cap, 1), CANDIDATE_DTYPE)
        else:
            assert out.dtype == CANDIDATE_DTYPE and out.flags["C_CONTIGUOUS"]
            cap = len(out)
        n, npairs = C.c_size_t(0), C.c_size_t(0)
        _check(self._lib.lcm_all_vs_all_loops(self._h, _vp(d_query_rows) if d_query_rows else None,
                                              _vp(d_query_counts) if d_query_counts else None, _ptr(ids), _ptr(kps), nq,
                                              q_stride_rows, out.ctypes.data_as(_vp), cap, C.byref(n), C.byref(npairs)))
        return out[: n.value], npairs.value

    def last_bulk_scores(self) -> np.ndarray:
        """Download the score records the last all_vs_all_loops call left on the device."""
        p, n = _vp(), C.c_size_t(0)
        _check(self._lib.lcm_last_bulk_scores(self._h, C.byref(p), C.byref(n)))
        out = np.zeros(n.value, SCORE_DTYPE)
        if n.value:
            self.sync()
            self.dev_download(p.value, out)
        return out

    def merge_shards_device(self, d_gathered: int, shard_counts: Sequence[int], ids, min_gap: int, d_merged: int, cap: int) -> int:
        world = len(shard_counts)
        counts = (C.c_size_t * world)(*[int(c) for c in shard_counts])
        ids = np.ascontiguousarray(ids, np.int32)
        n = C.c_size_t(0)
        _check(self._lib.lcm_merge_shard_scores_device(self._h, _vp(d_gathered), counts, world, _ptr(ids), len(ids), min_gap,
                                                       _vp(d_merged), cap, C.byref(n)))
        return n.value

    def launch_info(self) -> LaunchInfo:
        info = LaunchInfo()
        _check(self._lib.lcm_last_launch_info(self._h, C.byref(info)))
        return info

    # -- device scratch ----------------------------------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = _vp()
        _check(self._lib.lcm_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, d_ptr: int):
        _check(self._lib.lcm_dev_free(self._h, _vp(d_ptr)))

    def dev_upload(self, d_ptr: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        _check(self._lib.lcm_dev_upload(self._h, _vp(d_ptr), a.ctypes.data_as(_vp), a.nbytes))

    def dev_download(self, d_ptr: int, out: np.ndarray):
        assert out.flags["C_CONTIGUOUS"]
        _check(self._lib.lcm_dev_download(self._h, out.ctypes.data_as(_vp), _vp(d_ptr), out.nbytes))


def merge_shard_scores_host(shard_scores: Sequence[np.ndarray], ids, min_gap: int) -> Tuple[np.ndarray, np.ndarray]:
    """lcm_merge_shard_scores (host-only C function): W per-shard arrays -> (merged, offsets[n_frames + 1])."""
    lib = load_library()
    world = len(shard_scores)
    arrs = [np.ascontiguousarray(a, SCORE_DTYPE) for a in shard_scores]
    ptrs = (_vp * world)(*[a.ctypes.data if a.size else None for a in arrs])
    counts = (C.c_size_t * world)(*[len(a) for a in arrs])
    ids = np.ascontiguousarray(ids, np.int32)
    n = C.c_size_t(0)
    offs = np.zeros(len(ids) + 1, np.uintp)
    _check(lib.lcm_merge_shard_scores(ptrs, counts, world, _ptr(ids), len(ids), min_gap, None, 0, C.byref(n),
                                      offs.ctypes.data_as(_vp)))
    out = np.zeros(max(n.value, 1), SCORE_DTYPE)
    _check(lib.lcm_merge_shard_scores(ptrs, counts, world, _ptr(ids), len(ids), min_gap, out.ctypes.data_as(_vp),
                                      len(out), C.byref(n), None))
    return out[: n.value], offs


class Group:
    """lcm_group: one process, W devices, stored frames sharded cyclically by arrival position, RCCL inside."""

    def __init__(self, params: Optional[Params] = None, n_devices: int = 1, device_ids: Optional[Sequence[int]] = None,
                 loopback_device: Optional[int] = None, peer_copies: bool = False):
        """loopback_device: rehearsal form — n_devices shards on that ONE device, exchange steps as device-local copies.
        peer_copies: a real group whose exchange steps are device-to-device copies instead of RCCL calls."""
        self._lib = load_library()
        self._g = _vp()
        p = params if params is not None else default_params()
        if loopback_device is not None:
            _check(self._lib.lcm_group_create_loopback(C.byref(p), n_devices, loopback_device, C.byref(self._g)))
            return
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, np.int32)
        create = self._lib.lcm_group_create_peer if peer_copies else self._lib.lcm_group_create
        _check(create(C.byref(p), n_devices, None if ids is None else ids.ctypes.data_as(_i32p), C.byref(self._g)))

    @property
    def transport(self) -> str:
        return self._lib.lcm_group_transport(self._g).decode()

    def close(self):
        if getattr(self, "_g", None):
            self._lib.lcm_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return self._lib.lcm_group_db_size(self._g)

    @property
    def world(self) -> int:
        return self._lib.lcm_group_size(self._g)

    def set_params(self, p: Params):
        _check(self._lib.lcm_group_set_params(self._g, C.byref(p)))

    def reserve(self, n_frames: int, max_desc: int):
        _check(self._lib.lcm_group_reserve(self._g, n_frames, max_desc))

    def append(self, frame_id: int, desc, n_keypoints: int = -1):
        d = _rows(desc)
        _check(self._lib.lcm_group_append(self._g, frame_id, _ptr(d), d.shape[0], n_keypoints))

    def clear(self):
        _check(self._lib.lcm_group_clear(self._g))

    def all_vs_all(self) -> Tuple[np.ndarray, np.ndarray]:
        """(merged scores in (query asc, stored asc) order, offsets[len + 1])."""
        n = C.c_size_t(0)
        offs = np.zeros(len(self) + 1, np.uintp)
        _check(self._lib.lcm_group_all_vs_all(self._g, None, 0, C.byref(n), offs.ctypes.data_as(_vp)))
        out = np.zeros(max(n.value, 1), SCORE_DTYPE)
        _check(self._lib.lcm_group_all_vs_all(self._g, out.ctypes.data_as(_vp), len(out), C.byref(n), None))
        return out[: n.value], offs

    def truncate(self, n_frames: int):
        _check(self._lib.lcm_group_truncate(self._g, n_frames))

    def save(self, path: str):
        _check(self._lib.lcm_group_save(self._g, path.encode()))

    def load(self, path: str):
        _check(self._lib.lcm_group_load(self._g, path.encode()))

    def sync(self):
        _check(self._lib.lcm_group_sync(self._g))

    def set_tuning(self, knob: int, value: int):
        _check(self._lib.lcm_group_set_tuning(self._g, knob, value))

    def set_kernel_variant(self, v: int):
        _check(self._lib.lcm_group_set_kernel_variant(self._g, v))

    def all_vs_all_argmin(self, out: Optional[np.ndarray] = None, out_idx: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(merged scores, merged per-pair index checksums, offsets[len + 1]) — lcm_group_all_vs_all_argmin.
        `out` / `out_idx`: reusable host buffers (SCORE_DTYPE / uint32) of at least the pair count."""
        n = C.c_size_t(0)
        offs = np.zeros(len(self) + 1, np.uintp)
        _check(self._lib.lcm_group_all_vs_all_argmin(self._g, None, None, 0, C.byref(n), offs.ctypes.data_as(_vp)))
        if out is None or len(out) < n.value:
            out = np.zeros(max(n.value, 1), SCORE_DTYPE)
        if out_idx is None or len(out_idx) < n.value:
            out_idx = np.zeros(max(n.value, 1), np.uint32)
        _check(self._lib.lcm_group_all_vs_all_argmin(self._g, out.ctypes.data_as(_vp), out_idx.ctypes.data_as(_vp),
                                                     min(len(out), len(out_idx)), C.byref(n), None))
        return out[: n.value], out_idx[: n.value], offs

    def all_vs_all_loops(self, cap: int = 1 << 20, out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
        """(loop candidates in (current, matched) order, pairs scored) — lcm_group_all_vs_all_loops."""
        if out is None:
            out = np.zeros(max(cap, 1), CANDIDATE_DTYPE)
        n, npairs = C.c_size_t(0), C.c_size_t(0)
        _check(self._lib.lcm_group_all_vs_all_loops(self._g, out.ctypes.data_as(_vp), len(out), C.byref(n), C.byref(npairs)))
        return out[: n.value], npairs.value

    def query_submit_batch(self, queries: Sequence[np.ndarray], query_frame_ids: Sequence[int]) -> int:
        qs = [_rows(q) for q in queries]
        B = len(qs)
        ptrs = (_vp * B)(*[q.ctypes.data if q.shape[0] else None for q in qs])
        nq = np.array([q.shape[0] for q in qs], np.int32)
        ids = np.ascontiguousarray(query_frame_ids, np.int32)
        t = C.c_int32(-1)
        _check(self._lib.lcm_group_query_submit_batch(self._g, ptrs, nq.ctypes.data_as(_i32p), ids.ctypes.data_as(_i32p), B, C.byref(t)))
        return t.value

    def query_collect_batch(self, ticket: int, cap: int, n_queries: int = 16) -> Tuple[np.ndarray, np.ndarray]:
        scores = np.zeros(max(cap, 1), SCORE_DTYPE)
        offs = np.zeros(n_queries + 1, np.uintp)
        n = C.c_size_t(0)
        _check(self._lib.lcm_group_query_collect_batch(self._g, ticket, scores.ctypes.data_as(_vp), len(scores), C.byref(n),
                                                       offs.ctypes.data_as(_vp)))
        return scores[: n.value], offs

    def online_stats(self, reset: bool = False) -> OnlineStats:
        st = OnlineStats()
        _check(self._lib.lcm_group_online_stats_read(self._g, C.byref(st), 1 if reset else 0))
        return st

    def shard_launch_info(self, rank: int) -> LaunchInfo:
        """lcm_last_launch_info of shard `rank`'s own matcher (its last bulk launch: route, chunks, per-launch times)."""
        h = _vp()
        _check(self._lib.lcm_group_handle(self._g, rank, C.byref(h)))
        info = LaunchInfo()
        _check(self._lib.lcm_last_launch_info(h, C.byref(info)))
        return info

    def info(self) -> GroupInfo:
        gi = GroupInfo()
        _check(self._lib.lcm_group_last_info(self._g, C.byref(gi)))
        return gi

    def query_scores(self, query, query_frame_id: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _rows(query)
        cap = max(len(self), 1)
        scores = np.zeros(cap, SCORE_DTYPE)
        ids = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        _check(self._lib.lcm_group_query_scores(self._g, _ptr(q), q.shape[0], query_frame_id, scores.ctypes.data_as(_vp),
                                                ids.ctypes.data_as(_vp), cap, C.byref(n)))
        return scores[: n.value], ids[: n.value]

    def query_scores_batch(self, queries: Sequence[np.ndarray], query_frame_ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
        qs = [_rows(q) for q in queries]
        B = len(qs)
        ptrs = (_vp * B)(*[q.ctypes.data if q.shape[0] else None for q in qs])
        nq = np.array([q.shape[0] for q in qs], np.int32)
        ids = np.ascontiguousarray(query_frame_ids, np.int32)
        cap = max(len(self), 1) * B
        scores = np.zeros(cap, SCORE_DTYPE)
        offs = np.zeros(B + 1, np.uintp)
        n = C.c_size_t(0)
        _check(self._lib.lcm_group_query_scores_batch(self._g, ptrs, nq.ctypes.data_as(_i32p), ids.ctypes.data_as(_i32p), B,
                                                      scores.ctypes.data_as(_vp), cap, C.byref(n), offs.ctypes.data_as(_vp)))
        return scores[: n.value], offs

    def detect_loops(self, current_frame_id: int, query, n_keypoints: int = -1) -> np.ndarray:
        q = _rows(query)
        cap = max(len(self), 1)
        out = np.zeros(cap, CANDIDATE_DTYPE)
        n = C.c_int32(0)
        qp = q.ctypes.data_as(_vp) if q.shape[0] else np.zeros((1, DESC_BYTES), np.uint8).ctypes.data_as(_vp)
        _check(self._lib.lcm_group_detect_loops(self._g, current_frame_id, qp, q.shape[0], n_keypoints,
                                                out.ctypes.data_as(_vp), cap, C.byref(n)))
        return out[: n.value]


# ---------------------------------------------------------------------------------------------------------
# include/lcm_host.h: the C shim over the C++ host class loop_closing::LoopClosingSystem
# ---------------------------------------------------------------------------------------------------------
_HOST_SIGNATURES = {
    "lcs_create": (C.c_int, [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lcs_create_group": (C.c_int, [C.c_double, C.c_int, C.c_int, _i32p, C.c_int, C.POINTER(_vp)]),
    "lcs_destroy": (None, [_vp]),
    "lcs_process_frame": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "lcs_process_frames": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "lcs_match_features": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _i32p]),
    "lcs_detect_loops": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i32p]),
    "lcs_get_consecutive_matches": (C.c_int, [_vp, _vp, C.c_int, _i32p]),
    "lcs_match_loop_closures": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_int, _i32p]),
    "lcs_set_gap_by_position": (C.c_int, [_vp, C.c_int]),
    "lcs_num_frames": (C.c_int, [_vp]),
    "lcs_num_loop_closures": (C.c_int, [_vp]),
    "lcs_get_loop_closures": (C.c_int, [_vp, _vp, C.c_int, _i32p]),
    "lcs_save_results": (C.c_int, [_vp, C.c_char_p]),
}


class LoopClosingSystem:
    """Python handle on the C++ loop_closing::LoopClosingSystem (csrc/loop_closing_system.hpp) — same method names as
    the reference class (include/loop_closing.hpp:29-80), driven through the C shim."""

    def __init__(self, loop_threshold: float = 0.7, min_loop_gap: int = 30, device: int = 0, shard_rank: int = 0,
                 shard_world: int = 1, group_devices: Optional[Sequence[int]] = None, loopback_shards: int = 0):
        """group_devices: the multi-device constructor (one process, an lcm_group over those devices);
        loopback_shards > 0: its rehearsal form, that many shards on `device`."""
        self._lib = load_library()
        for name, (res, args) in _HOST_SIGNATURES.items():
            fn = getattr(self._lib, name)
            fn.restype, fn.argtypes = res, args
        self._s = _vp()
        if loopback_shards > 0:
            _check(self._lib.lcs_create_group(loop_threshold, min_loop_gap, loopback_shards, None, device, C.byref(self._s)))
        elif group_devices is not None:
            ids = np.ascontiguousarray(group_devices, np.int32)
            _check(self._lib.lcs_create_group(loop_threshold, min_loop_gap, len(ids), ids.ctypes.data_as(_i32p), -1, C.byref(self._s)))
        else:
            _check(self._lib.lcs_create(loop_threshold, min_loop_gap, device, shard_rank, shard_world, C.byref(self._s)))

    def close(self):
        if getattr(self, "_s", None):
            self._lib.lcs_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _hcheck(self, rc):
        if rc != 0:
            raise LcmError(rc, self._lib.lcm_last_error().decode("utf-8", "replace"))

    def processFrame(self, descriptors, frame_id: int, num_keypoints: int = -1):
        d = _rows(descriptors)
        self._hcheck(self._lib.lcs_process_frame(self._s, _ptr(d), d.shape[0], num_keypoints, frame_id))

    def processFrames(self, descriptor_list, frame_ids, num_keypoints=None):
        """Several frames in order, scored in micro-batches (same result as processFrame for each)."""
        ds = [_rows(d) for d in descriptor_list]
        n = len(ds)
        ptrs = (_vp * max(n, 1))(*[d.ctypes.data if d.shape[0] else None for d in ds])
        rows = np.array([d.shape[0] for d in ds], np.int32)
        ids = np.ascontiguousarray(frame_ids, np.int32)
        kps = None if num_keypoints is None else np.ascontiguousarray(num_keypoints, np.int32)
        self._hcheck(self._lib.lcs_process_frames(self._s, ptrs, _ptr(rows), _ptr(kps), _ptr(ids), n))

    def matchFeatures(self, frame1_id: int, frame2_id: int, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, DMATCH_DTYPE)
        n = C.c_int32(0)
        self._hcheck(self._lib.lcs_match_features(self._s, frame1_id, frame2_id, out.ctypes.data_as(_vp), cap, C.byref(n)))
        return out[: n.value]

    def detectLoops(self, current_frame_id: int) -> np.ndarray:
        cap = max(self._lib.lcs_num_frames(self._s), 1)
        out = np.zeros(cap, CANDIDATE_DTYPE)
        n = C.c_int32(0)
        self._hcheck(self._lib.lcs_detect_loops(self._s, current_frame_id, out.ctypes.data_as(_vp), cap, C.byref(n)))
        return out[: n.value]

    def getLoopClosures(self) -> np.ndarray:
        cap = max(self._lib.lcs_num_loop_closures(self._s), 1)
        out = np.zeros(cap, CANDIDATE_DTYPE)
        n = C.c_int32(0)
        self._hcheck(self._lib.lcs_get_loop_closures(self._s, out.ctypes.data_as(_vp), cap, C.byref(n)))
        return out[: n.value]

    def getConsecutiveMatches(self, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, DMATCH_DTYPE)
        n = C.c_int32(0)
        self._hcheck(self._lib.lcs_get_consecutive_matches(self._s, out.ctypes.data_as(_vp), cap, C.byref(n)))
        return out[: n.value]

    def matchLoopClosures(self, current_frame_id: int, cap: int = 1 << 20):
        out = np.zeros(cap, DMATCH_DTYPE)
        offs = np.zeros(4096, np.uintp)
        n = C.c_int32(0)
        self._hcheck(self._lib.lcs_match_loop_closures(self._s, current_frame_id, out.ctypes.data_as(_vp), cap,
                                                       offs.ctypes.data_as(_vp), len(offs), C.byref(n)))
        return [out[int(offs[i]): int(offs[i + 1])] for i in range(n.value)]

    def setGapByPosition(self, on: bool = True):
        """Count min_loop_gap on arrival positions (the tree's own loop, src/main.cpp:1375-1379) instead of frame ids."""
        self._hcheck(self._lib.lcs_set_gap_by_position(self._s, 1 if on else 0))

    def numFrames(self) -> int:
        return self._lib.lcs_num_frames(self._s)

    def saveResults(self, output_dir: str):
        self._hcheck(self._lib.lcs_save_results(self._s, output_dir.encode()))
