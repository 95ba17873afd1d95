"""CPU oracle for the tvidz inspector hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (tvidz_amd/) never does and has no CPU fallback.

Two layers:
  * pure-Python restatements with the reference's own loop shape
    (find_duplicates_py  <- /root/reference/inspector/db.py:85-91,
     streaming_verdict_py <- /root/reference/inspector/app.py:228-255);
  * ctypes bindings to oracle/tvz_oracle.c (same semantics in plain C, fast
    enough for frame data and for the CPU baseline).

Parity status (details in tvz_oracle.c):
  matcher + streaming verdict : PINNED (reference KAT test_app.py:66-83 and
                                golden fixtures generated from the reference's
                                db.find_duplicates, see oracle/gen_golden.py)
  scene score + pts_time text : PARITY UNPINNED (arithmetic lives in the
                                unpinned external ffmpeg binary; no reference
                                test or fixture covers it)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtvz_oracle.so")
_lib = None

INT32_MAX = 2**31 - 1
PTS_POLICY_G6 = 0      # FFmpeg <= 6.x  "%.6g"
PTS_POLICY_F6TRIM = 1  # FFmpeg >= 7.0  "%.*f" trimmed


def build(force: bool = False) -> str:
    """Compile oracle/tvz_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "tvz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O3", "-fPIC", "-std=c11", "-fvisibility=hidden", "-shared",
             "-o", _SO, src, "-lm"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        c = ctypes
        L.orc_luma_sad_u8.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_int32,
                                      c.c_int64, c.c_int64, c.c_void_p]
        L.orc_luma_sad_u16.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_int32,
                                       c.c_int64, c.c_int64, c.c_void_p]
        L.orc_luma_sad_u8_range.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_int32,
                                            c.c_int32, c.c_int64, c.c_int64, c.c_void_p]
        L.orc_scene_select.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_int32, c.c_int32,
                                       c.c_double, c.c_double, c.c_int32, c.c_void_p,
                                       c.c_void_p, c.c_void_p, c.c_void_p]
        L.orc_fmt_pts_time.argtypes = [c.c_int64, c.c_int32, c.c_int32, c.c_int32,
                                       c.c_char_p, c.c_int32]
        L.orc_pts_time_value.argtypes = [c.c_int64, c.c_int32, c.c_int32, c.c_int32]
        L.orc_pts_time_value.restype = c.c_double
        L.orc_find_duplicates.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p,
                                          c.c_void_p, c.c_int64, c.c_int32, c.c_void_p,
                                          c.c_void_p]
        L.orc_find_duplicates.restype = c.c_int64
        L.orc_match_kth.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_int64,
                                    c.c_int32, c.c_void_p, c.c_void_p]
        L.orc_match_kth_sorted.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p,
                                           c.c_int64, c.c_int64, c.c_int32, c.c_void_p,
                                           c.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------- scene score

def luma_sad(luma: np.ndarray) -> np.ndarray:
    """uint8 or uint16 [T,H,W] (any strides with contiguous pixels in a row) -> uint64[T]; sad[0]=0."""
    assert luma.dtype in (np.uint8, np.uint16) and luma.ndim == 3 and luma.strides[2] == luma.itemsize
    T, H, W = luma.shape
    out = np.zeros(T, dtype=np.uint64)
    if T:
        fn = lib().orc_luma_sad_u8 if luma.dtype == np.uint8 else lib().orc_luma_sad_u16
        fn(_p(luma), T, H, W, luma.strides[0], luma.strides[1], _p(out))
    return out


def luma_sad_range(luma: np.ndarray, t0: int, t1: int, out: np.ndarray) -> None:
    T, H, W = luma.shape
    lib().orc_luma_sad_u8_range(_p(luma), t0, t1, H, W, luma.strides[0], luma.strides[1], _p(out))


def scene_select(sad: np.ndarray, H: int, W: int, threshold: float = 0.3, bitdepth: int = 8,
                 prev_mafd: float = 0.0, have_prev: bool = False):
    """-> (selected uint8[T], score f64[T], mafd f64[T], last_mafd)."""
    sad = np.ascontiguousarray(sad, dtype=np.uint64)
    T = sad.shape[0]
    sel = np.zeros(T, dtype=np.uint8)
    score = np.zeros(T, dtype=np.float64)
    mafd = np.zeros(T, dtype=np.float64)
    last = ctypes.c_double(prev_mafd)
    lib().orc_scene_select(_p(sad), T, H, W, bitdepth, threshold, prev_mafd, int(have_prev),
                           _p(sel), _p(score), _p(mafd), ctypes.byref(last))
    return sel, score, mafd, last.value


def scene_select_py(sad: Sequence[int], H: int, W: int, threshold: float = 0.3,
                    bitdepth: int = 8):
    """Pure-Python/numpy restatement of f_select.c get_scene_score (small cases)."""
    prev = 0.0
    sel, score = [], []
    for t, s in enumerate(sad):
        if t == 0:
            sel.append(0); score.append(0.0); continue
        mafd = float(int(s)) / float(W * H) / float(1 << (bitdepth - 8))
        diff = abs(mafd - prev)
        m = diff if mafd > diff else mafd
        f = np.float32(m / 100.0)
        f = np.float32(0.0) if f < 0 else (np.float32(1.0) if f > 1 else f)
        prev = mafd
        score.append(float(f))
        sel.append(1 if float(f) > threshold else 0)
    return np.array(sel, dtype=np.uint8), np.array(score, dtype=np.float64)


def fmt_pts_time(pts: int, tb_num: int, tb_den: int, policy: int = PTS_POLICY_G6) -> str:
    buf = ctypes.create_string_buffer(64)
    lib().orc_fmt_pts_time(pts, tb_num, tb_den, policy, buf, 64)
    return buf.value.decode()


def pts_time_value(pts: int, tb_num: int, tb_den: int, policy: int = PTS_POLICY_G6) -> float:
    return lib().orc_pts_time_value(pts, tb_num, tb_den, policy)


# -------------------------------------------------------------------- matcher

def find_duplicates_py(corpus: Iterable[Tuple[int, Sequence[float]]],
                       new_timestamps: Sequence[float], min_match: int = 5):
    """Same loop shape as /root/reference/inspector/db.py:85-91."""
    results = []
    for video_id, cand_ts in corpus:
        match_count = 0
        for new_ts in new_timestamps:
            if new_ts in cand_ts:
                match_count += 1
        if match_count >= min_match:
            results.append((video_id, match_count))
    return results


def streaming_verdict_py(ts_stream: Iterable[float],
                         corpus: List[Tuple[int, List[float]]],
                         self_id: int, min_match: int = 2):
    """Replay of /root/reference/inspector/app.py:228-255 over an in-memory corpus.

    `corpus` is mutated the way add_timestamps (db.py:43-64) would: the row of
    `self_id` is upserted with the growing prefix before every match.
    Returns (scene_timestamps, dup_ids) — dup_ids in corpus order, [] if none.
    """
    scene_timestamps: List[float] = []
    row = None
    for r in corpus:
        if r[0] == self_id:
            row = r
    for ts in ts_stream:
        if not scene_timestamps or ts != scene_timestamps[-1]:          # app.py:231
            scene_timestamps.append(ts)                                  # :232
            if row is None:                                              # :234 upsert
                row = (self_id, [])
                corpus.append(row)
            row[1][:] = scene_timestamps
            dups = find_duplicates_py(corpus, scene_timestamps, min_match)  # :235
            dups = [d for d in dups if d[0] != self_id]                  # :237
            if dups:                                                     # :238
                return scene_timestamps, [d[0] for d in dups]
    return scene_timestamps, []


def _csr(corpus: Sequence[Tuple[int, Sequence[float]]]):
    ids = np.array([v for v, _ in corpus], dtype=np.int32)
    lens = np.array([len(t) for _, t in corpus], dtype=np.int64)
    offs = np.zeros(len(corpus) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    keys = np.zeros(max(int(offs[-1]), 1), dtype=np.float64)
    pos = 0
    for _, t in corpus:
        keys[pos:pos + len(t)] = np.asarray(t, dtype=np.float64)
        pos += len(t)
    return ids, offs, keys


def find_duplicates_c(corpus, new_timestamps, min_match=5):
    ids, offs, keys = _csr(corpus)
    q = np.ascontiguousarray(np.asarray(new_timestamps, dtype=np.float64))
    if q.size == 0:
        q = np.zeros(1, dtype=np.float64)[:0]
    out_ids = np.zeros(max(len(ids), 1), dtype=np.int32)
    out_cnt = np.zeros(max(len(ids), 1), dtype=np.int32)
    qq = np.zeros(max(q.size, 1), dtype=np.float64); qq[:q.size] = q
    n = lib().orc_find_duplicates(_p(qq), q.size, _p(offs), _p(keys), _p(ids), len(ids),
                                  min_match, _p(out_ids), _p(out_cnt))
    return [(int(out_ids[i]), int(out_cnt[i])) for i in range(n)]


def match_kth_csr(query: np.ndarray, offs: np.ndarray, keys: np.ndarray, min_match: int,
                  sorted_unique: bool = False):
    """(count int32[C], kth int32[C]) for one query against a CSR corpus."""
    C = len(offs) - 1
    cnt = np.zeros(max(C, 1), dtype=np.int32)
    kth = np.zeros(max(C, 1), dtype=np.int32)
    q = np.zeros(max(len(query), 1), dtype=np.float64); q[:len(query)] = query
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    keys = np.ascontiguousarray(keys, dtype=np.float64)
    if keys.size == 0:
        keys = np.zeros(1, dtype=np.float64)
    if sorted_unique:
        lib().orc_match_kth_sorted(_p(q), len(query), _p(offs), _p(keys), 0, C, min_match,
                                   _p(cnt), _p(kth))
    else:
        lib().orc_match_kth(_p(q), len(query), _p(offs), _p(keys), C, min_match,
                            _p(cnt), _p(kth))
    return cnt[:C], kth[:C]


def match_kth(corpus, query, min_match):
    ids, offs, keys = _csr(corpus)
    cnt, kth = match_kth_csr(np.asarray(query, dtype=np.float64), offs, keys, min_match)
    return ids, cnt, kth


def verdict_from_kth(ids: np.ndarray, kth: np.ndarray, self_id: int = -1):
    """Batch equivalent of app.py:235-255: (k*, sorted dup ids) or (None, [])."""
    mask = (ids != self_id) & (kth < INT32_MAX)
    if not mask.any():
        return None, []
    kstar = int(kth[mask].min())
    return kstar, sorted(int(v) for v in ids[mask & (kth == kstar)])


def align_py(corpus, query, eps=0.1, max_offset=60.0):
    """Restatement of tvz_align (an opt-in extra with no reference counterpart): difference
    histogram per row, best bin with ties to the smaller |bin| then the negative one.
    Rows are treated as sets (sorted unique, NaN dropped), as the device corpus stores them."""
    import math
    B = int(math.floor(max_offset / eps + 0.5))
    out = []
    q = np.asarray([x for x in query], dtype=np.float64)
    q = q[~np.isnan(q)]
    for vid, ts in corpus:
        c = np.asarray(ts, dtype=np.float64)
        c = np.unique(c[~np.isnan(c)] + 0.0)
        hist = np.zeros(2 * B + 1, dtype=np.int64)
        if len(c) and len(q):
            d = np.floor((c[:, None] - q[None, :]) / eps + 0.5)
            d = d[(d >= -B) & (d <= B)].astype(np.int64)
            np.add.at(hist, d + B, 1)
        best_key, best_bin = -1, 0
        for b in range(2 * B + 1):
            bn = b - B
            order = 2 * abs(bn) + (1 if bn > 0 else 0)
            key = (int(hist[b]) << 14) | (16383 - order)
            if key > best_key:
                best_key, best_bin = key, bn
        out.append((int(vid), int(len(c)), int(best_bin), int(hist[best_bin + B]), int(hist[B])))
    return out
