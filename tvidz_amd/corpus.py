"""Device-resident image of the `video_timestamps` table and the matcher over it.

Host side of the seam at /root/reference/inspector/db.py:76-94 (`find_duplicates`) and of the
per-cut loop around it (inspector/app.py:231-255).  All compute is in csrc/tvz_match.hip; this
module only marshals arrays through the C ABI of include/tvz.h.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

KTH_NEVER = _lib.KTH_NEVER


def rows_to_csr(rows: Iterable[Tuple[int, Sequence[float]]]):
    """[(video_id, [ts...])] -> (ids int32[C], offsets int64[C+1], keys float64[n])."""
    rows = list(rows)
    ids = np.fromiter((int(v) for v, _ in rows), dtype=np.int32, count=len(rows))
    lens = np.fromiter((len(t) for _, t in rows), dtype=np.int64, count=len(rows))
    offs = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    keys = np.empty(int(offs[-1]), dtype=np.float64)
    for (_, t), o, n in zip(rows, offs[:-1], lens):
        if n:
            keys[o:o + n] = np.asarray(t, dtype=np.float64)
    return ids, offs, keys


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else C.c_void_p(a.ctypes.data)


def workspace_bytes(Q: int, max_query_len: int, cap: int = 0, k: int = 0, n_ranks: int = 1,
                    total_query_keys: int = 0) -> int:
    """tvz_match_workspace_bytes: scratch of the batched calls (k = 0: the hash-join tables only).
    `total_query_keys` (the length of d_queries) matters for a batch that holds queries of more than 4,095
    timestamps: their sorted distinct keys are made on the device in a tail of the workspace
    (tvz_match_workspace_bytes_long)."""
    if max_query_len > 4095 and total_query_keys:
        return int(_lib.load().tvz_match_workspace_bytes_long(int(Q), int(max_query_len), int(cap), int(k),
                                                              int(n_ranks), int(total_query_keys)))
    return int(_lib.load().tvz_match_workspace_bytes(int(Q), int(max_query_len), int(cap), int(k),
                                                     int(n_ranks)))


class DeviceCorpus:
    """tvz_corpus handle: rows of (video_id, sorted-unique canonical float64 keys) in HBM."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        self.device = int(device)
        h = C.c_void_p()
        _lib.check(self.lib.tvz_corpus_create(C.byref(h), self.device))
        self._h = h
        self._row_bound = 0                 # upper bound on the row count (sizes the output arrays)
        self._tls = threading.local()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.tvz_corpus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- mutation (db.py:43-64 add_timestamps; app.py:325-333 clear-db) ----
    def reserve(self, n_rows: int, n_keys: int) -> None:
        _lib.check(self.lib.tvz_corpus_reserve(self._h, int(n_rows), int(n_keys)))

    def upload_csr(self, ids: np.ndarray, offsets: np.ndarray, keys: np.ndarray) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        if offsets.size != ids.size + 1:
            raise RuntimeError("offsets must have len(ids)+1 entries")
        _lib.check(self.lib.tvz_corpus_upload(self._h, _ptr(ids), C.c_void_p(offsets.ctypes.data),
                                              _ptr(keys), ids.size, keys.size))
        self._row_bound = int(ids.size)

    def upload(self, rows: Iterable[Tuple[int, Sequence[float]]]) -> None:
        self.upload_csr(*rows_to_csr(rows))

    def upsert(self, video_id: int, timestamps: Sequence[float]) -> None:
        """Stream-ordered on the device: returns without waiting for matches in flight; matches
        enqueued afterwards see the new row (include/tvz.h tvz_corpus_upsert)."""
        k = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64))
        _lib.check(self.lib.tvz_corpus_upsert(self._h, int(video_id), _ptr(k), k.size))
        self._row_bound += 1                # may over-count (a replaced row): only a size bound

    def clear(self) -> None:
        _lib.check(self.lib.tvz_corpus_clear(self._h))
        self._row_bound = 0

    def build_index(self) -> None:
        """Rebuild the inverted index over the current rows now (upload builds it, and it is rebuilt
        automatically as the delta table fills); waits for matches in flight."""
        _lib.check(self.lib.tvz_corpus_build_index(self._h))

    def index_stats(self) -> dict:
        """indexed_rows / delta_rows / postings / distinct_keys / builds; zeros while there is no index."""
        v = [C.c_int64() for _ in range(5)]
        _lib.check(self.lib.tvz_corpus_index_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("indexed_rows", "delta_rows", "postings", "distinct_keys", "builds"),
                        (int(x.value) for x in v)))

    def bucket_stats(self) -> dict:
        """The bucket directory of a one-sub-index handle (tvz_corpus_bucket_stats): buckets (0: none), keys outside
        their home bucket, the farthest walk, keys with external lists, external postings, sub-indexes."""
        v = (C.c_int64 * 6)()
        _lib.check(self.lib.tvz_corpus_bucket_stats(self._h, v))
        return dict(zip(("buckets", "keys_walked_on", "max_walk", "external_lists", "external_postings", "sub_indexes"),
                        (int(x) for x in v)))

    def stats(self) -> Tuple[int, int, int]:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self.lib.tvz_corpus_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- single query, host in / host out: the db.find_duplicates drop-in ----
    def _out_buffers(self, cap: int):
        b = getattr(self._tls, "buf", None)
        if b is None or b[0].size < cap:
            n = max(cap, 1024)
            b = (np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32),
                 C.c_int64())
            b = b + (C.c_void_p(b[0].ctypes.data), C.c_void_p(b[1].ctypes.data), C.c_void_p(b[2].ctypes.data),
                     C.byref(b[3]))
            self._tls.buf = b
        return b

    def find_duplicates(self, new_timestamps: Sequence[float], min_match: int = 5,
                        exclude_id: int = -1, with_kth: bool = False):
        """One kernel launch + one stream synchronisation inside the library (for queries of up to
        4095 timestamps with min_match <= 5); the output arrays are per-thread and reused."""
        q = np.ascontiguousarray(np.asarray(new_timestamps, dtype=np.float64))
        cap = max(self._row_bound, 1)
        while True:
            ids, cnt, kth, n, p_ids, p_cnt, p_kth, p_n = self._out_buffers(cap)
            _lib.check(self.lib.tvz_find_duplicates(self._h, _ptr(q), q.size, int(min_match),
                                                    int(exclude_id), ids.size, p_ids, p_cnt, p_kth, p_n))
            if n.value <= ids.size:
                break
            cap = int(n.value)  # rows were added concurrently: retry with room for all
        m = n.value
        if with_kth:
            return list(zip(ids[:m].tolist(), cnt[:m].tolist(), kth[:m].tolist()))
        return list(zip(ids[:m].tolist(), cnt[:m].tolist()))

    # ---- opt-in alignment score (never the verdict; see include/tvz.h tvz_align) ----
    def align(self, timestamps: Sequence[float], eps: float = 0.1, max_offset: float = 60.0):
        """-> int32 array [n_rows,5]: (video_id, row_len, best_bin, votes, votes_at_zero_shift)."""
        n_rows = self.stats()[0]
        dev = torch.device("cuda", self.device)
        q = torch.as_tensor(np.asarray(timestamps, dtype=np.float64)).to(dev)
        out = torch.empty((max(n_rows, 1), 5), dtype=torch.int32, device=dev)
        _lib.check(self.lib.tvz_align(self._h, q.data_ptr() if q.numel() else None, q.numel(),
                                      float(eps), float(max_offset), out.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream))
        return out[:n_rows].cpu().numpy()

    # ---- batched, device resident ----
    def _check_queries(self, d_queries, d_q_offsets):
        dev = d_queries.device
        if dev.type != "cuda" or dev.index != self.device:
            raise RuntimeError(f"queries must live on cuda:{self.device}")
        if d_queries.dtype != torch.float64 or d_q_offsets.dtype != torch.int64:
            raise RuntimeError("queries must be float64 and offsets int64")
        return dev, d_q_offsets.numel() - 1

    def _workspace(self, workspace, need: int, dev, stream):
        if workspace is None:
            # per call, from torch's caching allocator (no hipMalloc once warm); pass a persistent
            # one to keep even that off the hot path
            workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
            workspace.record_stream(stream)
        elif workspace.device != dev or workspace.dtype != torch.uint8 or workspace.numel() < need:
            raise RuntimeError(f"workspace must be a uint8 tensor of >= {need} bytes on {dev}")
        return workspace

    def match(self, d_queries: torch.Tensor, d_q_offsets: torch.Tensor, max_query_len: int,
              min_match: int, cap: int, d_exclude_ids: Optional[torch.Tensor] = None,
              out_hits: Optional[torch.Tensor] = None, out_n: Optional[torch.Tensor] = None,
              stream: Optional[torch.cuda.Stream] = None, workspace: Optional[torch.Tensor] = None,
              algo: int = _lib.ALGO_AUTO):
        """Enqueue Q queries; returns (hits int32[Q,cap,3], hits_n int32[Q]) device tensors.
        `algo`: per-call kernel choice (_lib.ALGO_*); results never depend on it."""
        dev, Q = self._check_queries(d_queries, d_q_offsets)
        if out_hits is None:
            out_hits = torch.empty((Q, cap, 3), dtype=torch.int32, device=dev)
        if out_n is None:
            out_n = torch.empty(Q, dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        ws = self._workspace(workspace, workspace_bytes(Q, max_query_len, total_query_keys=d_queries.numel()), dev, s)
        _lib.check(self.lib.tvz_match(
            self._h, d_queries.data_ptr(), d_q_offsets.data_ptr(), Q, int(max_query_len),
            int(min_match), d_exclude_ids.data_ptr() if d_exclude_ids is not None else None,
            int(cap), out_hits.data_ptr(), out_n.data_ptr(), ws.data_ptr(), ws.numel(), int(algo),
            s.cuda_stream))
        return out_hits, out_n

    def match_topk(self, d_queries: torch.Tensor, d_q_offsets: torch.Tensor, max_query_len: int,
                   min_match: int, cap: int, k: int, d_exclude_ids: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, stream: Optional[torch.cuda.Stream] = None,
                   workspace: Optional[torch.Tensor] = None, algo: int = _lib.ALGO_AUTO) -> torch.Tensor:
        """Sweep + per-shard top-k behind ONE library call (hit lists stay in the workspace):
        -> int32 [Q,k+1,3] = the k best hits by (kth, video_id, count) + a (-1, n_hits, NEVER) row."""
        dev, Q = self._check_queries(d_queries, d_q_offsets)
        if out is None:
            out = torch.empty((Q, k + 1, 3), dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        ws = self._workspace(workspace, workspace_bytes(Q, max_query_len, cap, k, total_query_keys=d_queries.numel()), dev, s)
        _lib.check(self.lib.tvz_match_topk(
            self._h, d_queries.data_ptr(), d_q_offsets.data_ptr(), Q, int(max_query_len),
            int(min_match), d_exclude_ids.data_ptr() if d_exclude_ids is not None else None,
            int(cap), int(k), out.data_ptr(), ws.data_ptr(), ws.numel(), int(algo), s.cuda_stream))
        return out


class Comm:
    """tvz_comm handle: the RCCL communicator of the sharded match, owned by libtvz.so (a non-Python
    host gets the same path through include/tvz.h).  `unique_id()` on rank 0, ship the 128 bytes to
    the other ranks by any means, then Comm(id, n_ranks, rank, device) on every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(_lib.load().tvz_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, n_ranks: int, rank: int, device: int = 0):
        self.lib = _lib.load()
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise RuntimeError("unique id must be 128 bytes")
        h = C.c_void_p()
        _lib.check(self.lib.tvz_comm_init(C.byref(h), C.c_char_p(unique_id), int(n_ranks), int(rank),
                                          int(device)))
        self._h = h
        self.n_ranks, self.rank, self.device = int(n_ranks), int(rank), int(device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.tvz_comm_destroy(self._h)
            self._h = None

    def info(self) -> Tuple[int, int]:
        """(ranks, rank) as the RCCL communicator inside the library reports them (tvz_comm_info)."""
        n, r = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.tvz_comm_info(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def match_sharded(self, corpus: DeviceCorpus, d_queries: torch.Tensor, d_q_offsets: torch.Tensor,
                      max_query_len: int, min_match: int, cap: int, k: int,
                      d_exclude_ids: Optional[torch.Tensor] = None,
                      workspace: Optional[torch.Tensor] = None,
                      stream: Optional[torch.cuda.Stream] = None, algo: int = _lib.ALGO_AUTO,
                      out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """local match + top-k -> ncclAllGather -> merge, all enqueued on `stream` by ONE library
        call; -> (merged int32 [Q,k,3], totals int32 [Q]), identical on every rank.  `out` =
        (merged, totals) buffers to write into (a caller streaming batches keeps its own)."""
        dev, Q = corpus._check_queries(d_queries, d_q_offsets)
        if out is not None:
            merged, totals = out
            if merged.shape != (Q, k, 3) or totals.shape != (Q,) or merged.dtype != torch.int32 \
                    or totals.dtype != torch.int32 or not merged.is_contiguous():
                raise RuntimeError("out must be (int32 [Q,k,3], int32 [Q])")
        else:
            merged = torch.empty((Q, k, 3), dtype=torch.int32, device=dev)
            totals = torch.empty(Q, dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        ws = corpus._workspace(workspace, workspace_bytes(Q, max_query_len, cap, k, self.n_ranks, d_queries.numel()), dev, s)
        _lib.check(self.lib.tvz_match_sharded(
            corpus._h, self._h, d_queries.data_ptr(), d_q_offsets.data_ptr(), Q, int(max_query_len),
            int(min_match), d_exclude_ids.data_ptr() if d_exclude_ids is not None else None,
            int(cap), int(k), merged.data_ptr(), totals.data_ptr(), ws.data_ptr(), ws.numel(),
            int(algo), s.cuda_stream))
        return merged, totals


def topk(lists: torch.Tensor, lists_n: Optional[torch.Tensor], k: int,
         out: Optional[torch.Tensor] = None, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """Per-query k best hits ordered by (kth, video_id, count).
    lists: int32 [Q,cap,3] or [n_lists,Q,cap,3] (all-gathered shards); -> int32 [Q,k,3]."""
    if lists.dim() == 3:
        lists = lists.unsqueeze(0)
    if lists.dtype != torch.int32 or lists.device.type != "cuda" or not lists.is_contiguous():
        raise RuntimeError("lists must be a contiguous int32 CUDA tensor")
    n_lists, Q, cap, _ = lists.shape
    if out is None:
        out = torch.empty((Q, k, 3), dtype=torch.int32, device=lists.device)
    s = stream if stream is not None else torch.cuda.current_stream(lists.device)
    with torch.cuda.device(lists.device):
        _lib.check(_lib.load().tvz_topk(lists.data_ptr(),
                                        lists_n.data_ptr() if lists_n is not None else None,
                                        n_lists, Q, cap, k, out.data_ptr(), s.cuda_stream))
    return out


def topk_shard(hits: torch.Tensor, hits_n: torch.Tensor, k: int,
               stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """Local lists -> int32 [Q,k+1,3]: the k best + a (-1, n_hits, NEVER) totals row."""
    Q, cap, _ = hits.shape
    out = torch.empty((Q, k + 1, 3), dtype=torch.int32, device=hits.device)
    s = stream if stream is not None else torch.cuda.current_stream(hits.device)
    with torch.cuda.device(hits.device):
        _lib.check(_lib.load().tvz_topk_shard(hits.data_ptr(), hits_n.data_ptr(), Q, cap, k,
                                              out.data_ptr(), s.cuda_stream))
    return out


def match_topk_shards(shards: Sequence["DeviceCorpus"], d_queries: torch.Tensor, d_q_offsets: torch.Tensor,
                      max_query_len: int, min_match: int, cap: int, k: int, workspace: torch.Tensor,
                      d_exclude_ids: Optional[torch.Tensor] = None, stream: Optional[torch.cuda.Stream] = None,
                      algo: int = _lib.ALGO_AUTO):
    """tvz_match_topk on every handle of ONE device + the merge behind one library call:
    -> (blocks int32 [R,Q,k+1,3], merged int32 [Q,k,3], totals int32 [Q])."""
    dev = d_queries.device
    R, Q = len(shards), d_q_offsets.numel() - 1
    blocks = torch.empty((R, Q, k + 1, 3), dtype=torch.int32, device=dev)
    merged = torch.empty((Q, k, 3), dtype=torch.int32, device=dev)
    totals = torch.empty(Q, dtype=torch.int32, device=dev)
    handles = (C.c_void_p * R)(*[s._h for s in shards])
    s = stream if stream is not None else torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tvz_match_topk_shards(
            handles, R, d_queries.data_ptr(), d_q_offsets.data_ptr(), Q, int(max_query_len), int(min_match),
            d_exclude_ids.data_ptr() if d_exclude_ids is not None else None, int(cap), int(k), blocks.data_ptr(),
            merged.data_ptr(), totals.data_ptr(), workspace.data_ptr(), workspace.numel(), int(algo), s.cuda_stream))
    return blocks, merged, totals


def topk_merge(gathered: torch.Tensor, k: int, stream: Optional[torch.cuda.Stream] = None):
    """All-gathered int32 [R,Q,k+1,3] -> (merged int32 [Q,k,3], totals int32 [Q])."""
    R, Q, k1, _ = gathered.shape
    if k1 != k + 1 or not gathered.is_contiguous():
        raise RuntimeError("gathered must be contiguous [R,Q,k+1,3]")
    out = torch.empty((Q, k, 3), dtype=torch.int32, device=gathered.device)
    totals = torch.empty(Q, dtype=torch.int32, device=gathered.device)
    s = stream if stream is not None else torch.cuda.current_stream(gathered.device)
    with torch.cuda.device(gathered.device):
        _lib.check(_lib.load().tvz_topk_merge(gathered.data_ptr(), R, Q, k, out.data_ptr(),
                                              totals.data_ptr(), s.cuda_stream))
    return out, totals


def pack_queries(queries: Sequence[Sequence[float]], device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Host lists -> (float64 keys, int64 offsets, max_len) on `device`."""
    lens = np.fromiter((len(q) for q in queries), dtype=np.int64, count=len(queries))
    offs = np.zeros(len(queries) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.empty(max(int(offs[-1]), 1), dtype=np.float64)
    for q, o, n in zip(queries, offs[:-1], lens):
        if n:
            flat[o:o + n] = np.asarray(q, dtype=np.float64)
    return (torch.from_numpy(flat).to(device), torch.from_numpy(offs).to(device),
            int(lens.max()) if len(queries) else 0)
