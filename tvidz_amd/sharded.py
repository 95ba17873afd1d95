"""Corpus match sharded over the GPUs of one node.

The candidate axis shards naturally: rows are independent.  Each rank (one process per GPU)
holds a contiguous range of the `video_timestamps` rows balanced by key count, queries are
replicated (tiny), every rank runs the local match + per-shard top-k, and ONE all-gather of
[Q, k+1] int32 triples (the extra row carries the shard's hit total) (RCCL over xGMI; <= 768 B per query per rank, latency-bound) is followed by
the same k-way merge on every rank.  There is no other collective on the data path.

The reference has no counterpart (it scans one Postgres table in one Python process,
inspector/db.py:83-91); the merged result is what its find_duplicates + the first-hit rule of
inspector/app.py:235-255 would report, ordered by (kth, video_id).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

KTH_NEVER = 0x7FFFFFFF


def shard_bounds(offsets: np.ndarray, world: int) -> np.ndarray:
    """Row boundaries [world+1] of contiguous shards with ~equal key counts (not row counts)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    C = offsets.size - 1
    total = int(offsets[-1])
    targets = (np.arange(world + 1, dtype=np.float64) * total / world)
    b = np.searchsorted(offsets, targets, side="left").astype(np.int64)
    b[0], b[-1] = 0, C
    return np.maximum.accumulate(np.clip(b, 0, C))


def shard_csr(ids: np.ndarray, offsets: np.ndarray, keys: np.ndarray, rank: int, world: int):
    b = shard_bounds(offsets, world)
    r0, r1 = int(b[rank]), int(b[rank + 1])
    k0, k1 = int(offsets[r0]), int(offsets[r1])
    return ids[r0:r1], offsets[r0:r1 + 1] - k0, keys[k0:k1]


class HipBackend:
    """The product backend: DeviceCorpus.match + the top-k kernels on this rank's GPU."""

    def __init__(self, corpus):
        from . import corpus as tc
        self._tc = tc
        self.corpus = corpus

    def match(self, d_q, d_off, max_len, min_match, cap, d_excl):
        return self.corpus.match(d_q, d_off, max_len, min_match, cap, d_exclude_ids=d_excl)

    def local_topk(self, d_q, d_off, max_len, min_match, cap, k, d_excl):
        """sweep + per-shard top-k behind one library call (tvz_match_topk)."""
        return self.corpus.match_topk(d_q, d_off, max_len, min_match, cap, k, d_exclude_ids=d_excl)

    def topk_shard(self, hits, hits_n, k):
        return self._tc.topk_shard(hits, hits_n, k)

    def topk_merge(self, gathered, k):
        return self._tc.topk_merge(gathered, k)


class ShardedMatcher:
    """rank-local shard + ONE all-gather of per-shard top-k (+ hit totals) + identical merge on
    every rank."""

    def __init__(self, backend, k: int = 64, cap: int = 1024, group=None,
                 always_collective: bool = False):
        self.backend = backend
        self.k = int(k)
        self.cap = max(int(cap), self.k)
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        # run the all-gather even in a 1-rank group (used to rehearse the RCCL path on one GPU)
        self.collective = self.world > 1 or (always_collective and inited)

    def submit(self, d_queries: torch.Tensor, d_q_offsets: torch.Tensor, max_query_len: int,
               min_match: int, d_exclude_ids: Optional[torch.Tensor] = None):
        """Enqueue local match + per-shard top-k and START the all-gather; returns a ticket for
        finish().  Submitting batch i+1 before finishing batch i overlaps the collective of one
        batch with the match kernels of the next (RCCL runs on its own stream)."""
        if hasattr(self.backend, "local_topk"):
            local = self.backend.local_topk(d_queries, d_q_offsets, max_query_len, min_match, self.cap,
                                            self.k, d_exclude_ids)     # [Q, k+1, 3]
        else:
            hits, n = self.backend.match(d_queries, d_q_offsets, max_query_len, min_match, self.cap,
                                         d_exclude_ids)
            local = self.backend.topk_shard(hits, n, self.k)          # [Q, k+1, 3]
        Q = local.shape[0]
        if not self.collective:
            return (local.view(1, Q, self.k + 1, 3), None, local)
        # dim-0 concatenation is the layout both RCCL and gloo accept for all_gather_into_tensor
        flat = torch.empty((self.world * Q, self.k + 1, 3), dtype=torch.int32, device=local.device)
        work = dist.all_gather_into_tensor(flat, local.contiguous(), group=self.group, async_op=True)
        return (flat.view(self.world, Q, self.k + 1, 3), work, local)

    def finish(self, ticket):
        """-> (merged int32 [Q,k,3] of (video_id, count, kth), total_hits int32 [Q]) — identical on
        every rank.  total_hits > k means the list was truncated to the k best; a NEGATIVE total
        means a shard's hit list overflowed `cap` (re-run that query with a larger cap)."""
        gathered, work, _keepalive = ticket
        if work is not None:
            work.wait()                     # makes the current stream wait for the collective
        return self.backend.topk_merge(gathered, self.k)

    def match_topk(self, d_queries: torch.Tensor, d_q_offsets: torch.Tensor, max_query_len: int,
                   min_match: int, d_exclude_ids: Optional[torch.Tensor] = None):
        return self.finish(self.submit(d_queries, d_q_offsets, max_query_len, min_match, d_exclude_ids))


def make_comm(device: int, group=None):
    """Create the libtvz RCCL communicator of this rank.  torch.distributed (any backend) is used
    ONLY to ship rank 0's 128-byte unique id; a non-Python host ships it by its own means."""
    from . import corpus as tc
    if not dist.is_initialized():          # a single process: a one-rank communicator, same data path
        return tc.Comm(tc.Comm.unique_id(), 1, 0, device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [tc.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                               group=group)
    return tc.Comm(box[0], world, rank, device)


class RcclShardedMatcher:
    """The sharded match with the collective BEHIND the C ABI (tvz_match_sharded): local sweep +
    per-shard top-k -> ncclAllGather -> merge are enqueued on one HIP stream by one library call.
    submit() alternates between two side streams with their own workspaces, so the all-gather and
    merge of batch i overlap the sweep of batch i+1; finish() makes the current stream wait for
    the ticket.  Same results as ShardedMatcher (which keeps the merge logic testable on gloo)."""

    def __init__(self, corpus, comm, k: int = 64, cap: int = 1024, n_streams: int = 2, priority: int = 0,
                 algo: int = 0):
        """`priority=-1`: the service's ticks - a few tiny launches that should not queue behind the upload
        workers' scene kernels."""
        self.corpus, self.comm = corpus, comm
        from . import _lib
        self.algo = int(algo)               # _lib.ALGO_* (+ ALGO_PAIR / ALGO_NO_PAIR) of every batch
        if not self.algo & (_lib.ALGO_PAIR | _lib.ALGO_NO_PAIR) and n_streams < 3:
            # two queries per lookup block pay off when a THIRD batch's blocks fill the longer tail of a launch
            # (42 against 46 us per batch on a 1/8 shard); with two batches in flight they cost (57 against 52)
            self.algo |= _lib.ALGO_NO_PAIR
        if not self.algo & (_lib.ALGO_WAVE | _lib.ALGO_NO_WAVE | _lib.ALGO_PAIR):
            # a stream of batches: on a shard of one sub-index the lookup that gives every query to one wave executes
            # a third fewer instructions per batch than the block kernel (which answers a LONE batch sooner: tvz.h)
            self.algo |= _lib.ALGO_PREFER_WAVE
        self.k = int(k)
        self.cap = max(int(cap), self.k)
        self.world, self.rank = comm.n_ranks, comm.rank
        self.collective = True
        self.dev = torch.device("cuda", corpus.device)
        self.streams = [torch.cuda.Stream(self.dev, priority=priority) for _ in range(n_streams)]
        self.ws = [None] * n_streams
        self.out = [None] * n_streams
        self.events = [torch.cuda.Event() for _ in range(n_streams)]
        self._i = 0
        self._plans = {}                    # (slot, query tensors, shape) -> the library call's arguments (submit)
        self._lib = _lib
        self._call = comm.lib.tvz_match_sharded

    def submit(self, d_queries: torch.Tensor, d_q_offsets: torch.Tensor, max_query_len: int,
               min_match: int, d_exclude_ids: Optional[torch.Tensor] = None, inputs_ready: bool = False):
        """Enqueue one batch; the returned ticket's tensors belong to this matcher and are
        overwritten by the submit() `n_streams` calls later - consume them (or copy) before that.
        Nothing is allocated per batch: at 8 GPUs a batch is ~0.1 ms of device time, and fresh
        output tensors + events per call were a comparable amount of host time.
        `inputs_ready`: the query tensors are complete already and nothing pending on the caller's
        stream still reads this slot's previous results, so the side stream does not wait for the
        caller's stream.  That wait
        is a marker in the caller's queue BEHIND the caller's earlier finish() waits - a chain of
        cross-queue barriers the command processor resolves in 10-40 us, which at 8 GPUs (a batch
        is ~60 us of kernels) left the chip idle between two batches' match kernels
        (profiles/r3_shard_pipeline.txt)."""
        from . import corpus as tc
        i = self._i
        self._i = (i + 1) % len(self.streams)
        st = self.streams[i]
        # Steady state: the same query tensors come round again (a service's staging slots, a benchmark's rotating
        # batches).  Everything that does not change with them - workspace, outputs, the seventeen arguments of the
        # library call - is kept as a PLAN per (slot, tensors, shape): a batch on a 1/8 shard is ~40 us of GPU time and
        # the un-planned submit was 30 us of interpreter time, the pipeline's actual bound.
        key = (i, d_queries.data_ptr(), d_q_offsets.data_ptr(), int(max_query_len), int(min_match),
               d_exclude_ids.data_ptr() if d_exclude_ids is not None else 0, d_queries.numel(), d_q_offsets.numel())
        plan = self._plans.get(key)
        if plan is None:
            Q = d_q_offsets.numel() - 1
            self.corpus._check_queries(d_queries, d_q_offsets)
            need = tc.workspace_bytes(Q, max_query_len, self.cap, self.k, self.world, d_queries.numel())
            if self.ws[i] is None or self.ws[i].numel() < need:
                self.ws[i] = torch.empty(need, dtype=torch.uint8, device=self.dev)
                self._plans = {k_: v for k_, v in self._plans.items() if k_[0] != i}     # (their workspace is gone)
            if self.out[i] is None or self.out[i][0].shape[0] != Q:
                self.out[i] = (torch.empty((Q, self.k, 3), dtype=torch.int32, device=self.dev),
                               torch.empty(Q, dtype=torch.int32, device=self.dev))
                self._plans = {k_: v for k_, v in self._plans.items() if k_[0] != i}
            merged, totals = self.out[i]
            args = (self.corpus._h, self.comm._h, key[1], key[2], Q, key[3], key[4], key[5] or None, self.cap, self.k,
                    merged.data_ptr(), totals.data_ptr(), self.ws[i].data_ptr(), self.ws[i].numel(), self.algo,
                    st.cuda_stream)
            if len(self._plans) > 256:
                self._plans.clear()
            plan = self._plans[key] = (args, merged, totals, self.ws[i])
        args, merged, totals, _ws = plan
        if not inputs_ready:
            st.wait_stream(torch.cuda.current_stream(self.dev))   # the queries are complete before the match reads them
        rc = self._call(*args)
        if rc:
            self._lib.check(rc)
        ev = self.events[i]
        ev.record(st)
        return (merged, totals, ev)

    def finish(self, ticket, host: bool = False):
        """Order the consumer behind the batch: the caller's current stream waits for it (default),
        or - `host=True`, what a consumer that reads the verdicts on the host does anyway - the
        calling thread does (no barrier packet in the caller's queue)."""
        merged, totals, ev = ticket
        if host:
            ev.synchronize()
        else:
            torch.cuda.current_stream(self.dev).wait_event(ev)
        return merged, totals

    def match_topk(self, d_queries, d_q_offsets, max_query_len, min_match, d_exclude_ids=None):
        return self.finish(self.submit(d_queries, d_q_offsets, max_query_len, min_match, d_exclude_ids))


def verdicts_from_topk(merged: np.ndarray):
    """Per query: (k*, sorted dup ids) = rows whose kth is minimal (app.py:235-255 batch form),
    or (None, []) when nothing reached min_match.  `truncated` is True when the tie set may
    exceed k (caller should re-query with a larger k)."""
    out = []
    for rows in merged:
        valid = rows[rows[:, 0] >= 0]
        valid = valid[valid[:, 2] < KTH_NEVER]
        if valid.shape[0] == 0:
            out.append((None, [], False))
            continue
        kstar = int(valid[:, 2].min())
        ids = sorted(int(v) for v in valid[valid[:, 2] == kstar][:, 0])
        truncated = bool(rows[-1, 0] >= 0 and rows[-1, 2] == kstar)
        out.append((kstar, ids, truncated))
    return out
