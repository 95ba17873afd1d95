"""BASELINE.json configs[4] as written: the inspector service over a corpus SHARDED across the GPUs
of one node - "GPU scene-cut + sharded corpus match, duplicate verdicts streamed".

The reference is one Python process scanning one Postgres table (inspector/app.py:31-44 notify ->
worker, :234-255 persist / match / first-hit stop; inspector/db.py:83 reads every row per call).
Here the rows live in R shards; what the driver (inspector.Inspector) sees is unchanged - an object
with DeviceCorpus's call shapes - so the per-upload loop, the records and the routes are exactly
those of the one-GPU service:

  * a row lives in the shard of whoever ingested it (the initial table: video_id mod R), so an
    add_timestamps (app.py:234) never leaves its GPU;
  * queries do: every micro-batch of every upload asks "which rows share >= min_match cuts with my
    list, and on which prefix first" (app.py:235-238).  The asks of all concurrent uploads are
    collected for a TICK and answered by ONE batched sharded match - per shard the index lookup
    over the whole batch, per-shard top-k, (between ranks: one all-gather of the [Q, k+1, 3]
    blocks,) the merge - the shape the lookup kernel is best at.  An upload's verdict is read off
    its merged top-k (rows with the minimal kth); a tie set that may continue past k is re-asked
    exactly.

Two forms of the same composition:
  ShardedCorpus  R shard handles in ONE process / on one GPU: what this environment can run and
                 test (tests/test_config4_gpu.py runs the 64-upload scenario over 8 shards).
  RankCorpus     one process per GPU: this rank's shard + the tick exchange with the other ranks
                 through torch.distributed (RCCL on the GPUs, gloo in the CPU test) around
                 sharded.ShardedMatcher / RcclShardedMatcher.  UNMEASURED ON HARDWARE: no box here
                 has more than one GPU; tests/test_service_cpu.py runs it at world size 2 on gloo.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from concurrent.futures import Future
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

KTH_NEVER = 0x7FFFFFFF
MAX_BATCH_LEN = 4095           # include/tvz.h: batched calls take queries of up to 4095 timestamps


def _hits_from_topk(rows: np.ndarray, total: int):
    """merged top-k rows [k,3] + total -> ([(video_id, count, kth)], exact?).  Not exact when a
    shard's hit list overflowed (total < 0) or the set of rows with the minimal kth may continue
    past k (the driver reports exactly that set: app.py:238-245)."""
    valid = rows[rows[:, 0] >= 0]
    hits = [(int(v), int(c), int(k)) for v, c, k in valid]
    if total < 0:
        return hits, False
    live = [h for h in hits if h[2] < KTH_NEVER]
    if live and len(valid) == rows.shape[0] and abs(total) > rows.shape[0]:
        kstar = min(h[2] for h in live)
        if hits[-1][2] == kstar:
            return hits, False
    return hits, True


class _Stage:
    """A tick's inputs in ONE pinned block and ONE host-to-device copy: [keys f64 | offsets i64 | exclude
    ids i32] (three small copies and a fresh pinned allocation each were most of a tick's host time at 64
    asks, and every one of them is a point where the tick thread gives up the interpreter lock).  The
    block is reused: a tick ends with a device-to-host copy of its results, so the next one starts
    behind it on the same stream."""

    def __init__(self, dev):
        self.dev = torch.device(dev)
        self._pin = None
        self._dev = None

    def put(self, queries, excl):
        """queries: a list of float64 arrays."""
        lens = np.fromiter((len(q) for q in queries), dtype=np.int64, count=len(queries))
        return self.put_flat(np.concatenate(queries) if int(lens.sum()) else np.zeros(0), lens, excl)

    def put_flat(self, flat, lens, excl):
        """flat: all keys back to back (float64[sum(lens)]); -> (d_keys, d_offsets, d_exclude, max_len)."""
        Q = len(lens)
        total = int(lens.sum())
        nk = max(total, 1)
        need = 8 * nk + 8 * (Q + 1) + 4 * Q
        if self._pin is None or self._pin.numel() < need:
            self._pin = torch.empty(max(need, 1 << 16), dtype=torch.uint8).pin_memory()
            self._dev = torch.empty(self._pin.numel(), dtype=torch.uint8, device=self.dev)
        h = self._pin.numpy()
        keys = h[:8 * nk].view(np.float64)
        offs = h[8 * nk:8 * nk + 8 * (Q + 1)].view(np.int64)
        ex = h[8 * nk + 8 * (Q + 1):need].view(np.int32)
        offs[0] = 0
        np.cumsum(lens, out=offs[1:])
        keys[:total] = flat
        ex[:] = excl
        d = self._dev
        d[:need].copy_(self._pin[:need], non_blocking=True)
        return (d[:8 * nk].view(torch.float64), d[8 * nk:8 * nk + 8 * (Q + 1)].view(torch.int64),
                d[8 * nk + 8 * (Q + 1):need].view(torch.int32), int(lens.max()) if Q else 0)


class TickBatcher:
    """Collects find_duplicates asks of concurrent uploads and answers them in batches: one
    `run_batch(items)` per tick, items = [(timestamps float64[], min_match, exclude_id)] sharing one
    min_match, -> [(rows int32[k,3], total)].  A tick starts as soon as something is pending and the
    previous tick is done, so a lone upload pays one batch latency and 64 uploads share it."""

    def __init__(self, run_batch: Callable, max_batch: int = 1024, linger_s: float = 0.0):
        self.run_batch = run_batch
        self.max_batch = int(max_batch)
        self.linger_s = float(linger_s)
        self._cv = threading.Condition()
        self._pending: List[tuple] = []
        self._stop = False
        self.ticks = 0
        self.asks = 0
        self._thread = threading.Thread(target=self._loop, name="tvz-tick", daemon=True)
        self._thread.start()

    def submit(self, timestamps, min_match: int, exclude_id: int) -> Future:
        fut: Future = Future()
        q = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64))
        with self._cv:
            if self._stop:
                raise RuntimeError("tick batcher is closed")
            self._pending.append((q, int(min_match), int(exclude_id), fut))
            self._cv.notify()
        return fut

    def _loop(self):
        while True:
            with self._cv:
                while not self._pending and not self._stop:
                    self._cv.wait(timeout=0.5)
                if self._stop and not self._pending:
                    return
            if self.linger_s > 0:
                time.sleep(self.linger_s)                   # let concurrent uploads join this tick
            with self._cv:
                mm = self._pending[0][1]
                take = [p for p in self._pending if p[1] == mm][: self.max_batch]
                ids = {id(p) for p in take}
                self._pending = [p for p in self._pending if id(p) not in ids]
            try:
                res = self.run_batch([(q, m, e) for q, m, e, _ in take])
                for (_, _, _, fut), r in zip(take, res):
                    fut.set_result(r)
            except BaseException as e:                      # every waiting upload sees the failure
                for _, _, _, fut in take:
                    if not fut.done():
                        fut.set_exception(e)
            self.ticks += 1
            self.asks += len(take)

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=10)


class ShardedCorpus:
    """DeviceCorpus's call shapes over R shard handles on one GPU.  db.Store(url, corpus=this) +
    inspector.Inspector(store) is the sharded service in one process."""

    def __init__(self, device: int = 0, n_shards: int = 8, k: int = 64, cap: int = 4096, linger_s: float = 0.0):
        from . import corpus as tc
        self._tc = tc
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        self.R = int(n_shards)
        self.k, self.cap = int(k), max(int(cap), int(k))
        self.shards = [tc.DeviceCorpus(self.device) for _ in range(self.R)]
        # the tick thread's own stream, at high priority: a tick is a few tiny kernels and two tiny copies
        # that otherwise queue behind the upload workers' scene kernels
        self.stream = torch.cuda.Stream(self.dev, priority=-1)
        self._owner = {}
        self._lock = threading.Lock()
        self._ws = [None] * self.R
        self._stager = _Stage(self.dev)
        self.exact_asks = 0
        self.tick_host_s = 0.0              # wall time inside _run_batch (host work + the wait for the GPU)
        self.batcher = TickBatcher(self._run_batch, linger_s=linger_s)

    # ---- rows: a video's row stays in the shard of whoever ingested it ----
    def _shard_of(self, video_id: int) -> int:
        with self._lock:
            return self._owner.setdefault(int(video_id), int(video_id) % self.R)

    def upload(self, rows) -> None:
        rows = [(int(v), list(t)) for v, t in rows]
        parts = [[] for _ in range(self.R)]
        with self._lock:
            self._owner.clear()
        for v, t in rows:
            parts[self._shard_of(v)].append((v, t))          # duplicate ids (no UNIQUE constraint) stay together
        for s, p in zip(self.shards, parts):
            s.upload(p)

    def upsert(self, video_id: int, timestamps) -> None:
        self.shards[self._shard_of(video_id)].upsert(int(video_id), timestamps)

    def clear(self) -> None:
        for s in self.shards:
            s.clear()
        with self._lock:
            self._owner.clear()

    def stats(self):
        st = [s.stats() for s in self.shards]
        return tuple(sum(x[i] for x in st) for i in range(3))

    def index_stats(self):
        st = [s.index_stats() for s in self.shards]
        return {k: sum(x[k] for x in st) for k in st[0]}

    def close(self) -> None:
        self.batcher.close()
        for s in self.shards:
            s.close()

    # ---- matches ----
    def _exact(self, q, min_match, exclude_id, with_kth):
        out = []
        for s in self.shards:
            out += s.find_duplicates(q, min_match, exclude_id=exclude_id, with_kth=True)
        out.sort()
        return out if with_kth else [(v, c) for v, c, _ in out]

    def find_duplicates(self, new_timestamps: Sequence[float], min_match: int = 5, exclude_id: int = -1,
                        with_kth: bool = False):
        """with_kth (the driver's per-micro-batch ask): through the tick -> the k best rows by
        (kth, video_id), which hold the verdict; otherwise (db.find_duplicates, db.py:76-94: every
        row with its count) the shards are asked one by one."""
        q = np.ascontiguousarray(np.asarray(new_timestamps, dtype=np.float64))
        if not with_kth or q.size > MAX_BATCH_LEN or not 1 <= int(min_match) <= 5:
            return self._exact(q, min_match, exclude_id, with_kth)
        rows, total = self.batcher.submit(q, min_match, exclude_id).result()
        hits, exact = _hits_from_topk(rows, total)
        if exact:
            return hits
        self.exact_asks += 1
        return self._exact(q, min_match, exclude_id, True)

    def _stage(self, queries, excl):
        return self._stager.put(queries, excl)

    def _run_batch(self, items):
        """One tick: every shard's lookup (top-k kept in the lookup's epilogue) over the whole batch, the
        blocks written where the merge reads them ([R, Q, k+1, 3], as an all-gather would deliver them),
        the merge, ONE device-to-host copy.  All on the tick thread's stream, back to back: the shards
        share one device, and a cross-stream wait per shard (12-30 us each on this system before the
        waiting queue moves, profiles/r3_shard_pipeline.txt) cost more than the overlap gave."""
        t0 = time.perf_counter()
        tc = self._tc
        mm = items[0][1]
        Q = len(items)
        with torch.cuda.device(self.dev), torch.cuda.stream(self.stream):
            d_q, d_off, d_ex, max_len = self._stage([q for q, _, _ in items], [e for _, _, e in items])
            need = tc.workspace_bytes(Q, max_len, self.cap, self.k, total_query_keys=d_q.numel())
            if self._ws[0] is None or self._ws[0].numel() < need:
                self._ws[0] = torch.empty(need, dtype=torch.uint8, device=self.dev)       # one workspace: the shards run in turn
            # ONE library call for the R lookups + the merge (tvz_match_topk_shards): every return to the
            # interpreter is a chance to wait for its lock behind 16 upload threads busy in the ORM
            _, merged, totals = tc.match_topk_shards(self.shards, d_q, d_off, max_len, mm, self.cap, self.k,
                                                     self._ws[0], d_exclude_ids=d_ex)
            both = torch.cat([merged.reshape(Q, self.k * 3), totals.reshape(Q, 1)], dim=1).cpu().numpy()
        self.tick_host_s += time.perf_counter() - t0
        k3 = self.k * 3
        return [(both[i, :k3].reshape(self.k, 3), int(both[i, k3])) for i in range(Q)]


ASK_TOPK, ASK_EXACT = 0, 1


class RankCorpus:
    """One process per GPU: this rank's shard + the tick exchange.  Every rank runs the same loop
    at the same cadence (the exchange is collective): gather how many asks each rank has, gather the
    padded asks (ONE collective: keys and the per-ask fields travel in one float64 block), answer ALL
    of them against the own shard and keep the answers to the own asks.  Two kinds of ask:

      ASK_TOPK   the driver's per-micro-batch question (app.py:235-238): through `matcher`
                 (sharded.ShardedMatcher or RcclShardedMatcher: local match -> per-shard top-k -> one
                 all-gather -> merge, identical on every rank); the k best rows by (kth, video_id) hold
                 the verdict unless the rows sharing the earliest prefix may continue past k;
      ASK_EXACT  every row that reaches min_match, as db.find_duplicates returns them (db.py:85-91):
                 each rank asks its own shard (`shard.find_duplicates`, any query length, any min_match),
                 the per-rank hit counts are all-gathered, then the variable-length hit lists (padded to
                 the longest).  Taken by db.find_duplicates-shaped calls whose answer does not fit k, by a
                 top-k ask whose tie set may exceed k (the reference reports ALL rows of the earliest
                 prefix: an upload never fails because of k), by queries of more than 4095 timestamps
                 and by min_match outside 1..5.

    `shard`: this rank's DeviceCorpus (or a stand-in with upload/upsert/clear/find_duplicates);
    `matcher.match_topk(d_q, d_off, max_len, min_match, d_excl) -> (merged [Q,k,3], totals [Q])`;
    `xdev`: where the exchanged tensors live ("cpu" for gloo, the GPU for RCCL);
    `group`: a process group used by NOTHING else (the tick thread issues collectives on it
    concurrently with whatever the other threads of the process do on theirs);
    `owner_fn(video_id) -> rank` decides which rank keeps a row of the initial table (default
    video_id mod world; the N-rank service routes by file name and passes its own).

    Failures.  What a rank computes ALONE inside a tick - `shard.find_duplicates` for the exact asks - may fail
    without stranding the others: the failing rank sends a count of -1 into the tick's next collective, every
    rank reads the same counts and raises in the SAME tick: all pending and later asks fail with the error on
    every rank, nobody waits in a collective and nobody is answered from an incomplete set of shards
    (tests/test_service_cpu.py, a shard that raises on one rank).  A failure INSIDE a collective step (the matcher's all-gather, an exchange that errors) cannot be
    announced - the other ranks are in that collective - and the failing rank's loop ends at once: it marks
    the corpus BROKEN (`broken`; every pending and later ask raises it) and only the END OF ITS PROCESS frees the
    others (the service's child watchdog exits with code 3 on `broken`; a host that embeds RankCorpus elsewhere,
    e.g. bench.py's ranked e2e leg, must do the same)."""

    def __init__(self, shard, matcher, group=None, xdev="cpu", tick_s: float = 0.0005, max_batch: int = 1024,
                 owner_fn: Optional[Callable[[int], int]] = None, idle_tick_s: float = 0.005, idle_after: int = 400):
        self.shard, self.matcher, self.group = shard, matcher, group
        self.xdev = torch.device(xdev)
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.owner_fn = owner_fn
        self.tick_s, self.max_batch = float(tick_s), int(max_batch)
        # a service nobody uploads to still exchanges "nothing to do" at every cadence point: after `idle_after`
        # empty ticks in a row (the same count on every rank: it is read off the exchanged meta) the cadence
        # drops to `idle_tick_s`; an ask of this rank still wakes its loop at once, the siblings follow within
        # one idle tick
        self.idle_tick_s, self.idle_after = max(float(idle_tick_s), float(tick_s)), int(idle_after)
        self._cv = threading.Condition()
        self._pending: List[tuple] = []
        self._stop = False
        self.broken: Optional[BaseException] = None
        self._local_error: Optional[BaseException] = None   # this rank's own work failed inside a tick (announced to the others)
        self.ticks = 0
        self.busy_ticks = 0
        self.exact_asks = 0
        self._stager = None                 # pinned staging of the device matcher's inputs (created on first use)
        self.tick_host_s = 0.0              # wall time of the busy ticks (profiles/e2e_service.py ... -1)
        self._thread = threading.Thread(target=self._loop, name="tvz-rank-tick", daemon=True)
        self._thread.start()

    # ---- rows ----
    def upload(self, rows) -> None:
        """The initial table: this rank keeps the rows `owner_fn` gives it (video_id mod world)."""
        own = self.owner_fn or (lambda v: int(v) % self.world)
        self.shard.upload([(int(v), list(t)) for v, t in rows if own(int(v)) % self.world == self.rank])

    def upsert(self, video_id: int, timestamps) -> None:
        self.shard.upsert(int(video_id), timestamps)         # ingested here: lives here

    def clear(self) -> None:
        self.shard.clear()

    def stats(self):
        return self.shard.stats()

    # ---- matches ----
    def _ask(self, q, min_match, exclude_id, kind):
        fut: Future = Future()
        with self._cv:
            if self.broken is not None:
                raise RuntimeError(f"rank corpus is broken: {self.broken!r}")
            if self._stop:
                raise RuntimeError("rank corpus is closed")
            self._pending.append((q, int(min_match), int(exclude_id), int(kind), fut))
            self._cv.notify()
        return fut.result()

    def find_duplicates(self, new_timestamps, min_match: int = 5, exclude_id: int = -1, with_kth: bool = False):
        """with_kth (the driver): the rows that decide the verdict - the merged top-k when it is
        conclusive, every matching row otherwise.  Without (db.find_duplicates, db.py:76-94): every
        matching row with its count, sorted by video_id."""
        q = np.ascontiguousarray(np.asarray(new_timestamps, dtype=np.float64))
        if q.size <= MAX_BATCH_LEN and 1 <= int(min_match) <= 5:
            rows, total = self._ask(q, min_match, exclude_id, ASK_TOPK)
            hits, exact = _hits_from_topk(rows, total)
            if with_kth and exact:
                return hits
            if not with_kth and 0 <= total <= len(hits):          # the k best ARE all of them
                return sorted((v, c) for v, c, _ in hits)
        self.exact_asks += 1
        hits = self._ask(q, min_match, exclude_id, ASK_EXACT)
        return hits if with_kth else [(v, c) for v, c, _ in hits]

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        """[...] -> [world, ...] (dim-0 concatenation is the layout RCCL and gloo both accept)."""
        if self.world == 1:
            return t.unsqueeze(0)
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def _fail(self, err: BaseException, take=()) -> None:
        with self._cv:
            if self.broken is None:
                self.broken = err
            pend, self._pending = self._pending, []
        for p in list(take) + pend:
            if not p[4].done():
                p[4].set_exception(RuntimeError(f"rank corpus is broken: {err!r}"))

    def _loop(self):
        take = []
        idle = 0
        try:
            while True:
                with self._cv:
                    stop = self._stop
                    take = self._pending[: self.max_batch]
                    self._pending = self._pending[len(take):]
                # 1) who has how much; does anybody want to stop (all ranks leave together); is anybody broken
                meta = torch.tensor([len(take), max((len(p[0]) for p in take), default=0), 1 if stop else 0,
                                     1 if self._local_error is not None else 0], dtype=torch.int64, device=self.xdev)
                allmeta = self._gather(meta).cpu().numpy()
                self.ticks += 1
                if int(allmeta[:, 3].max()) == 1:            # the same tick on every rank: they stop together
                    bad = [int(r) for r in np.flatnonzero(allmeta[:, 3])]
                    raise RuntimeError(f"rank(s) {bad} failed in the last tick: the sharded service stops"
                                       + (f" ({self._local_error!r})" if self._local_error is not None else ""))
                if int(allmeta[:, 0].sum()) == 0:
                    if int(allmeta[:, 2].min()) == 1:        # every rank is closing and nothing is pending anywhere
                        return
                    idle += 1
                    with self._cv:                           # idle: until the next cadence point, or an ask of this rank
                        if not self._pending and not self._stop:
                            self._cv.wait(timeout=self.tick_s if idle < self.idle_after else self.idle_tick_s)
                    continue
                idle = 0
                self.busy_ticks += 1
                t0 = time.perf_counter()
                self._exchange_and_answer(take, allmeta)
                self.tick_host_s += time.perf_counter() - t0
                take = []
        except BaseException as e:                           # noqa: BLE001 - every waiting upload must see it
            self._fail(e, take)

    def _exchange_and_answer(self, take, allmeta):
        Qcap, Lcap = int(allmeta[:, 0].max()), max(int(allmeta[:, 1].max()), 1)
        # 2) the asks, padded to the largest rank's block, as ONE float64 block per rank:
        #    [Qcap, 4 + Lcap] = (len, exclude, min_match, kind | keys...); small integers are exact in float64
        blk = np.zeros((Qcap, 4 + Lcap), dtype=np.float64)
        for i, (q, mm, ex, kind, _) in enumerate(take):
            blk[i, 0], blk[i, 1], blk[i, 2], blk[i, 3] = len(q), ex, mm, kind
            blk[i, 4:4 + len(q)] = q
        g = self._gather(torch.from_numpy(blk).to(self.xdev)).cpu().numpy()       # [world, Qcap, 4 + Lcap]
        g_info = g[:, :, :4].astype(np.int64)
        g_keys = g[:, :, 4:]
        # the global batch, in the same order on every rank: rank-major; rows past a rank's count are padding
        valid = np.arange(Qcap)[None, :] < allmeta[:, 0][:, None]                  # [world, Qcap]
        rr, ii = np.nonzero(valid)
        lens_all, excl_all, mm_all, kind_all = (g_info[rr, ii, c] for c in range(4))
        mine = rr == self.rank
        mdev = getattr(self.matcher, "dev", torch.device("cpu"))
        # 3) top-k asks, one batched sharded match per min_match (vectorised un-padding: no per-ask Python)
        for mm in sorted(set(int(m) for m in mm_all[kind_all == ASK_TOPK])):
            sel = np.flatnonzero((kind_all == ASK_TOPK) & (mm_all == mm))
            lens = lens_all[sel]
            offs = np.zeros(len(sel) + 1, dtype=np.int64)
            np.cumsum(lens, out=offs[1:])
            keys2d = g_keys[rr[sel], ii[sel]]                                      # [n, Lcap]
            flat = keys2d[np.arange(Lcap)[None, :] < lens[:, None]]                # row-major: ask after ask
            if mdev.type == "cuda":
                # one pinned block, one host-to-device copy for keys + offsets + exclusions; the tick's own
                # high-priority stream (its tiny kernels and copies otherwise queue behind the scene kernels)
                if self._stager is None:
                    self._stager = _Stage(mdev)
                    self._stream = torch.cuda.Stream(mdev, priority=-1)
                with torch.cuda.device(mdev), torch.cuda.stream(self._stream):
                    d_q, d_off, d_ex, _ = self._stager.put_flat(flat, lens, excl_all[sel].astype(np.int32))
                    merged, totals = self.matcher.match_topk(d_q, d_off, int(lens.max()) if len(lens) else 0, mm, d_ex)
                    Qn, k = merged.shape[0], merged.shape[1]
                    # ONE device-to-host copy per batch: rows and totals together
                    both = torch.cat([merged.reshape(Qn, k * 3), totals.reshape(Qn, 1)], dim=1).cpu().numpy()
            else:
                if flat.size == 0:
                    flat = np.zeros(1)
                d_q = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float64))
                d_off = torch.from_numpy(offs)
                d_ex = torch.from_numpy(excl_all[sel].astype(np.int32))
                merged, totals = self.matcher.match_topk(d_q, d_off, int(lens.max()) if len(lens) else 0, mm, d_ex)
                Qn, k = merged.shape[0], merged.shape[1]
                both = torch.cat([merged.reshape(Qn, k * 3), totals.reshape(Qn, 1).to(merged.dtype)], dim=1).numpy()
            for j in np.flatnonzero(mine[sel]):
                a = int(sel[j])
                take[int(ii[a])][4].set_result((both[j, :k * 3].reshape(k, 3).copy(), int(both[j, k * 3])))
        # 4) exact asks: every rank asks its own shard, counts and padded hit lists are all-gathered
        sel = np.flatnonzero(kind_all == ASK_EXACT)
        if len(sel):
            local = []
            for a in sel:
                q = g_keys[rr[a], ii[a], :int(lens_all[a])]
                try:
                    if self._local_error is not None:
                        raise self._local_error
                    local.append(self.shard.find_duplicates(q, int(mm_all[a]), exclude_id=int(excl_all[a]), with_kth=True))
                except Exception as e:                       # noqa: BLE001 - this rank's own work: finish the tick's collectives
                    self._local_error = e                    # with a poisoned count (below): every rank stops in this tick
                    local.append([])
            # (a rank whose own shard failed sends -1: every rank reads the same counts and stops in THIS tick, before
            # anybody's ask is answered from an incomplete set of shards)
            counts = torch.tensor([-1 if self._local_error is not None else len(h) for h in local], dtype=torch.int64,
                                  device=self.xdev)
            allcounts = self._gather(counts).cpu().numpy()                         # [world, n]
            if (allcounts < 0).any():
                bad = [int(r) for r in np.flatnonzero((allcounts < 0).any(axis=1))]
                raise RuntimeError(f"the shard of rank(s) {bad} failed: the sharded service stops"
                                   + (f" ({self._local_error!r})" if self._local_error is not None else ""))
            width = int(allcounts.max())
            if width:
                pad = np.full((len(sel), width, 3), -1, dtype=np.int32)
                for j, h in enumerate(local):
                    if h:
                        pad[j, :len(h)] = np.asarray(h, dtype=np.int32)
                allhits = self._gather(torch.from_numpy(pad).to(self.xdev)).cpu().numpy()   # [world, n, width, 3]
            for j in np.flatnonzero(mine[sel]):
                a = int(sel[j])
                rows = [allhits[r, j, :int(allcounts[r, j])] for r in range(self.world) if allcounts[r, j]] if width else []
                hits = sorted((int(v), int(c), int(k)) for v, c, k in np.concatenate(rows)) if rows else []
                take[int(ii[a])][4].set_result(hits)

    def close(self) -> None:
        """Collective: every rank calls it; the loops leave together once nothing is pending."""
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=60)
        if self._thread.is_alive():
            # still exchanging (another rank has not closed): closing the shard under it would free
            # memory its matches read
            raise RuntimeError("rank corpus: the tick thread did not stop (close() is collective)")
        if hasattr(self.shard, "close"):
            self.shard.close()


# ==================================================================================================
# configs[4] as a COMMAND:  python -m tvidz_amd.service --ranks N [--port 5000] [--db URL]
#
# A parent that never touches a GPU spawns one fresh child per device (a child is started as a new
# program: nothing that has initialised the GPU is ever exec'ed over) and serves a thin FRONT on
# --port; each child is the whole one-GPU service - db.Store + inspector.Inspector + the Flask routes
# of the reference (inspector/app.py:31-115) - over RankCorpus(RcclShardedMatcher): its own shard of
# the table, the tick exchange with its siblings.  The front keeps the reference's surface:
#   POST /notify                 -> the rank that owns the upload (crc32 of the clean file name mod N:
#                                   the key the reference stores in `videos.filename`, app.py:122-150),
#                                   so an upload's add_timestamps never leave its GPU (app.py:234);
#   GET  /status/<f>             -> the owning rank's record (the others are asked too when it has none:
#                                   the reference also accepts an analysis key here, app.py:56);
#   GET  /status/stream/<f>      -> the owning rank's SSE stream, relayed chunk by chunk (app.py:64-115).
# A child that dies (or whose tick loop broke) takes the service down with a non-zero exit: the other
# ranks would wait for it in the next collective.
# UNMEASURED ON HARDWARE for N > 1 (no box here has two GPUs): tests/test_service_launch_cpu.py drives
# two ranks on gloo through the front's HTTP surface, tests/test_service_gpu.py one rank on the GPU.
# ==================================================================================================
def owner_rank(clean_filename: str, world: int) -> int:
    import zlib
    return zlib.crc32(str(clean_filename).encode("utf-8", "surrogatepass")) % max(int(world), 1)


def clean_name(name_or_key: str) -> str:
    """The name `videos.filename` holds for an upload (inspector.split_filenames, app.py:122-130)."""
    from .inspector import split_filenames
    return split_filenames(name_or_key)[1]


class FilenameOwner:
    """owner_fn for RankCorpus.upload: a row of the initial table belongs to the rank its video's file
    name routes to - the rank that would have ingested it."""

    def __init__(self, url: str, world: int):
        self.url, self.world = url, int(world)
        self.names: dict = {}

    def __call__(self, video_id: int) -> int:
        if int(video_id) not in self.names:
            from . import db
            self.names = db.video_filenames(self.url)           # (one SELECT per reload, not per row)
            self.names.setdefault(int(video_id), None)          # timestamps of a video `videos` does not hold: asked once
        name = self.names.get(int(video_id))
        return owner_rank(name, self.world) if name is not None else int(video_id) % self.world


def _load_hook(spec: str):
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


def _hip_parts(rank: int, world: int, group, a):
    """What a rank is made of on an MI355X: its DeviceCorpus, the RCCL matcher behind the C ABI, the
    real driver.  (`--parts module:function` swaps this for test doubles on a CPU box.)"""
    from . import corpus as tc, sharded
    from .inspector import Inspector
    shard = tc.DeviceCorpus(a.device)
    comm = sharded.make_comm(a.device)                          # rank 0's 128-byte id over the default group
    matcher = sharded.RcclShardedMatcher(shard, comm, k=a.k, cap=a.cap, priority=-1)
    # the asks are host data: they are exchanged on the host (`group`: gloo by default) - no H2D / D2H and no
    # stream synchronisation per tick just to learn who asks what; the device collective (one ncclAllGather
    # of the top-k blocks per batch) lives behind the C ABI in `comm`
    xdev = f"cuda:{a.device}" if a.backend == "nccl" else "cpu"
    return dict(shard=shard, matcher=matcher, xdev=xdev,
                inspector=lambda store: Inspector(store, device=f"cuda:{a.device}", max_workers=a.workers))


def parent_gone(parent_pid: int) -> bool:
    """True once this process's parent is no longer the process that started it (it was re-parented: to pid 1,
    or to a subreaper)."""
    return os.getppid() != parent_pid


def _child_main(a) -> int:
    """One rank: process group, shard, tick exchange, store, driver, routes."""
    from . import db
    from .inspector import create_app
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(a.master_port)
    if a.switch_interval > 0:
        # (an option, off by default: with 16 upload threads taking the interpreter lock in turn a shorter
        # switch interval was expected to shorten a tick's waits for it; measured, 0.5 ms against the default
        # 5 ms made no difference to a tick's ~2.8 ms of wall time or to the fps - gpurun_out/r4_e2e_e.txt)
        sys.setswitchinterval(a.switch_interval)
    if a.backend == "nccl":
        torch.cuda.set_device(a.device)
    dist.init_process_group(a.backend, rank=a.rank, world_size=a.ranks,
                            init_method=f"tcp://127.0.0.1:{a.master_port}")
    group = dist.new_group(backend=a.backend)                   # the tick thread's own (see RankCorpus)
    parts = (_load_hook(a.parts) if a.parts else _hip_parts)(a.rank, a.ranks, group, a)
    rc = RankCorpus(parts["shard"], parts["matcher"], group=group, xdev=parts["xdev"], tick_s=a.tick_s,
                    owner_fn=FilenameOwner(a.db, a.ranks))
    store = db.Store(a.db, corpus=rc, census=False)
    inspector = parts["inspector"](store)
    app = create_app(inspector)

    @app.route("/rank-info", methods=["GET"])
    def rank_info():                                            # what the front's readiness probe and the tests read
        from flask import jsonify
        rows, keys, _ = rc.stats()
        return jsonify({"rank": a.rank, "ranks": a.ranks, "pid": os.getpid(), "rows": rows, "keys": keys, "ticks": rc.ticks,
                        "busy_ticks": rc.busy_ticks, "exact_asks": rc.exact_asks,
                        "tick_host_s": rc.tick_host_s, "broken": repr(rc.broken) if rc.broken else None})

    # The launcher's pid as the launcher itself passed it (--parent-pid); a child started by hand takes its parent
    # at start-up.  NOT `os.getppid() == 1`: the reference runs its app as the container's PID 1
    # (inspector/entrypoint.sh ends with `exec python app.py`), so a launcher started the same way IS pid 1 and
    # every rank would leave at once; and under a subreaper an orphan is re-parented to a pid other than 1.
    parent_pid = a.parent_pid if a.parent_pid > 0 else os.getppid()

    def watchdog():
        while True:
            time.sleep(0.2)
            if rc.broken is not None:
                sys.stderr.write(f"[tvidz rank {a.rank}] tick loop broke: {rc.broken!r}\n")
                os._exit(3)
            if parent_gone(parent_pid):                         # the launcher is gone: so is the service
                os._exit(4)
    threading.Thread(target=watchdog, name="tvz-watchdog", daemon=True).start()
    app.run(host="127.0.0.1", port=a.http_port, threaded=True, use_reloader=False)
    return 0


def create_front(urls: List[str], timeout: float = 30.0):
    """The thin front: routes by file name, relays bodies untouched."""
    import requests
    from flask import Flask, Response, jsonify, request

    app = Flask("tvidz-front")
    N = len(urls)

    def cors(resp):                                             # app.py:15-19
        resp.headers["Access-Control-Allow-Origin"] = "*"
        resp.headers["Access-Control-Allow-Methods"] = "GET, POST, OPTIONS"
        resp.headers["Access-Control-Allow-Headers"] = "Content-Type"
        return resp

    app.after_request(cors)

    def relay(r):
        return Response(r.content, status=r.status_code, mimetype=r.headers.get("Content-Type", "application/json"))

    @app.route("/notify", methods=["POST"])                     # app.py:31-44
    def notify():
        data = request.get_json(silent=True)
        try:
            key = data["Records"][0]["s3"]["object"]["key"]
            data["Records"][0]["s3"]["bucket"]["name"]
        except Exception as e:
            return jsonify({"error": "Invalid event format", "details": str(e)}), 400
        r = owner_rank(clean_name(key), N)
        return relay(requests.post(f"{urls[r]}/notify", json=data, timeout=timeout))

    @app.route("/status/<filename>", methods=["GET"])           # app.py:46-62
    def status(filename):
        first = owner_rank(clean_name(filename), N)
        for r in [first] + [x for x in range(N) if x != first]:
            resp = requests.get(f"{urls[r]}/status/{filename}", timeout=timeout)
            try:
                pending = resp.json().get("status") == "pending"
            except Exception:
                pending = False
            if not pending:
                return relay(resp)
        return jsonify({"status": "pending"})

    @app.route("/status/stream/<filename>", methods=["OPTIONS"])    # app.py:23-25
    def status_stream_options(filename):
        return cors(Response())

    @app.route("/status/stream/<filename>")                     # app.py:64-115
    def status_stream(filename):
        r = owner_rank(clean_name(filename), N)
        up = requests.get(f"{urls[r]}/status/stream/{filename}", stream=True, timeout=(timeout, None))

        def pump():
            try:
                for chunk in up.iter_content(chunk_size=None):
                    if chunk:
                        yield chunk
            finally:
                up.close()
        return cors(Response(pump(), mimetype="text/event-stream"))

    @app.route("/admin/clear-db", methods=["POST"])             # app.py:325-333: every rank drops its shard
    def clear_db():
        for u in urls:
            requests.post(f"{u}/admin/clear-db", timeout=timeout)
        return jsonify({"status": "cleared"})

    @app.route("/ranks", methods=["GET"])
    def ranks():
        return jsonify({"ranks": [requests.get(f"{u}/rank-info", timeout=timeout).json() for u in urls]})

    @app.route("/<path:rest>", methods=["GET", "POST"])          # /build-info, /debug/*: rank 0 answers
    def other(rest):
        fn = requests.post if request.method == "POST" else requests.get
        kw = {"json": request.get_json(silent=True)} if request.method == "POST" else {}
        return relay(fn(f"{urls[0]}/{rest}", timeout=timeout, **kw))

    return app


class RankService:
    """The parent's side: spawn the rank processes, wait until they answer, watch them, stop them.
    Used by main() and by the tests (which talk to the front through Flask's test client or HTTP)."""

    def __init__(self, ranks: int, db_url: str, base_port: int = 5000, backend: str = "gloo", parts: str = "",
                 devices: Optional[List[int]] = None, k: int = 64, cap: int = 4096, workers: int = 16,
                 tick_s: float = 0.0005, env: Optional[dict] = None, ready_timeout: float = 300.0):
        import socket
        import subprocess
        from . import db
        self.ranks = int(ranks)
        db.create_schema(db_url)                                # once, before N processes open it side by side
        with socket.socket() as s:                              # a free rendezvous port
            s.bind(("127.0.0.1", 0))
            master_port = s.getsockname()[1]
        self.urls = [f"http://127.0.0.1:{base_port + 1 + r}" for r in range(self.ranks)]
        devices = devices or list(range(self.ranks))
        self.procs = []
        for r in range(self.ranks):
            cmd = [sys.executable, "-m", "tvidz_amd.service", "--child", "--rank", str(r), "--ranks", str(self.ranks),
                   "--master-port", str(master_port), "--http-port", str(base_port + 1 + r), "--db", db_url,
                   "--backend", backend, "--device", str(devices[r]), "--k", str(k), "--cap", str(cap),
                   "--workers", str(workers), "--tick-s", str(tick_s), "--parent-pid", str(os.getpid())] + \
                  (["--parts", parts] if parts else [])
            e = dict(os.environ)
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it here
            e.update(env or {})
            self.procs.append(subprocess.Popen(cmd, env=e))
        self._wait_ready(ready_timeout)

    def _wait_ready(self, timeout: float) -> None:
        import requests
        deadline = time.time() + timeout
        pending = set(range(self.ranks))
        while pending:
            dead = self.dead()
            if dead:
                self.stop()
                raise RuntimeError(f"rank process(es) {dead} exited during start-up")
            if time.time() > deadline:
                self.stop()
                raise RuntimeError(f"ranks {sorted(pending)} did not come up within {timeout:.0f} s")
            for r in list(pending):
                try:
                    if requests.get(f"{self.urls[r]}/rank-info", timeout=2).status_code == 200:
                        pending.discard(r)
                except Exception:
                    pass
            time.sleep(0.2)

    def dead(self) -> List[int]:
        return [r for r, p in enumerate(self.procs) if p.poll() is not None]

    def stop(self) -> None:
        for p in self.procs:
            if p.poll() is None:
                p.terminate()
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()


def main(argv=None) -> int:  # pragma: no cover - exercised through subprocesses by the tests
    import argparse
    ap = argparse.ArgumentParser(prog="python -m tvidz_amd.service",
                                 description="the inspector service over a corpus sharded across N GPUs (one process each)")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--port", type=int, default=5000, help="the front; rank r listens on port + 1 + r (loopback)")
    ap.add_argument("--db", default=os.environ.get("POSTGRES_URL", "postgresql://tvidz:tvidz@postgres:5432/tvidz"))
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"],
                    help="torch.distributed backend of the HOST-side exchange (who asks what); the device collective "
                         "is RCCL behind the C ABI either way")
    ap.add_argument("--switch-interval", type=float, default=0.0, help="sys.setswitchinterval of a rank process (0: leave)")
    ap.add_argument("--parts", default="", help="module:function building a rank's shard/matcher/driver (tests)")
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--cap", type=int, default=4096)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--tick-s", type=float, default=0.0005)
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--master-port", type=int, default=29500, help=argparse.SUPPRESS)
    ap.add_argument("--http-port", type=int, default=5001, help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--parent-pid", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.child:
        return _child_main(a)
    svc = RankService(a.ranks, a.db, base_port=a.port, backend=a.backend, parts=a.parts, k=a.k, cap=a.cap,
                      workers=a.workers, tick_s=a.tick_s)

    def monitor():
        while True:
            time.sleep(0.5)
            dead = svc.dead()
            if dead:
                sys.stderr.write(f"[tvidz] rank process(es) {dead} exited: stopping the service\n")
                svc.stop()
                os._exit(1)
    threading.Thread(target=monitor, name="tvz-monitor", daemon=True).start()
    try:
        create_front(svc.urls).run(host="0.0.0.0", port=a.port, threaded=True, use_reloader=False)
    finally:
        svc.stop()
    return 0


if __name__ == "__main__":  # pragma: no cover
    sys.exit(main())
