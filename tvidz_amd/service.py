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
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.R)]
        self._owner = {}
        self._lock = threading.Lock()
        self._ws = [None] * self.R
        self.exact_asks = 0
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

    def _run_batch(self, items):
        tc = self._tc
        queries = [q for q, _, _ in items]
        mm = items[0][1]
        Q = len(queries)
        cur = torch.cuda.current_stream(self.dev)
        d_q, d_off, max_len = tc.pack_queries(queries, self.dev)
        d_ex = torch.tensor([e for _, _, e in items], dtype=torch.int32, device=self.dev)
        need = tc.workspace_bytes(Q, max_len, self.cap, self.k)
        blocks = []
        for r, (shard, st) in enumerate(zip(self.shards, self.streams)):
            if self._ws[r] is None or self._ws[r].numel() < need:
                self._ws[r] = torch.empty(need, dtype=torch.uint8, device=self.dev)
            st.wait_stream(cur)
            blocks.append(shard.match_topk(d_q, d_off, max_len, mm, self.cap, self.k, d_exclude_ids=d_ex,
                                           stream=st, workspace=self._ws[r]))
        for st in self.streams:
            cur.wait_stream(st)
        merged, totals = tc.topk_merge(torch.stack(blocks).contiguous(), self.k)     # as the all-gather delivers them
        merged, totals = merged.cpu().numpy(), totals.cpu().numpy()
        return [(merged[i], int(totals[i])) for i in range(Q)]


class RankCorpus:
    """One process per GPU: this rank's shard + the tick exchange.  Every rank runs the same loop
    at the same cadence (the exchange is collective): gather how many asks each rank has, gather the
    padded asks, answer ALL of them against the own shard through `matcher` (sharded.ShardedMatcher
    or RcclShardedMatcher: local match -> per-shard top-k -> one all-gather -> merge, identical on
    every rank) and keep the answers to the own asks.

    `shard`: this rank's DeviceCorpus (or a stand-in with upload/upsert/clear/find_duplicates);
    `matcher.match_topk(d_q, d_off, max_len, min_match, d_excl) -> (merged [Q,k,3], totals [Q])`;
    `xdev`: where the exchanged tensors live ("cpu" for gloo, the GPU for RCCL);
    `group`: a process group used by NOTHING else (the tick thread issues collectives on it
    concurrently with whatever the other threads of the process do on theirs)."""

    def __init__(self, shard, matcher, group=None, xdev="cpu", tick_s: float = 0.0005, max_batch: int = 1024,
                 wide_k_matcher=None):
        self.shard, self.matcher, self.group = shard, matcher, group
        self.wide = wide_k_matcher              # a matcher with a larger k for tie sets that exceed k (optional)
        self.xdev = torch.device(xdev)
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.tick_s, self.max_batch = float(tick_s), int(max_batch)
        self._cv = threading.Condition()
        self._pending: List[tuple] = []
        self._stop = False
        self.ticks = 0
        self.busy_ticks = 0
        self._thread = threading.Thread(target=self._loop, name="tvz-rank-tick", daemon=True)
        self._thread.start()

    # ---- rows ----
    def upload(self, rows) -> None:
        """The initial table: this rank keeps video_id mod world == rank."""
        self.shard.upload([(int(v), list(t)) for v, t in rows if int(v) % self.world == self.rank])

    def upsert(self, video_id: int, timestamps) -> None:
        self.shard.upsert(int(video_id), timestamps)         # ingested here: lives here

    def clear(self) -> None:
        self.shard.clear()

    def stats(self):
        return self.shard.stats()

    # ---- matches ----
    def find_duplicates(self, new_timestamps, min_match: int = 5, exclude_id: int = -1, with_kth: bool = False):
        q = np.ascontiguousarray(np.asarray(new_timestamps, dtype=np.float64))
        if q.size > MAX_BATCH_LEN or not 1 <= int(min_match) <= 5:
            raise RuntimeError("RankCorpus answers asks of up to 4095 timestamps with min_match 1..5")
        fut: Future = Future()
        with self._cv:
            if self._stop:
                raise RuntimeError("rank corpus is closed")
            self._pending.append((q, int(min_match), int(exclude_id), 0, fut))
            self._cv.notify()
        rows, total = fut.result()
        hits, exact = _hits_from_topk(rows, total)
        if not exact and self.wide is not None:
            fut = Future()
            with self._cv:
                self._pending.append((q, int(min_match), int(exclude_id), 1, fut))
            rows, total = fut.result()
            hits, exact = _hits_from_topk(rows, total)
        if not exact:
            raise RuntimeError("more rows share the verdict's prefix than the exchanged top-k holds; "
                               "raise k (sharded matcher) for this corpus")
        return hits if with_kth else sorted((v, c) for v, c, _ in hits)

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        """[...] -> [world, ...] (dim-0 concatenation is the layout RCCL and gloo both accept)."""
        if self.world == 1:
            return t.unsqueeze(0)
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def _loop(self):
        while True:
            with self._cv:
                stop = self._stop
                take = self._pending[: self.max_batch]
                self._pending = self._pending[len(take):]
            # 1) who has how much (and does anybody want to stop: all ranks leave together)
            meta = torch.tensor([len(take), max((len(p[0]) for p in take), default=0), 1 if stop else 0],
                                dtype=torch.int64, device=self.xdev)
            allmeta = self._gather(meta).cpu().numpy()
            self.ticks += 1
            if int(allmeta[:, 0].sum()) == 0:
                if int(allmeta[:, 2].min()) == 1:            # every rank is closing and nothing is pending anywhere
                    return
                time.sleep(self.tick_s)
                continue
            self.busy_ticks += 1
            try:
                self._exchange_and_answer(take, allmeta)
            except BaseException as e:
                for p in take:
                    if not p[4].done():
                        p[4].set_exception(e)

    def _exchange_and_answer(self, take, allmeta):
        Qcap, Lcap = int(allmeta[:, 0].max()), max(int(allmeta[:, 1].max()), 1)
        # 2) the asks, padded to the largest rank's block: keys [Qcap, Lcap] + (len, exclude, min_match, wide)
        keys = np.zeros((Qcap, Lcap), dtype=np.float64)
        info = np.zeros((Qcap, 4), dtype=np.int64)
        for i, (q, mm, ex, wide, _) in enumerate(take):
            keys[i, :len(q)] = q
            info[i] = (len(q), ex, mm, wide)
        g_keys = self._gather(torch.from_numpy(keys).to(self.xdev)).cpu().numpy()
        g_info = self._gather(torch.from_numpy(info).to(self.xdev)).cpu().numpy()
        # the global batch, in the same order on every rank: rank-major, then (min_match, wide) groups
        asks = [(r, i) for r in range(self.world) for i in range(int(allmeta[r, 0]))]
        groups = sorted({(int(g_info[r, i, 2]), int(g_info[r, i, 3])) for r, i in asks})
        for mm, wide in groups:
            sel = [(r, i) for r, i in asks if (int(g_info[r, i, 2]), int(g_info[r, i, 3])) == (mm, wide)]
            qs = [g_keys[r, i, :int(g_info[r, i, 0])] for r, i in sel]
            lens = np.array([len(x) for x in qs], dtype=np.int64)
            offs = np.zeros(len(qs) + 1, dtype=np.int64)
            np.cumsum(lens, out=offs[1:])
            flat = np.concatenate(qs) if int(offs[-1]) else np.zeros(1)
            mdev = getattr(self.matcher, "dev", torch.device("cpu"))
            d_q = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float64)).to(mdev)
            d_off = torch.from_numpy(offs).to(mdev)
            d_ex = torch.tensor([int(g_info[r, i, 1]) for r, i in sel], dtype=torch.int32, device=mdev)
            m = self.wide if wide and self.wide is not None else self.matcher
            merged, totals = m.match_topk(d_q, d_off, int(lens.max()) if len(lens) else 0, mm, d_ex)
            merged, totals = merged.cpu().numpy(), totals.cpu().numpy()
            for j, (r, i) in enumerate(sel):
                if r == self.rank:
                    take[i][4].set_result((merged[j].copy(), int(totals[j])))

    def close(self) -> None:
        """Collective: every rank calls it; the loops leave together once nothing is pending."""
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=60)
        if hasattr(self.shard, "close"):
            self.shard.close()
