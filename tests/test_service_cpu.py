"""configs[4]'s N-rank composition at world size 2 on gloo (CPU): service.RankCorpus - every rank
keeps the rows of the uploads it ingests, the asks of all ranks are exchanged on a tick, answered
against every shard and merged (sharded.ShardedMatcher; the shards' match is the oracle here, the
HIP backend is covered by the -m gpu tests).  Checked: concurrent asks from threads on both ranks
get the whole-table answer; an add_timestamps on one rank is seen by the next ask of the other;
the streaming verdict (app.py:235-255) over the sharded table equals the oracle's replay."""
import os
import threading

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from tests.fakes import OracleBackend, OracleCorpus
from tvidz_amd import service, sharded, synth


def _whole_table_hits(rows, q, mm, excl):
    ids, cnt, kth = oracle.match_kth(rows, list(q), mm)
    return sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(rows)) if cnt[c] >= mm and ids[c] != excl)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rc = None
    try:
        ids, offs, keys = synth.synth_timestamp_corpus(60, seed=9, mean_len=30, dup_frac=0.1, frag_frac=0.1)
        table = [(int(ids[c]), keys[offs[c]:offs[c + 1]].tolist()) for c in range(60)]
        shard = OracleCorpus()
        # the tick thread's collectives get a process group of their own: this thread's barriers use the
        # default group, and two threads must not interleave collectives on one group
        g = dist.new_group(backend="gloo")
        matcher = sharded.ShardedMatcher(OracleBackend(live=shard), k=8, cap=64, group=g)
        rc = service.RankCorpus(shard, matcher, group=g, xdev="cpu", tick_s=0.002)
        rc.upload(table)
        assert sorted(v for v, _ in shard.rows) == [v for v, _ in table if v % world == rank]
        dist.barrier()
        # (1) concurrent asks from four threads per rank: each gets the whole-table answer
        queries = synth.synth_queries(ids, offs, keys, 12, seed=3 + rank, mean_len=30)
        errs = []

        def ask(qi):
            try:
                q = queries[qi]
                exp = _whole_table_hits(table, q, 2, excl=int(ids[qi]))
                got = rc.find_duplicates(q, 2, exclude_id=int(ids[qi]), with_kth=True)
                top = sorted(exp, key=lambda h: (h[2], h[0], h[1]))
                assert sorted(got) == sorted(top[:len(got)]) and (len(got) == len(exp) or len(got) >= 8), (qi, got, exp)
                kstar = min((h[2] for h in exp), default=None)
                assert sorted(h[0] for h in got if h[2] == kstar) == sorted(h[0] for h in exp if h[2] == kstar)
            except Exception as e:                                        # pragma: no cover
                errs.append(repr(e))
        th = [threading.Thread(target=ask, args=(qi,)) for qi in range(12)]
        [t.start() for t in th]
        [t.join(120) for t in th]
        assert not errs, errs[:2]
        dist.barrier()
        # (2) read-your-writes across ranks: rank 0 ingests a new video, rank 1's next ask sees it
        new_ts = [5000.5 + i for i in range(9)]
        if rank == 0:
            rc.upsert(9001, new_ts)
        dist.barrier()
        got = rc.find_duplicates(new_ts[:4], 2, exclude_id=-1, with_kth=True)
        assert got == [(9001, 4, 1)], (rank, got)
        dist.barrier()
        # (3) the driver's loop (app.py:231-255) over the sharded table == the oracle's replay
        full = table + [(9001, new_ts)]
        stream = list(table[7][1][:3]) + [7777.25] + list(table[7][1][3:6]) if rank == 0 else [123.5, 9.25, 88.0, 5000.5, 5001.5]
        my_id = 9100 + rank
        exp_ts, exp_dups = oracle.streaming_verdict_py(stream, full, my_id, 2)
        seen, dups = [], []
        for ts in stream:
            seen.append(ts)
            hits = [h for h in rc.find_duplicates(seen, 2, exclude_id=my_id, with_kth=True) if h[2] < service.KTH_NEVER]
            if hits:
                kstar = min(h[2] for h in hits)
                dups = sorted(h[0] for h in hits if h[2] == kstar)
                seen = seen[:kstar + 1]
                break
        assert (seen, dups) == (list(exp_ts), sorted(exp_dups)), (rank, seen, dups, exp_ts, exp_dups)
        dist.barrier()
        # (4) never an error because ties exceed k (db.py:85-91 returns EVERY row, app.py:238-245
        # reports all rows of the earliest prefix): 24 copies spread over both ranks share kth 1 with k = 8
        copy_ts = [8000.25 + i for i in range(6)]
        for j in range(12):
            rc.upsert(20000 + 2 * j + rank, copy_ts)              # 12 rows ingested on each rank
        dist.barrier()
        full2 = full + [(20000 + i, copy_ts) for i in range(24)]
        before = rc.exact_asks
        got = rc.find_duplicates(copy_ts, 2, exclude_id=20000 + rank, with_kth=True)
        exp = _whole_table_hits(full2, copy_ts, 2, excl=20000 + rank)
        assert len(exp) == 23 and got == exp and all(h[2] == 1 for h in got), (rank, got, exp)
        assert rc.exact_asks == before + 1                        # the top-k said "may continue": answered exactly
        # db.find_duplicates' shape (every row with its count) beyond k, and through db.Store's call
        pairs = rc.find_duplicates(copy_ts, 2)
        assert pairs == sorted((v, c) for v, c, _ in _whole_table_hits(full2, copy_ts, 2, excl=-1)) and len(pairs) == 24
        # a query of more than 4095 timestamps and a min_match outside 1..5: the exact exchange, not an error
        longq = [-1.0 - i for i in range(4200)] + copy_ts     # (negative: in no row)
        got = rc.find_duplicates(longq, 2, exclude_id=-1, with_kth=True)
        assert got == _whole_table_hits(full2, longq, 2, excl=-1) and len(got) == 24
        got = rc.find_duplicates(copy_ts, 6, exclude_id=-1, with_kth=True)
        assert got == _whole_table_hits(full2, copy_ts, 6, excl=-1) and len(got) == 24
        assert rc.find_duplicates([1.5], 1) == [] and rc.find_duplicates([], 2, with_kth=True) == []
        dist.barrier()
        assert rc.busy_ticks >= 3 and rc.broken is None
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        if rc is not None:
            rc.close()                                    # collective: the tick loops leave together
        dist.destroy_process_group()


def test_rank_service_tick_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(240) for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(p.exitcode == 0 for p in procs)


def _failing_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rc = None
    try:
        shard = OracleCorpus()
        g = dist.new_group(backend="gloo")
        matcher = sharded.ShardedMatcher(OracleBackend(live=shard), k=4, cap=64, group=g)
        rc = service.RankCorpus(shard, matcher, group=g, xdev="cpu", tick_s=0.002)
        rc.upload([(10 + i, [float(i), float(i) + 0.5, 100.25]) for i in range(8)])
        dist.barrier()
        assert len(rc.find_duplicates([100.25, 1.0, 1.5], 1)) == 8           # the exchange works (an exact ask: 8 > k rows)
        dist.barrier()
        if rank == 1:                                                         # rank 1's own shard breaks
            def boom(*a, **k):
                raise ValueError("shard of rank 1 is gone")
            shard.find_duplicates = boom
        dist.barrier()
        # an exact ask from EITHER rank now fails on BOTH, in the same tick, with the failing rank named - nobody
        # hangs in a collective, nobody is answered from one shard only
        try:
            rc.find_duplicates([100.25, 1.0, 1.5], 1)
            out.put((rank, "answered"))
        except RuntimeError as e:
            msg = str(e)
            out.put((rank, "raised" if ("rank(s) [1]" in msg and "stops" in msg) or "broken" in msg else msg))
        # ... and so does every later ask, at once
        try:
            rc.find_duplicates([1.0], 1, with_kth=True)
            out.put((rank, "answered-later"))
        except RuntimeError:
            out.put((rank, "raised-later"))
        assert rc.broken is not None
    finally:
        dist.destroy_process_group()


def test_a_shard_that_fails_on_one_rank_stops_every_rank_in_the_same_tick():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 40
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(not p.is_alive() for p in procs), "a rank hung in a collective"
    res = sorted(q.get(timeout=5) for _ in range(4))
    assert res == [(0, "raised"), (0, "raised-later"), (1, "raised"), (1, "raised-later")], res


def test_hits_from_topk_flags_tie_sets_that_may_continue():
    N = service.KTH_NEVER
    full = np.array([[1, 2, 0], [2, 2, 0], [3, 2, 0], [4, 2, 0]], dtype=np.int32)
    assert service._hits_from_topk(full, 4) == ([(1, 2, 0), (2, 2, 0), (3, 2, 0), (4, 2, 0)], True)      # nothing beyond k
    assert service._hits_from_topk(full, 9)[1] is False                                                  # ties may go on
    done = np.array([[1, 2, 0], [2, 2, 0], [3, 2, 5], [4, 2, 6]], dtype=np.int32)
    assert service._hits_from_topk(done, 9)[1] is True                                                   # the tie set is complete
    assert service._hits_from_topk(done, -9)[1] is False                                                 # a shard's list overflowed
    pad = np.array([[7, 3, 2], [-1, 0, N], [-1, 0, N], [-1, 0, N]], dtype=np.int32)
    assert service._hits_from_topk(pad, 1) == ([(7, 3, 2)], True)
