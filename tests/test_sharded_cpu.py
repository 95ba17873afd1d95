"""CPU tests of the N>1 path: world_size-2 gloo, one process per "GPU".

The sharding plan, the all-gather of per-shard top-k and the merge order are exercised with a
stand-in backend built on the oracle (the product backend is HIP-only and is covered by the
-m gpu tests); what is checked is that the merged result of the sharded run equals the oracle's
answer over the whole corpus, on every rank.
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from tests.fakes import OracleBackend
from tvidz_amd import sharded, synth

NEVER = sharded.KTH_NEVER


def _expected(ids, offs, keys, queries, mm, k, excl=None):
    exp = []
    for qi, q in enumerate(queries):
        cnt, kth = oracle.match_kth_csr(q, offs, keys, mm)
        rows = [(int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(ids))
                if cnt[c] >= mm and (excl is None or ids[c] != excl[qi])]
        tot = len(rows)
        rows = sorted(rows, key=lambda h: (h[2], h[0], h[1]))[:k]
        rows += [(-1, 0, NEVER)] * (k - len(rows))
        exp.append((rows, tot))
    return exp


def _worker(rank, world, port, C, Q, mm, k, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids, offs, keys = synth.synth_timestamp_corpus(C, seed=77, mean_len=40, dup_frac=0.05, frag_frac=0.05)
        queries = synth.synth_queries(ids, offs, keys, Q, seed=5, mean_len=40)
        excl = [int(ids[(3 * i) % C]) for i in range(Q)]
        s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, keys, rank, world)
        sm = sharded.ShardedMatcher(OracleBackend(s_ids, s_offs, s_keys), k=k, cap=cap)
        assert sm.world == world and sm.rank == rank
        lens = [len(x) for x in queries]
        d_off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
        d_q = torch.from_numpy(np.concatenate(queries))
        ex = torch.tensor(excl, dtype=torch.int32)
        exp = _expected(ids, offs, keys, queries, mm, k, excl)
        # which queries overflow the per-shard capacity on SOME shard (the totals must say so)
        any_overflow = [False] * Q
        for r in range(world):
            r_ids, r_offs, r_keys = sharded.shard_csr(ids, offs, keys, r, world)
            for qi, qq in enumerate(queries):
                c_, _ = oracle.match_kth_csr(qq, r_offs, r_keys, mm)
                n_r = sum(1 for c in range(len(r_ids)) if c_[c] >= mm and r_ids[c] != excl[qi])
                any_overflow[qi] = any_overflow[qi] or n_r > cap
        # plain call, then two batches pipelined (submit i+1 before finishing i)
        t1 = sm.submit(d_q, d_off, max(lens), mm, ex)
        t2 = sm.submit(d_q, d_off, max(lens), mm, ex)
        for merged, totals in (sm.match_topk(d_q, d_off, max(lens), mm, ex), sm.finish(t1), sm.finish(t2)):
            for qi in range(Q):
                if not any_overflow[qi]:
                    assert [tuple(int(x) for x in r) for r in merged[qi]] == exp[qi][0], (rank, qi)
                assert abs(int(totals[qi])) == exp[qi][1]
                assert (int(totals[qi]) < 0) == any_overflow[qi]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


# (world 8: the rank count of BASELINE.json configs[3] - eight blocks gathered and merged on every rank)
@pytest.mark.parametrize("world,C,Q,mm,k,cap", [(2, 300, 6, 2, 8, 64), (2, 120, 4, 1, 16, 16), (3, 90, 3, 2, 4, 8),
                                                (8, 240, 4, 2, 8, 32)])
def test_sharded_match_equals_whole_corpus(world, C, Q, mm, k, cap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + world * 7 + C) % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, Q, mm, k, cap, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, "ok") for r in range(world)], res
    assert all(p.exitcode == 0 for p in procs)


def test_shard_bounds_balance_by_keys_not_rows():
    offs = np.concatenate([[0], np.cumsum([1000] * 10 + [10] * 990)]).astype(np.int64)
    for world in (1, 2, 4, 8):
        b = sharded.shard_bounds(offs, world)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all()
        per = [int(offs[b[i + 1]] - offs[b[i]]) for i in range(world)]
        assert sum(per) == int(offs[-1])
        assert max(per) - min(per) <= 2000          # within two of the longest rows
    ids = np.arange(1000, dtype=np.int32)
    keys = np.arange(int(offs[-1]), dtype=np.float64)
    got = [sharded.shard_csr(ids, offs, keys, r, 4) for r in range(4)]
    assert np.concatenate([g[0] for g in got]).tolist() == ids.tolist()
    assert np.concatenate([g[2] for g in got]).tolist() == keys.tolist()
    for g in got:
        assert g[1][0] == 0 and g[1][-1] == len(g[2])


def test_empty_shards_are_legal():
    offs = np.array([0, 5, 9], dtype=np.int64)
    b = sharded.shard_bounds(offs, 8)
    assert b[0] == 0 and b[-1] == 2 and (np.diff(b) >= 0).all()
    ids, o, k = sharded.shard_csr(np.array([1, 2], dtype=np.int32), offs, np.arange(9.0), 7, 8)
    assert o[0] == 0 and len(o) == len(ids) + 1


def test_verdicts_from_topk():
    m = np.array([[[5, 2, 3], [9, 2, 3], [4, 2, 7], [-1, 0, NEVER]],
                  [[-1, 0, NEVER]] * 4,
                  [[1, 2, 0], [2, 2, 0], [3, 2, 0], [4, 2, 0]]], dtype=np.int32)
    v = sharded.verdicts_from_topk(m)
    assert v[0] == (3, [5, 9], False)
    assert v[1] == (None, [], False)
    assert v[2] == (0, [1, 2, 3, 4], True)      # tie set may continue past k
