"""GPU parity of the lookup that keeps the per-shard top-k itself (ts_match_index_topk_kernel: the path
tvz_match_topk / tvz_match_sharded take on an indexed corpus) against the oracle's restatement of
db.find_duplicates (inspector/db.py:76-94) ordered as the streaming verdict needs it (earliest prefix
first, inspector/app.py:238-245: ascending kth, then video_id), and against the unfused pipeline
(tvz_match -> tvz_topk_shard).  Bit-exact rows, totals and padding."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tvidz_amd import _lib, corpus as tc, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NEVER = tc.KTH_NEVER
WAVE_RUNS = []          # (indexed rows, queries) of every comparison that ran the one-wave lookup


@pytest.fixture()
def dc():
    c = tc.DeviceCorpus(0)
    yield c
    c.close()


def _expected_rows(rows, queries, mm, excl=None):
    ids, offs, keys = tc.rows_to_csr(rows)
    out = []
    for qi, q in enumerate(queries):
        cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
        out.append([(int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(ids))
                    if cnt[c] >= mm and (excl is None or ids[c] != excl[qi])])
    return out


def _check_topk(dc, exp_rows, queries, mm, k, cap, excl=None, algo=_lib.ALGO_AUTO):
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    Q = len(queries)
    d_ex = torch.tensor(excl, dtype=torch.int32, device=DEV) if excl is not None else None
    ws = torch.empty(tc.workspace_bytes(Q, max_len, cap, k), dtype=torch.uint8, device=DEV)
    # both shapes of the fused lookup - two queries per block (their probes share one phase) and one - must agree
    out = dc.match_topk(d_q, d_off, max_len, mm, cap, k, d_exclude_ids=d_ex, workspace=ws, algo=algo | _lib.ALGO_PAIR)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    one = dc.match_topk(d_q, d_off, max_len, mm, cap, k, d_exclude_ids=d_ex, workspace=ws,
                        algo=algo | _lib.ALGO_NO_PAIR | _lib.ALGO_NO_WAVE)
    assert (one.cpu().numpy() == out).all()
    # ... and, on a handle of one sub-index, the lookup that gives every query to one WAVE (ts_match_wq_topk_kernel)
    st = dc.index_stats()
    if 0 < st["indexed_rows"] <= 16384 and max_len <= 512 and 1 <= mm <= 5 and k <= 64 and algo != _lib.ALGO_JOIN:
        WAVE_RUNS.append((st["indexed_rows"], Q))
        wv = dc.match_topk(d_q, d_off, max_len, mm, cap, k, d_exclude_ids=d_ex, workspace=ws, algo=algo | _lib.ALGO_WAVE)
        wv = wv.cpu().numpy()
        bad = np.flatnonzero((wv != out).reshape(Q, -1).any(axis=1))
        assert bad.size == 0, (mm, k, cap, bad[:8], wv[bad[0]][:4], out[bad[0]][:4])
    for qi in range(Q):
        rows = exp_rows[qi]
        assert tuple(out[qi, k][[0, 2]]) == (-1, NEVER)
        tot = int(out[qi, k, 1])
        got = [tuple(int(x) for x in r) for r in out[qi, :k]]
        exp = sorted(rows, key=lambda h: (h[2], h[0], h[1]))[:k]
        exp += [(-1, 0, NEVER)] * (k - len(exp))
        if len(rows) <= cap:
            assert tot == len(rows), (qi, tot, len(rows))
            assert got == exp, (mm, k, cap, qi)
        else:
            # the contract of a truncated hit list: the total is negated and the rows are real hits
            # (the fused lookup has no list to truncate: its rows are still the exact k best)
            assert tot == -len(rows), (qi, tot, len(rows))
            assert set(r for r in got if r[0] >= 0) <= set(rows)
    return out


@pytest.mark.parametrize("C,mean_len,Q", [(3000, 200, 40), (400, 30, 9)])
def test_fused_topk_equals_oracle(dc, C, mean_len, Q):
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=C + 1, mean_len=mean_len, dup_frac=0.05, frag_frac=0.05)
    rows = [(int(ids[r]), keys[offs[r]:offs[r + 1]].tolist()) for r in range(C)]
    dc.upload(rows)
    assert dc.index_stats()["indexed_rows"] == C
    queries = synth.synth_queries(ids, offs, keys, Q, seed=Q, mean_len=mean_len)
    queries[0] = np.concatenate([queries[0], queries[0][:7]])        # multiplicity
    queries[1] = np.array([float("nan"), -0.0, 0.0] + queries[1][:5].tolist())
    queries[2] = np.zeros(0)                                         # empty query: padding + total 0
    excl = [int(ids[(7 * i) % C]) for i in range(Q)]
    for mm in (1, 2, 3, 5):
        exp = _expected_rows(rows, queries, mm)
        exp_x = _expected_rows(rows, queries, mm, excl)
        for k in (1, 16, 64):
            _check_topk(dc, exp, queries, mm, k, C)
            _check_topk(dc, exp_x, queries, mm, k, C, excl=excl, algo=_lib.ALGO_INDEX)
        _check_topk(dc, exp, queries, mm, 16, 5)                      # tiny cap: totals negated where they exceed it
        _check_topk(dc, exp, queries, mm, 100, C)                     # k > 64: the unfused pipeline, same answer


def test_fused_topk_with_huge_ties_and_late_first_hits(dc):
    """The rare paths of the in-block selection: ~3000 hits in ONE kth bin (more than the 128-entry
    list holds: reduced by rank counting, fed in rounds), hits whose kth is all beyond the histogram's
    exact bins (the first 70 query positions match nothing), duplicate video ids (identical rows)."""
    C = 6000
    rng = np.random.default_rng(21)
    grid = np.arange(1, 2001) / 8.0
    rows = []
    for c in range(C):
        r = rng.choice(grid, size=int(rng.integers(5, 40)), replace=False)
        if rng.random() < 0.5:
            r = np.append(r, 777.125)                      # half the rows share one key
        rows.append((c + 1 if c % 50 else 7, r.tolist()))  # video id 7 owns 120 rows
    dc.upload(rows)
    Q = 12
    queries = [rng.choice(grid, size=int(rng.integers(20, 160)), replace=False) for _ in range(Q)]
    queries[1] = np.array([777.125] * 3)                   # every hit has kth == 0
    queries[2] = np.concatenate([np.arange(70) + 5000.5, queries[2]])       # nothing matches before position 70
    queries[3] = np.concatenate([np.arange(70) + 5000.5, [777.125], queries[3]])
    queries[4] = np.zeros(0)
    excl = [int(rows[(5 * i) % C][0]) for i in range(Q)]
    for mm in (1, 2, 4):
        exp = _expected_rows(rows, queries, mm)
        assert mm > 1 or (len(exp[1]) > 2500 and len({r[2] for r in exp[1]}) == 1)
        exp_x = _expected_rows(rows, queries, mm, excl)
        for k, cap in ((16, C), (1, C), (64, C), (16, 300), (8, 512)):
            _check_topk(dc, exp, queries, mm, k, cap)
            _check_topk(dc, exp_x, queries, mm, k, cap, excl=excl)


def test_fused_topk_over_several_sub_indexes_and_a_delta_table(dc):
    """40k rows = 3 sub-indexes; then upserts (replaced indexed rows = dead postings + delta rows, new
    rows): the lookup's block and the delta sweep's block are merged; equal to the unfused pipeline
    and the oracle before and after."""
    C = 40_000
    rng = np.random.default_rng(9)
    alphabet = np.arange(1, 30_001) / 4.0
    rows = [(c + 1, rng.choice(alphabet, size=int(rng.integers(3, 9)), replace=False).tolist()) for c in range(C)]
    dc.upload(rows)
    st = dc.index_stats()
    assert st["indexed_rows"] == C and st["delta_rows"] == 0
    queries = [rng.choice(alphabet, size=n, replace=False) for n in (40, 200, 300, 500, 64, 1, 0)]
    queries.append(np.asarray(rows[123][1] + rows[30_000][1]))
    Q = len(queries)
    excl = [int(rows[(11 * i) % C][0]) for i in range(Q)]

    def both(mm):
        exp = _expected_rows(rows, queries, mm)
        exp_x = _expected_rows(rows, queries, mm, excl)
        for k in (16, 64):
            out = _check_topk(dc, exp, queries, mm, k, 4096)
            _check_topk(dc, exp_x, queries, mm, k, 4096, excl=excl)
            # the unfused pipeline on the same handle (tvz_match -> tvz_topk_shard)
            d_q, d_off, max_len = tc.pack_queries(queries, DEV)
            hits, n = dc.match(d_q, d_off, max_len, mm, 4096, algo=_lib.ALGO_INDEX)
            blk = tc.topk_shard(hits, n, k)
            torch.cuda.synchronize()
            assert (blk.cpu().numpy() == out).all()

    for mm in (1, 2, 3):
        both(mm)
    for i in range(300):                                    # below the rebuild trigger: rows stay in the delta table
        if i % 3 == 0:
            v, ts = C + 1 + i, rng.choice(alphabet, size=5, replace=False).tolist()
            rows.append((v, ts))
        else:
            r = int(rng.integers(0, C))
            v, ts = rows[r][0], (queries[1][:6].tolist() if i % 2 else rng.choice(alphabet, size=4).tolist())
            rows[r] = (v, ts)
        dc.upsert(v, ts)
    st = dc.index_stats()
    assert 280 <= st["delta_rows"] <= 300 and st["builds"] == 1
    for mm in (1, 2, 3):
        both(mm)


def test_sharded_call_uses_the_fused_lookup_and_matches_the_oracle(dc):
    """tvz_match_sharded with a one-rank communicator would need RCCL in this process; the same
    pipeline without the collective: per-shard fused blocks of an 8-way split, stacked as the
    all-gather delivers them, merged - equal to the oracle's global top-k."""
    C, R, k = 24_000, 8, 16
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=77, mean_len=40, dup_frac=0.03, frag_frac=0.03)
    rows = [(int(ids[r]), keys[offs[r]:offs[r + 1]].tolist()) for r in range(C)]
    queries = synth.synth_queries(ids, offs, keys, 96, seed=5, mean_len=40)
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    shards, blocks = [], []
    try:
        for r in range(R):
            s = tc.DeviceCorpus(0)
            s.upload(rows[r * C // R:(r + 1) * C // R])
            shards.append(s)
            ws = torch.empty(tc.workspace_bytes(len(queries), max_len, 4096, k), dtype=torch.uint8, device=DEV)
            blocks.append(s.match_topk(d_q, d_off, max_len, 2, 4096, k, workspace=ws))
        merged, totals = tc.topk_merge(torch.stack(blocks).contiguous(), k)
        torch.cuda.synchronize()
    finally:
        for s in shards:
            s.close()
    merged, totals = merged.cpu().numpy(), totals.cpu().numpy()
    exp = _expected_rows(rows, queries, 2)
    for qi in range(len(queries)):
        want = sorted(exp[qi], key=lambda h: (h[2], h[0], h[1]))[:k]
        want += [(-1, 0, NEVER)] * (k - len(want))
        assert [tuple(int(x) for x in r) for r in merged[qi]] == want
        assert int(totals[qi]) == len(exp[qi])


def test_two_queries_per_block_refusals_odd_batches_and_neighbours(dc):
    """The lookup takes two queries per block when the LDS of both fits (their directory probes share the
    block's one probe phase).  What must not leak between the two: a query LONGER than the stated
    max_query_len is refused on its own (padding rows + total INT32_MIN: the contract of every batched
    kernel) whichever side of a pair it sits on, its neighbour is answered; an odd batch's last block
    holds one query; Q = 1 takes the one-query form; queries of very different lengths, empty ones and
    heavy ties sit next to each other; the second query's top-k list and histogram start clean."""
    C = 2500
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=77, mean_len=60, dup_frac=0.06, frag_frac=0.05)
    rows = [(int(ids[r]), keys[offs[r]:offs[r + 1]].tolist()) for r in range(C)]
    # forty true copies of one row: ~40 hits tied at one kth for the queries that hold its keys
    copy_of = rows[5][1]
    rows += [(90000 + i, list(copy_of)) for i in range(40)]
    dc.upload(rows)
    base = synth.synth_queries(ids, offs, keys, 9, seed=5, mean_len=60)
    long_q = np.concatenate([np.asarray(copy_of, dtype=np.float64), 1e7 + np.arange(400, dtype=np.float64)])
    queries = [base[0], long_q, long_q, base[1], np.zeros(0), np.asarray(copy_of, dtype=np.float64),
               np.asarray(copy_of[:3], dtype=np.float64), base[2], base[3]]           # 9 queries: odd
    honest = max(len(q) for q in queries)
    lying = 200                                                  # queries 1 and 2 (len > 400) exceed it
    for mm in (1, 2, 4):
        exp = _expected_rows(rows, queries, mm)
        for k in (4, 16):
            d_q, d_off, _ = tc.pack_queries(queries, DEV)
            for bound in (honest, lying):
                ws = torch.empty(tc.workspace_bytes(len(queries), bound, 4096, k), dtype=torch.uint8, device=DEV)
                out = dc.match_topk(d_q, d_off, bound, mm, 4096, k, workspace=ws, algo=_lib.ALGO_PAIR).cpu().numpy()
                for qi, q in enumerate(queries):
                    want = sorted(exp[qi], key=lambda h: (h[2], h[0], h[1]))[:k]
                    want += [(-1, 0, NEVER)] * (k - len(want))
                    got = [tuple(int(x) for x in r) for r in out[qi, :k]]
                    if len(q) > bound:
                        assert int(out[qi, k, 1]) == np.iinfo(np.int32).min and all(r == (-1, 0, NEVER) for r in got), (mm, k, qi)
                    else:
                        assert int(out[qi, k, 1]) == len(exp[qi]) and got == want, (mm, k, bound, qi)
            # the short queries alone (their bound leaves room for two per block), against the oracle
            short = [q for q in queries if len(q) <= lying]
            _check_topk(dc, [e for e, q in zip(exp, queries) if len(q) <= lying], short, mm, k, 4096)
            # every prefix of the batch: Q = 1 (one query per block), even and odd batches
            for Qn in (1, 2, 3, 6):
                d_q1, d_off1, ml1 = tc.pack_queries(queries[:Qn], DEV)
                out1 = dc.match_topk(d_q1, d_off1, ml1, mm, 4096, k, algo=_lib.ALGO_PAIR).cpu().numpy()
                full = dc.match_topk(d_q, d_off, honest, mm, 4096, k, algo=_lib.ALGO_NO_PAIR).cpu().numpy()
                assert (out1 == full[:Qn]).all(), (mm, k, Qn)


def test_one_wave_lookup_on_a_shard_of_configs3(dc):
    """ts_match_wq_topk_kernel on what a GPU of BASELINE.json configs[3] holds - rank 0's 1/8 of the 100k-video
    table, one sub-index - against the block kernel (all queries) and the oracle (sampled), min_match 2 and 5,
    with per-query exclusions; then with indexed rows replaced (dead postings) and a delta table behind it.
    A handle of more than one sub-index refuses TVZ_ALGO_WAVE by name."""
    from tvidz_amd import sharded
    ids, offs, keys = synth.synth_timestamp_corpus(100_000, seed=synth.CORPUS_SEED)
    s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, keys, 0, 8)
    dc.upload_csr(s_ids, s_offs, s_keys)
    assert dc.index_stats()["indexed_rows"] == len(s_ids) <= 16384
    Q = 1024
    queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 3)
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    excl = torch.tensor([int(s_ids[(11 * i) % len(s_ids)]) for i in range(Q)], dtype=torch.int32, device=DEV)
    rows = None
    for phase in range(2):
        if phase == 1:                                  # 300 indexed rows replaced, 100 rows added: a delta table
            rng = np.random.default_rng(5)
            for j in range(300):
                r = int(rng.integers(0, len(s_ids)))
                dc.upsert(int(s_ids[r]), s_keys[s_offs[r]:s_offs[r + 1]][::2].tolist())
            for j in range(100):
                dc.upsert(9_000_000 + j, queries[j][:50].tolist())
            assert dc.index_stats()["delta_rows"] > 0
        for mm in (2, 5):
            for ex in (None, excl):
                blk = dc.match_topk(d_q, d_off, max_len, mm, 4096, 16, d_exclude_ids=ex,
                                    algo=_lib.ALGO_NO_WAVE | _lib.ALGO_NO_PAIR).cpu().numpy()
                wav = dc.match_topk(d_q, d_off, max_len, mm, 4096, 16, d_exclude_ids=ex, algo=_lib.ALGO_WAVE).cpu().numpy()
                auto = dc.match_topk(d_q, d_off, max_len, mm, 4096, 16, d_exclude_ids=ex, algo=_lib.ALGO_PREFER_WAVE).cpu().numpy()
                assert (wav == blk).all() and (auto == blk).all(), (phase, mm)
        if phase == 0:
            rows = [(int(s_ids[r]), s_keys[s_offs[r]:s_offs[r + 1]].tolist()) for r in range(len(s_ids))]
    # the oracle on the first phase's answer (recomputed: the handle has moved on)
    dc.upload_csr(s_ids, s_offs, s_keys)
    wav = dc.match_topk(d_q, d_off, max_len, 2, 4096, 16, algo=_lib.ALGO_WAVE).cpu().numpy()
    exp = _expected_rows(rows, queries[:24], 2)
    for qi in range(24):
        e = sorted(exp[qi], key=lambda h: (h[2], h[0], h[1]))[:16]
        e += [(-1, 0, NEVER)] * (16 - len(e))
        assert [tuple(int(x) for x in r) for r in wav[qi, :16]] == e and int(wav[qi, 16, 1]) == len(exp[qi]), qi
    big = tc.DeviceCorpus(0)
    try:
        big.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 4))          # 25k rows: two sub-indexes
        with pytest.raises(RuntimeError, match="TVZ_ALGO_WAVE"):
            big.match_topk(d_q, d_off, max_len, 2, 4096, 16, algo=_lib.ALGO_WAVE)
        a = big.match_topk(d_q, d_off, max_len, 2, 4096, 16, algo=_lib.ALGO_PREFER_WAVE)      # ... a preference is not an error
        assert (a == big.match_topk(d_q, d_off, max_len, 2, 4096, 16)).all()
    finally:
        big.close()


def test_the_one_wave_lookup_was_compared_in_this_module():
    assert len(WAVE_RUNS) >= 20, WAVE_RUNS
