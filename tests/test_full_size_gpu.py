"""Parity at BASELINE.json's full sizes, through size-independent properties plus sampled
oracle checks (the oracle cannot redo 20 GB of frames or 10^8 pairs in seconds, but it can redo
any sampled frame pair and any sampled query exactly)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tvidz_amd import _lib, corpus as tc, scene, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_config1_10k_1080p_frames():
    T, H, W = 10000, 1080, 1920
    frames, layout = synth.synth_luma(T, H, W, device=DEV, seed=synth.FRAME_SEED)
    sc = scene.SceneScorer(H, W, T, DEV)
    sad, mafd, score, sel = sc.score_batch(frames, carry=False)
    torch.cuda.synchronize()
    sad_h = sad.cpu().numpy().view(np.uint64).copy()
    sel_h, score_h = sel.cpu().numpy().copy(), score.cpu().numpy().copy()
    assert sad_h[0] == 0
    # (1) sampled frame pairs against the oracle, bit-exact (incl. every generator boundary nearby)
    rng = np.random.default_rng(0)
    sample = sorted(set(rng.integers(1, T, 40).tolist() + layout[:12] + [1, T - 1, 255, 256, 257]))
    for t in sample:
        pair = frames[t - 1:t + 1].cpu().numpy()
        assert int(oracle.luma_sad(pair)[1]) == int(sad_h[t]), t
    # (2) the epilogue over the whole SAD vector equals the oracle's (exact double/float32 ops)
    o_sel, o_score, o_mafd, _ = oracle.scene_select(sad_h, H, W, 0.3)
    assert (sel_h == o_sel).all() and (score_h == o_score).all() and (mafd.cpu().numpy() == o_mafd).all()
    # (3) chunk invariance: ten 1000-frame batches with carried state == one 10k batch
    sc2 = scene.SceneScorer(H, W, 1000, DEV)
    parts = []
    for s0 in range(0, T, 1000):
        part = frames[s0:s0 + 1000]
        _, _, _, s_ = sc2.score_batch(part)
        parts.append(s_.cpu().numpy().copy())
    assert (np.concatenate(parts) == sel_h).all()
    # (4) the cuts are scene boundaries of the generator (or its adversarial flash), and isolated
    #     boundaries are all found
    got = set(np.flatnonzero(sel_h).tolist())
    assert len(got) >= 40
    for c in layout:
        if (c - 1) not in layout and (c + 1) not in layout and c not in got:
            assert o_mafd[c] <= 30.0 or abs(o_mafd[c] - o_mafd[c - 1]) <= 30.0


def test_config3_100k_corpus_both_kernels_and_oracle():
    C, Q, CAP = 100000, 256, 16384
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
    queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
    dc = tc.DeviceCorpus(0)
    dc.upload_csr(ids, offs, keys)
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    res = {}
    for mode, algo in ((0, _lib.ALGO_TILE), (2, _lib.ALGO_JOIN)):     # LDS tile kernel, hash join
        hits, n = dc.match(d_q, d_off, max_len, 2, CAP, algo=algo)
        torch.cuda.synchronize()
        res[mode] = (hits.cpu().numpy(), n.cpu().numpy())
    assert (res[0][1] == res[2][1]).all(), np.flatnonzero(res[0][1] != res[2][1])[:8]
    # the per-query sweep on a few of the queries: same hit sets
    d_q8, d_off8, ml8 = tc.pack_queries(queries[:8], DEV)
    h8, n8 = dc.match(d_q8, d_off8, ml8, 2, CAP, algo=_lib.ALGO_Q1)
    torch.cuda.synchronize()
    h8, n8 = h8.cpu().numpy(), n8.cpu().numpy()
    for qi in range(8):
        assert sorted(map(tuple, h8[qi, :n8[qi]].tolist())) == sorted(map(tuple, res[0][0][qi, :res[0][1][qi]].tolist())), qi
    assert (res[0][1] <= CAP).all(), int(res[0][1].max())
    for qi in range(Q):
        a = sorted(map(tuple, res[0][0][qi, :res[0][1][qi]].tolist()))
        b = sorted(map(tuple, res[2][0][qi, :res[2][1][qi]].tolist()))
        assert a == b, qi
    # sampled queries against the oracle (rows are sorted-unique in the generator: binary search form)
    srt = keys.copy()
    for c in range(C):
        srt[offs[c]:offs[c + 1]].sort()
    for qi in (0, 1, 2, 3, 100, 255):
        cnt, kth = oracle.match_kth_csr(queries[qi], offs, srt, 2, sorted_unique=True)
        exp = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in np.flatnonzero(cnt >= 2))
        got = sorted(map(tuple, res[0][0][qi, :res[0][1][qi]].tolist()))
        assert got == exp, qi
    # round trip: a corpus row queried by its own timestamps is found with count == len, kth == 1
    for c in (0, 12345, C - 1):
        row = keys[offs[c]:offs[c + 1]]
        assert (int(ids[c]), len(row), 1) in dc.find_duplicates(row, 2, with_kth=True)
    dc.close()


# ---------------------------------------------------------------------------------------------
# The path bench.py times (AUTO = inverted-index lookup at C=100k, Q=4096) and configs[3]'s 8-way
# shard -> top-k -> merge, oracle-checked at the benchmarked size.
# ---------------------------------------------------------------------------------------------
C100K, QBIG, CAP_BIG = 100000, 4096, 16384


@pytest.fixture(scope="module")
def corpus100k():
    ids, offs, keys = synth.synth_timestamp_corpus(C100K, seed=synth.CORPUS_SEED)
    queries = synth.synth_queries(ids, offs, keys, QBIG, seed=synth.CORPUS_SEED + 1)
    srt = keys.copy()
    for c in range(C100K):
        srt[offs[c]:offs[c + 1]].sort()
    return ids, offs, srt, queries


def _hit_table(hits: torch.Tensor, n: torch.Tensor):
    """device hit lists [Q,cap,3] + counts [Q] -> (q << 32 | video_id sorted, count, kth) on the GPU."""
    Q, cap, _ = hits.shape
    nn = n.clamp(min=0, max=cap).to(torch.int64)
    mask = torch.arange(cap, device=hits.device)[None, :] < nn[:, None]
    sel = hits[mask]
    q = torch.repeat_interleave(torch.arange(Q, device=hits.device), nn)
    key = (q << 32) | sel[:, 0].to(torch.int64)
    order = torch.argsort(key)
    return key[order], sel[order, 1].clone(), sel[order, 2].clone()


def _same_hits(a, b):
    return a[0].shape == b[0].shape and bool((a[0] == b[0]).all()) and bool((a[1] == b[1]).all()) and \
        bool((a[2] == b[2]).all())


def _oracle_hits(ids, offs, srt, q, mm):
    cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, srt, mm, sorted_unique=True)
    return sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in np.flatnonzero(cnt >= mm))


def _query_hits(table, qi):
    key, cnt, kth = table
    lo = int(torch.searchsorted(key, torch.tensor(qi << 32, device=key.device)))
    hi = int(torch.searchsorted(key, torch.tensor((qi + 1) << 32, device=key.device)))
    vid = (key[lo:hi] & 0xffffffff).cpu().numpy()
    return sorted(zip(vid.tolist(), cnt[lo:hi].cpu().numpy().tolist(), kth[lo:hi].cpu().numpy().tolist()))


@pytest.mark.parametrize("mm", [2, 5])
def test_config3_100k_benchmarked_index_path_vs_sweeps_and_oracle(corpus100k, mm):
    """bench.py's match object times ALGO_AUTO (= the index lookup) at C=100k x Q=4096: that path,
    at that size, equals the corpus sweeps for EVERY query and the oracle for sampled queries, at
    min_match 2 (the driver's, app.py:235) and 5 (db.py:76's default), with an empty and with a
    non-empty delta table."""
    ids, offs, srt, queries = corpus100k
    sweep = _lib.ALGO_JOIN if mm <= 2 else _lib.ALGO_TILE
    dc = tc.DeviceCorpus(0)
    try:
        dc.upload_csr(ids, offs, srt)
        st = dc.index_stats()
        assert st["indexed_rows"] == C100K and st["delta_rows"] == 0
        sample_small = (0, 1, 2, 3, 100, 255)
        sample_big = (0, 5, 1000, 2047, 3001, QBIG - 1)

        def compare(ids_, offs_, srt_, tag):
            for Q, sample in ((256, sample_small), (QBIG, sample_big)):
                d_q, d_off, max_len = tc.pack_queries(queries[:Q], DEV)
                tabs = {}
                for algo in (_lib.ALGO_AUTO, _lib.ALGO_INDEX, sweep):
                    if algo == _lib.ALGO_INDEX and Q == QBIG:
                        continue                      # AUTO resolves to it; one 4096-query batch each is enough
                    ws = torch.empty(tc.workspace_bytes(Q, max_len), dtype=torch.uint8, device=DEV)
                    hits, n = dc.match(d_q, d_off, max_len, mm, CAP_BIG, algo=algo, workspace=ws)
                    torch.cuda.synchronize()
                    assert int(n.min()) >= 0 and int(n.max()) <= CAP_BIG, (tag, Q, algo, int(n.max()))
                    tabs[algo] = _hit_table(hits, n)
                    del hits, n, ws
                assert _same_hits(tabs[_lib.ALGO_AUTO], tabs[sweep]), (tag, Q, "AUTO != sweep")
                if _lib.ALGO_INDEX in tabs:
                    assert _same_hits(tabs[_lib.ALGO_INDEX], tabs[sweep]), (tag, Q, "INDEX != sweep")
                for qi in sample:
                    assert _query_hits(tabs[_lib.ALGO_AUTO], qi) == _oracle_hits(ids_, offs_, srt_, queries[qi], mm), \
                        (tag, Q, qi)
                del tabs
                torch.cuda.empty_cache()

        compare(ids, offs, srt, "no delta")
        # add_timestamps after the build: 200 indexed rows replaced (half of them by copies of query
        # videos - new true duplicates - a few emptied) + 100 new rows -> stale postings + delta table
        rng = np.random.default_rng(99)
        rows = [srt[offs[c]:offs[c + 1]] for c in range(C100K)]
        ids2 = ids.tolist()
        for j, c in enumerate(rng.choice(C100K, 200, replace=False).tolist()):
            if j % 2 == 0:
                new = np.sort(np.asarray(queries[(7 * j + 1) % QBIG], dtype=np.float64))
            elif j % 25 == 1:
                new = np.zeros(0)
            else:
                new = np.sort(rng.permutation(rows[c])[: max(2, len(rows[c]) // 2)])
            rows[c] = new
            dc.upsert(int(ids[c]), new.tolist())
        for j in range(100):
            new = np.sort(np.asarray(queries[(11 * j + 3) % QBIG], dtype=np.float64))[: 150 + j]
            rows.append(new)
            ids2.append(C100K + 10 + j)
            dc.upsert(C100K + 10 + j, new.tolist())
        st = dc.index_stats()
        assert st["delta_rows"] == 300 and st["indexed_rows"] == C100K, st
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        offs2 = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs2[1:])
        compare(np.asarray(ids2, dtype=np.int32), offs2, np.concatenate(rows), "delta of 300 rows")
    finally:
        dc.close()


def test_config3_100k_eight_way_shard_topk_merge_vs_oracle(corpus100k):
    """configs[3] as written, minus the wire: the 100k-video corpus split 8 ways by key count
    (sharded.shard_csr), each shard matched + reduced to its top-16 block (tvz_match_topk), the
    blocks stacked as ncclAllGather would deliver them, merged (tvz_topk_merge).  The merged lists
    equal the unsharded top-16 for EVERY query and the oracle's global top-16 for sampled queries;
    the totals equal the global hit counts."""
    from tvidz_amd import sharded
    ids, offs, srt, queries = corpus100k
    K, R = 16, 8
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    blocks = []
    bounds = sharded.shard_bounds(offs, R)
    assert bounds[0] == 0 and bounds[-1] == C100K and (np.diff(bounds) > 0).all()
    for r in range(R):
        s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, srt, r, R)
        dc = tc.DeviceCorpus(0)
        dc.upload_csr(s_ids, s_offs, s_keys)
        blocks.append(dc.match_topk(d_q, d_off, max_len, 2, 4096, K))
        torch.cuda.synchronize()
        dc.close()
    merged, totals = tc.topk_merge(torch.stack(blocks).contiguous(), K)
    dc = tc.DeviceCorpus(0)
    dc.upload_csr(ids, offs, srt)
    whole = dc.match_topk(d_q, d_off, max_len, 2, CAP_BIG, K)
    torch.cuda.synchronize()
    dc.close()
    assert bool((whole[:, K, 1] >= 0).all()) and bool((totals >= 0).all())      # no list overflowed
    assert bool((merged == whole[:, :K]).all())
    assert bool((totals == whole[:, K, 1]).all())
    merged, totals = merged.cpu().numpy(), totals.cpu().numpy()
    for qi in (0, 1, 2, 3, 1000, 2047, 3001, QBIG - 1):
        exp = _oracle_hits(ids, offs, srt, queries[qi], 2)
        top = sorted(exp, key=lambda h: (h[2], h[0], h[1]))[:K]
        top += [(-1, 0, tc.KTH_NEVER)] * (K - len(top))
        assert [tuple(int(x) for x in row) for row in merged[qi]] == top, qi
        assert int(totals[qi]) == len(exp), qi
