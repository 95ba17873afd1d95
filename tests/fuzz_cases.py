"""Randomised differential cases: the HIP path (through the C ABI) against the oracle.

Shared by `tests/test_fuzz_gpu.py` (a seeded, bounded slice inside the driver-run `-m gpu` suite)
and `profiles/fuzz_parity.py` (the long soak).  Matcher: random corpora (sizes, row lengths,
duplicate rows and ids, special values), random batches, every min_match / exclusion / cap /
algorithm choice (index, Q1, tile, join, long queries), random upserts between the upload and the
match (replaced rows = stale postings + delta table, new rows, emptied rows, explicit rebuilds).
Top-k: per-shard block + merge of two gathered blocks.  Scene: random shapes, pitches, bit depths
and chunkings.  `run()` raises AssertionError at the first mismatch.

Round 4 points the differential at what round 3 built: BIG cases of 20k / 40k rows
of 2-5 keys (2-3 sub-indexes: key-major postings, per-sub-index uint16 counts, the fused top-k over
several sub-indexes) with 650-1400 upserts, half of them issued by a SECOND THREAD while this one
matches (the 512-row trigger fires: a background rebuild and its `since_snap` swap happen under
matches in flight; the concurrent upserts touch only keys no query of that phase holds, so every
answer stays decidable), and batches that hold a query of more than 4095 timestamps next to short
ones.  Counters: `multi_sub_cases`, `rebuilds_during_cases`, `concurrent_match_calls`,
`long_batch_cases`."""
import json
import threading
import time

import numpy as np
import torch

from oracle import oracle
from tvidz_amd import _lib, corpus as tc, scene

SPECIAL = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324, 1e300, 1.5, 0.1 + 0.2, 0.3])


def run(seconds: float = 60.0, seed: int = 12345, max_cases: int = 0, device: str = "cuda:0",
        progress: bool = False) -> dict:
    """Cases until `seconds` have passed or `max_cases` (> 0) ran; -> counters."""
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    dc = tc.DeviceCorpus(dev.index or 0)
    stats = {"match_cases": 0, "pairs": 0, "upserts": 0, "topk_cases": 0, "scene_cases": 0, "frames": 0,
             "long_batch_cases": 0, "multi_sub_cases": 0, "rebuilds_during_cases": 0, "concurrent_match_calls": 0}
    FAR = 1.0e7                                 # keys of the rows the second thread upserts: no query of that phase has them
    t_end = time.time() + seconds
    t_note = time.time() + 30

    def rand_keys(n, grid):
        k = np.round(rng.integers(1, grid, n) / 30.0, 4)
        m = rng.random(n) < 0.02
        k[m] = SPECIAL[rng.integers(0, len(SPECIAL), int(m.sum()))]
        return k

    try:
        while time.time() < t_end and (max_cases <= 0 or stats["match_cases"] < max_cases):
            # ---------------- matcher
            big = rng.random() < 0.22
            grid = int(rng.choice([50, 2000, 200000]))
            if big:
                # 2-3 sub-indexes of 16,384 rows, short rows (the oracle's db.py loop stays cheap)
                C = int(rng.choice([20000, 40000]))
                grid = int(rng.choice([2000, 200000]))
                lens = rng.integers(2, 6, size=C)
                flat = np.round(rng.integers(1, grid, int(lens.sum())) / 30.0, 4)
                cuts = np.concatenate([[0], np.cumsum(lens)])
                idv = rng.permutation(np.arange(1, C + 1))
                idv[rng.integers(0, C, C // 200)] = 7                      # duplicate ids (no UNIQUE constraint)
                rows = [(int(idv[c]), flat[cuts[c]:cuts[c + 1]].tolist()) for c in range(C)]
                n_park = 400                                               # rows the second thread may replace: far keys only
                for c in rng.choice(C, n_park, replace=False):
                    rows[c] = (int(3 * C + c), [FAR + c, FAR + c + 0.5])
                parked = [v for v, t in rows if t and t[0] >= FAR]
            else:
                C = int(rng.choice([1, 3, 50, 400, 3000]))
                rows = []
                for c in range(C):
                    L = int(rng.choice([0, 1, 2, 7, 40, 200, 700]))
                    rows.append((int(rng.integers(1, 10 * C + 2)), rand_keys(L, grid).tolist()))
                for _ in range(C // 10):
                    a, b = rng.integers(0, C, 2)
                    rows[b] = (rows[b][0], list(rows[a][1]))
            dc.upload(rows)
            builds0 = dc.index_stats()["builds"]

            def apply_upsert(v, ts):
                dc.upsert(v, ts)
                first = next((i for i, (vv, _) in enumerate(rows) if vv == v), None) if not big else first_of.get(v)
                if first is None:
                    rows.append((v, ts))
                    if big:
                        first_of[v] = len(rows) - 1
                else:
                    rows[first] = (v, ts)
                stats["upserts"] += 1

            if big:
                first_of = {}
                for i, (v, _) in enumerate(rows):
                    first_of.setdefault(v, i)
                # (a) a second thread replaces parked rows and adds new far-key rows - enough to cross the
                #     512-row trigger, so a background rebuild and its swap run - WHILE this thread matches
                n_conc = int(rng.choice([0, 650, 1400]))
                plan = []
                for j in range(n_conc):
                    v = int(parked[int(rng.integers(0, len(parked)))]) if rng.random() < 0.5 else int(5 * C + j)
                    plan.append((v, [FAR + 3 * j + 1.0, FAR + 3 * j + 1.5, FAR + 3 * j + 2.25][:int(rng.integers(1, 4))]))
                errs = []

                def upserter():
                    try:
                        for v, ts in plan:
                            dc.upsert(v, ts)
                    except BaseException as e:          # pragma: no cover
                        errs.append(e)
                th = threading.Thread(target=upserter)
                ids0, offs0, keys0 = tc.rows_to_csr(rows)
                cq = [rand_keys(int(rng.choice([5, 60, 250])), grid) for _ in range(6)]
                cq[1] = np.asarray(rows[int(rng.integers(0, C))][1] + rows[int(rng.integers(0, C))][1], dtype=np.float64)
                cq = [q[q < FAR] for q in cq]
                d_q, d_off, ml = tc.pack_queries(cq, dev)
                exp_c = []
                for q in cq:
                    cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs0, keys0, 2)
                    hit = np.flatnonzero(cnt >= 2)
                    exp_c.append(sorted((int(ids0[c]), int(cnt[c]), int(kth[c])) for c in hit))
                th.start()
                rounds = 0
                while th.is_alive() or rounds < 2:
                    hits, n = dc.match(d_q, d_off, ml, 2, 4096)
                    wsz = torch.empty(tc.workspace_bytes(len(cq), ml, 4096, 16, total_query_keys=d_q.numel()), dtype=torch.uint8, device=dev)
                    blk = dc.match_topk(d_q, d_off, ml, 2, 4096, 16, workspace=wsz, algo=int(rng.choice([0, _lib.ALGO_PAIR, _lib.ALGO_PREFER_WAVE])))
                    torch.cuda.synchronize()
                    hits, n, blk = hits.cpu().numpy(), n.cpu().numpy(), blk.cpu().numpy()
                    for qi in range(len(cq)):
                        got = sorted(map(tuple, hits[qi, :n[qi]].tolist()))
                        want = sorted(exp_c[qi], key=lambda h: (h[2], h[0], h[1]))[:16]
                        want += [(-1, 0, tc.KTH_NEVER)] * (16 - len(want))
                        ok = got == exp_c[qi] and [tuple(int(x) for x in r) for r in blk[qi, :16]] == want and \
                            int(blk[qi, 16, 1]) == len(exp_c[qi])
                        assert ok, ("CONCURRENT MATCH MISMATCH", dict(seed=seed, C=C, qi=qi, n=int(n[qi]),
                                                                       exp=len(exp_c[qi]), upserts=n_conc))
                    # find_duplicates too (the fused lookup + delta sweep, one launch)
                    got = dc.find_duplicates(cq[1], 2, with_kth=True)
                    assert got == exp_c[1], ("CONCURRENT FIND_DUPLICATES MISMATCH", dict(seed=seed, C=C))
                    stats["concurrent_match_calls"] += 1
                    rounds += 1
                th.join()
                assert not errs, errs
                for v, ts in plan:                      # the host's picture of what the second thread did
                    i = first_of.get(v)
                    if i is None:
                        rows.append((v, ts))
                        first_of[v] = len(rows) - 1
                    else:
                        rows[i] = (v, ts)
                    stats["upserts"] += 1
                n_seq = int(rng.choice([0, 5, 60]))
            else:
                n_seq = int(rng.choice([0, 0, 1, 5, 40]))
            # add_timestamps between the index build and the match (db.py:43-64: replaces the FIRST row)
            for _ in range(n_seq):
                kind = rng.random()
                if kind < 0.5 and rows:
                    v = rows[int(rng.integers(0, len(rows)))][0]
                else:
                    v = int(rng.integers(1, 10 * C + 50))
                ts = [] if rng.random() < 0.1 else rand_keys(int(rng.choice([1, 3, 30, 300])), grid).tolist()
                apply_upsert(v, ts)
            if rng.random() < 0.1:
                dc.build_index()
            C = len(rows)
            ids, offs, keys = tc.rows_to_csr(rows)
            st_ix = dc.index_stats()
            stats["rebuilds_during_cases"] += st_ix["builds"] - builds0
            if st_ix["indexed_rows"] > 16384:
                stats["multi_sub_cases"] += 1
            Q = int(rng.choice([2, 17, 40])) if big else int(rng.choice([1, 2, 17, 70, 300]))
            qlens = [0, 1, 5, 60, 250] if big else [0, 1, 5, 60, 250, 900]
            queries = [rand_keys(int(rng.choice(qlens)), grid) for _ in range(Q)]
            if C > 1 and Q > 1:
                queries[1] = np.asarray(rows[int(rng.integers(0, C))][1], dtype=np.float64)
            if Q > 2 and C <= 20000 and rng.random() < 0.25:
                # a query of more than 4095 timestamps INSIDE a batch (swept on its own, launch_match_with_long)
                queries[2] = np.concatenate([rand_keys(int(rng.integers(4096, 5200)), grid),
                                             np.asarray(rows[int(rng.integers(0, C))][1], dtype=np.float64)])
                stats["long_batch_cases"] += 1
            mm = int(rng.choice([-1, 0, 1, 2, 2, 2, 3, 5, 6, 9]))
            excl = None if rng.random() < 0.5 else [int(rng.integers(1, 10 * C + 2)) for _ in range(Q)]
            cap = int(rng.choice([1, 5, max(C, 1)]))
            mode = int(rng.choice([_lib.ALGO_AUTO, _lib.ALGO_AUTO, _lib.ALGO_INDEX, _lib.ALGO_Q1, _lib.ALGO_TILE,
                                   _lib.ALGO_JOIN]))
            if mode == _lib.ALGO_INDEX and mm < 1:
                mode = _lib.ALGO_AUTO              # the index answers min_match >= 1 only (an error otherwise)
            d_q, d_off, ml = tc.pack_queries(queries, dev)
            d_ex = torch.tensor(excl, dtype=torch.int32, device=dev) if excl is not None else None
            hits, n = dc.match(d_q, d_off, ml, mm, cap, d_exclude_ids=d_ex, algo=mode)
            torch.cuda.synchronize()
            hits, n = hits.cpu().numpy(), n.cpu().numpy()
            for qi, q in enumerate(queries):
                cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
                hit = np.flatnonzero((cnt >= mm) & ((ids != excl[qi]) if excl is not None else True))
                exp = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in hit)
                got = sorted(map(tuple, hits[qi, :min(n[qi], cap)].tolist()))
                ok = n[qi] == len(exp) and (set(got) <= set(exp) and len(got) == min(len(exp), cap))
                assert ok, ("MATCH MISMATCH", dict(seed=seed, C=C, Q=Q, mm=mm, cap=cap, mode=mode, qi=qi,
                                                   n=int(n[qi]), exp=len(exp)))
            # ---- top-k of the hit lists (one-wave kernel, flagged block fallback) + merge of per-shard blocks
            if mm >= 1 or rng.random() < 0.3:
                k = int(rng.choice([1, 8, 16, 64, 100]))
                capk = int(rng.choice([max(C, 1), 40, 2000]))
                ws = torch.empty(tc.workspace_bytes(Q, ml, capk, k, total_query_keys=d_q.numel()), dtype=torch.uint8, device=dev)
                # the fused lookup's shape - two queries per block, one, or the library's choice - never changes a result
                # (on a handle of one sub-index PREFER_WAVE takes the one-wave-per-query kernel, the others the block kernel)
                shape = int(rng.choice([0, _lib.ALGO_PAIR, _lib.ALGO_NO_PAIR, _lib.ALGO_PREFER_WAVE, _lib.ALGO_PREFER_WAVE | _lib.ALGO_NO_PAIR]))
                blk = dc.match_topk(d_q, d_off, ml, mm, capk, k, d_exclude_ids=d_ex, workspace=ws, algo=mode | shape)
                merged, totals = tc.topk_merge(torch.stack([blk, blk]).contiguous(), k)   # two identical "ranks"
                torch.cuda.synchronize()
                blk, merged, totals = blk.cpu().numpy(), merged.cpu().numpy(), totals.cpu().numpy()
                for qi, q in enumerate(queries):
                    cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
                    hit = np.flatnonzero((cnt >= mm) & ((ids != excl[qi]) if excl is not None else True))
                    exp = [(int(ids[c]), int(cnt[c]), int(kth[c])) for c in hit]
                    tot = int(blk[qi, k, 1])
                    ok = tuple(blk[qi, k][[0, 2]]) == (-1, tc.KTH_NEVER)
                    if len(exp) <= capk:
                        srt = sorted(exp, key=lambda h: (h[2], h[0], h[1]))
                        want = srt[:k] + [(-1, 0, tc.KTH_NEVER)] * (k - min(k, len(srt)))
                        ok = ok and tot == len(exp) and [tuple(int(x) for x in r) for r in blk[qi, :k]] == want
                        dbl = sorted(exp + exp, key=lambda h: (h[2], h[0], h[1]))[:k]
                        dbl += [(-1, 0, tc.KTH_NEVER)] * (k - len(dbl))
                        ok = ok and [tuple(int(x) for x in r) for r in merged[qi]] == dbl and \
                            int(totals[qi]) == 2 * len(exp)
                    else:                           # truncated list: signalled, entries are real hits
                        got = [tuple(int(x) for x in r) for r in blk[qi, :k] if r[0] >= 0]
                        ok = ok and tot == -len(exp) and set(got) <= set(exp) and int(totals[qi]) == -2 * len(exp)
                    assert ok, ("TOPK MISMATCH", dict(seed=seed, C=C, Q=Q, mm=mm, k=k, cap=capk, mode=mode, qi=qi,
                                                      n=len(exp)))
                stats["topk_cases"] += 1
            qi = int(rng.integers(0, Q))
            longq = rand_keys(int(rng.choice([10, 4500])), grid)
            for q in (queries[qi], longq):
                got = dc.find_duplicates(q, mm, with_kth=True)
                cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
                exp = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in np.flatnonzero(cnt >= mm))
                assert got == exp, ("FIND_DUPLICATES MISMATCH", dict(seed=seed, C=C, mm=mm, n=len(q)))
            stats["match_cases"] += 1
            stats["pairs"] += C * Q
            # ---------------- scene
            H, W, T = int(rng.integers(1, 120)), int(rng.integers(1, 200)), int(rng.integers(1, 150))
            if rng.random() < 0.3:
                H, W = int(rng.integers(1, 30)) * 4, int(rng.integers(1, 30)) * 16
            bd = int(rng.choice([8, 8, 10, 16]))
            dt = np.uint8 if bd == 8 else np.uint16
            ph, pw = int(rng.integers(0, 4)), int(rng.integers(0, 9))
            big = rng.integers(0, 1 << bd, size=(T, H + ph, W + pw)).astype(dt)
            big[T // 3:] = (big[T // 3:] // 4).astype(dt)
            v = big[:, :H, :W]
            d = torch.from_numpy(big.view(np.int16) if bd > 8 else big).to(dev)[:, :H, :W]
            step = int(rng.integers(1, T + 1))
            sc = scene.SceneScorer(H, W, step, dev, 0.3, bitdepth=bd)
            sels, sads = [], []
            for s0 in range(0, T, step):
                part = d[s0:s0 + step]
                sad, _, _, sel = sc.score_batch(part)      # the carried state stays on the device
                sads.append(sad.cpu().numpy().view(np.uint64).copy())
                sels.append(sel.cpu().numpy().copy())
            o_sad = oracle.luma_sad(v)
            o_sel, _, _, _ = oracle.scene_select(o_sad, H, W, 0.3, bitdepth=bd)
            assert (np.concatenate(sads) == o_sad).all() and (np.concatenate(sels) == o_sel).all(), \
                ("SCENE MISMATCH", dict(seed=seed, H=H, W=W, T=T, bd=bd, step=step, ph=ph, pw=pw))
            stats["scene_cases"] += 1
            stats["frames"] += T
            if progress and time.time() > t_note:      # keep long runs visibly alive
                print(json.dumps({"progress": stats}), flush=True)
                t_note = time.time() + 30
    finally:
        dc.close()
    return stats
