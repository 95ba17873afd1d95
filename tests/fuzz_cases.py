"""Randomised differential cases: the HIP path (through the C ABI) against the oracle.

Shared by `tests/test_fuzz_gpu.py` (a seeded, bounded slice inside the driver-run `-m gpu` suite)
and `profiles/fuzz_parity.py` (the long soak).  Matcher: random corpora (sizes, row lengths,
duplicate rows and ids, special values), random batches, every min_match / exclusion / cap /
algorithm choice (index, Q1, tile, join, long queries), random upserts between the upload and the
match (replaced rows = stale postings + delta table, new rows, emptied rows, explicit rebuilds).
Top-k: per-shard block + merge of two gathered blocks.  Scene: random shapes, pitches, bit depths
and chunkings.  `run()` raises AssertionError at the first mismatch."""
import json
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
             "long_batch_cases": 0}
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
            C = int(rng.choice([1, 3, 50, 400, 3000]))
            grid = int(rng.choice([50, 2000, 200000]))
            rows = []
            for c in range(C):
                L = int(rng.choice([0, 1, 2, 7, 40, 200, 700]))
                rows.append((int(rng.integers(1, 10 * C + 2)), rand_keys(L, grid).tolist()))
            for _ in range(C // 10):
                a, b = rng.integers(0, C, 2)
                rows[b] = (rows[b][0], list(rows[a][1]))
            dc.upload(rows)
            # add_timestamps between the index build and the match (db.py:43-64: replaces the FIRST row)
            for _ in range(int(rng.choice([0, 0, 1, 5, 40]))):
                kind = rng.random()
                if kind < 0.5 and rows:
                    v = rows[int(rng.integers(0, len(rows)))][0]
                else:
                    v = int(rng.integers(1, 10 * C + 50))
                ts = [] if rng.random() < 0.1 else rand_keys(int(rng.choice([1, 3, 30, 300])), grid).tolist()
                dc.upsert(v, ts)
                first = next((i for i, (vv, _) in enumerate(rows) if vv == v), None)
                if first is None:
                    rows.append((v, ts))
                else:
                    rows[first] = (v, ts)
                stats["upserts"] += 1
            if rng.random() < 0.1:
                dc.build_index()
            C = len(rows)
            ids, offs, keys = tc.rows_to_csr(rows)
            Q = int(rng.choice([1, 2, 17, 70, 300]))
            queries = [rand_keys(int(rng.choice([0, 1, 5, 60, 250, 900])), grid) for _ in range(Q)]
            if C > 1 and Q > 1:
                queries[1] = np.asarray(rows[int(rng.integers(0, C))][1], dtype=np.float64)
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
                exp = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(C)
                             if cnt[c] >= mm and (excl is None or ids[c] != excl[qi]))
                got = sorted(map(tuple, hits[qi, :min(n[qi], cap)].tolist()))
                ok = n[qi] == len(exp) and (set(got) <= set(exp) and len(got) == min(len(exp), cap))
                assert ok, ("MATCH MISMATCH", dict(seed=seed, C=C, Q=Q, mm=mm, cap=cap, mode=mode, qi=qi,
                                                   n=int(n[qi]), exp=len(exp)))
            # ---- top-k of the hit lists (one-wave kernel, flagged block fallback) + merge of per-shard blocks
            if mm >= 1 or rng.random() < 0.3:
                k = int(rng.choice([1, 8, 16, 64, 100]))
                capk = int(rng.choice([max(C, 1), 40, 2000]))
                ws = torch.empty(tc.workspace_bytes(Q, ml, capk, k), dtype=torch.uint8, device=dev)
                blk = dc.match_topk(d_q, d_off, ml, mm, capk, k, d_exclude_ids=d_ex, workspace=ws, algo=mode)
                merged, totals = tc.topk_merge(torch.stack([blk, blk]).contiguous(), k)   # two identical "ranks"
                torch.cuda.synchronize()
                blk, merged, totals = blk.cpu().numpy(), merged.cpu().numpy(), totals.cpu().numpy()
                for qi, q in enumerate(queries):
                    cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
                    exp = [(int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(C)
                           if cnt[c] >= mm and (excl is None or ids[c] != excl[qi])]
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
                exp = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(C) if cnt[c] >= mm)
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
