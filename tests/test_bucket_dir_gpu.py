"""The bucket directory of a one-sub-index handle (tvidz_amd/csrc/tvz_bucket_dir.h) on its EDGES, against the oracle's
restatement of db.find_duplicates (inspector/db.py:76-94).  Random corpora (tests/test_fuzz_gpu.py) rarely build
these: a bucket that more than eleven keys call home (the twelfth walks on to a neighbour and a lookup follows the
home bucket's `spill`), eleven keys whose lists are too short to be worth moving out (the last key walks on
instead), lists at the inline limit (48 postings stay in the bucket, 49 go to the external area), a list of
thousands, keys that hash to a crowded bucket but are in no row, and the row counts either side of one sub-index
(16,384 rows: buckets; 16,385: the open-addressing directory).  Keys are crafted with a numpy replica of the
library's hash for the bucket count the handle reports (tvz_corpus_bucket_stats)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tvidz_amd import _lib, corpus as tc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NEVER = tc.KTH_NEVER


def _bucket_of(values: np.ndarray, nb: int) -> np.ndarray:
    """bk_bucket(canonical key, nb) of tvz_bucket_dir.h, for positive float64 values."""
    k = np.asarray(values, dtype=np.float64).view(np.uint64)
    lo = (k & np.uint64(0xffffffff)).astype(np.uint64)
    hi = (k >> np.uint64(32)).astype(np.uint64)
    m = np.uint64(0xffffffff)
    x = (lo ^ ((hi + (hi << np.uint64(3))) & m) ^ (hi >> np.uint64(9))) & m          # q1_mix
    x ^= x >> np.uint64(20)
    h = (x * np.uint64(0x9E3779B1)) & m                                               # bk_hash
    return ((h * np.uint64(nb)) >> np.uint64(32)).astype(np.int64)


def _expected(rows, q, mm, excl=-1):
    ids, cnt, kth = oracle.match_kth(rows, list(q), mm)
    return sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(rows)) if cnt[c] >= mm and ids[c] != excl)


def _build(dc, base, nb_guess):
    """rows = base + crafted; the crafted keys are made for the bucket count the handle reports, so build, look, and
    build again until the count stands still."""
    pool = np.arange(1, 3_000_000, dtype=np.float64) / 128.0 + 50_000.0               # none of them in `base`
    nb = nb_guess
    for _ in range(6):
        b = _bucket_of(pool, nb)
        order = np.argsort(b, kind="stable")
        bs, first = np.unique(b[order], return_index=True)
        sizes = np.diff(np.append(first, len(order)))
        big = [i for i in np.argsort(-sizes) if (bs[i] % 256) not in (0, 255)][:4]    # (away from a slice's edges: room to walk)
        picks = [pool[order[first[i]:first[i] + sizes[i]]] for i in big]
        assert all(len(p) >= 26 for p in picks), [len(p) for p in picks]
        crowd, shorts, absent, mixed = picks[0][:24], picks[1][:11], picks[0][24:26], picks[2][:14]
        rows = list(base)
        vid = 1_000_000
        for i, kx in enumerate(crowd):                                               # 24 keys at home in ONE bucket, 1..3 postings each
            for rep in range(1 + i % 3):
                rows.append((vid, [float(kx), 7.5 + vid % 5, 900_000.25]))
                vid += 1
        for kx in shorts:                                                             # eleven keys of two postings: 132 bytes > payload
            for rep in range(2):
                rows.append((vid, [float(kx), 11.125]))
                vid += 1
        for i, kx in enumerate(mixed):                                               # a bucket of many keys with lists of 1..48
            for rep in range([1, 48, 49, 5, 20, 47, 2, 3, 33, 16, 8, 4, 12, 6][i]):
                rows.append((vid, [float(kx)]))
                vid += 1
        for rep in range(5_000):                                                      # one key in 5,000 rows: an external list
            rows.append((vid, [777_777.5, float(rep % 37) + 0.25]))
            vid += 1
        dc.upload(rows)
        st = dc.bucket_stats()
        assert st["sub_indexes"] == 1 and st["buckets"] > 0, st
        if st["buckets"] == nb:
            return rows, dict(crowd=crowd, shorts=shorts, absent=absent, mixed=mixed), st
        nb = st["buckets"]
    raise AssertionError("the bucket count did not settle")


def test_crowded_buckets_walks_external_lists_and_the_inline_limit():
    rng = np.random.default_rng(3)
    grid = np.arange(1, 40_001) / 8.0
    base = [(v + 1, rng.choice(grid, size=int(rng.integers(5, 40)), replace=False).tolist()) for v in range(2_500)]
    dc = tc.DeviceCorpus(0)
    try:
        dc.upload(base)
        rows, keys, st = _build(dc, base, dc.bucket_stats()["buckets"])
        assert st["keys_walked_on"] >= 13 and st["max_walk"] >= 1, st            # 24 keys cannot share a bucket of 11
        assert st["external_lists"] >= 2 and st["external_postings"] >= 5_000, st
        assert dc.index_stats()["indexed_rows"] == len(rows) <= 16_384
        queries = [keys["crowd"], keys["crowd"][::-1][:9], keys["shorts"], keys["mixed"],
                   np.concatenate([keys["absent"], keys["crowd"][:3], [900_000.25]]),       # absent keys of the crowded bucket
                   np.array([777_777.5, 0.25, 5.25]), np.array([777_777.5]), keys["absent"],
                   np.concatenate([keys["mixed"][1:3], keys["mixed"][1:3]]),                # 48 and 49 postings, with multiplicity
                   np.asarray(base[17][1] + [float(keys["crowd"][5])])]
        for mm in (1, 2, 3):
            exp = [_expected(rows, q, mm) for q in queries]
            # one query at a time (tvz_find_duplicates: the fused single-query lookup)
            for q, e in zip(queries, exp):
                assert sorted(dc.find_duplicates(q, mm, with_kth=True)) == e, (mm, q[:4])
            # batched: full hit lists, then every shape of the lookup that keeps the top-k
            d_q, d_off, ml = tc.pack_queries(queries, DEV)
            hits, n = dc.match(d_q, d_off, ml, mm, 8_192)
            hits, n = hits.cpu().numpy(), n.cpu().numpy()
            for qi, e in enumerate(exp):
                assert n[qi] == len(e) and sorted(tuple(int(x) for x in h) for h in hits[qi, :n[qi]]) == e, (mm, qi)
            for algo in (_lib.ALGO_NO_WAVE | _lib.ALGO_NO_PAIR, _lib.ALGO_PAIR, _lib.ALGO_WAVE):
                blk = dc.match_topk(d_q, d_off, ml, mm, 8_192, 16, algo=algo).cpu().numpy()
                for qi, e in enumerate(exp):
                    want = sorted(e, key=lambda h: (h[2], h[0], h[1]))[:16]
                    want += [(-1, 0, NEVER)] * (16 - len(want))
                    assert [tuple(int(x) for x in r) for r in blk[qi, :16]] == want and blk[qi, 16, 1] == len(e), (mm, algo, qi)
        # the same table after upserts (dead postings + a delta table) and after the rebuild they trigger
        for j in range(600):
            dc.upsert(rows[j][0], rows[j][1][::2] + [float(keys["crowd"][j % 24])])
            rows[j] = (rows[j][0], rows[j][1][::2] + [float(keys["crowd"][j % 24])])
        exp = [_expected(rows, q, 2) for q in queries]
        for q, e in zip(queries, exp):
            assert sorted(dc.find_duplicates(q, 2, with_kth=True)) == e
        d_q, d_off, ml = tc.pack_queries(queries, DEV)
        blk = dc.match_topk(d_q, d_off, ml, 2, 8_192, 16, algo=_lib.ALGO_PREFER_WAVE).cpu().numpy()
        for qi, e in enumerate(exp):
            want = sorted(e, key=lambda h: (h[2], h[0], h[1]))[:16]
            want += [(-1, 0, NEVER)] * (16 - len(want))
            assert [tuple(int(x) for x in r) for r in blk[qi, :16]] == want, qi
    finally:
        dc.close()


@pytest.mark.parametrize("n_rows", [1, 16_384, 16_385])
def test_either_side_of_one_sub_index(n_rows):
    """1 row and 16,384 rows: one sub-index, the bucket directory; 16,385 rows: two sub-indexes, the open-addressing
    directory.  The same answers as the oracle from all of them."""
    rng = np.random.default_rng(n_rows)
    grid = np.arange(1, 6_001) / 4.0
    rows = [(v + 1, rng.choice(grid, size=int(rng.integers(2, 9)), replace=False).tolist()) for v in range(n_rows)]
    dc = tc.DeviceCorpus(0)
    try:
        dc.upload(rows)
        st = dc.bucket_stats()
        if n_rows == 1:
            assert dc.index_stats()["indexed_rows"] in (0, 1)        # (a corpus this small may be swept without an index)
        else:
            assert (st["buckets"] > 0, st["sub_indexes"]) == ((True, 1) if n_rows <= 16_384 else (False, 2)), st
        queries = [np.asarray(rows[i % n_rows][1] + rows[(7 * i) % n_rows][1]) for i in range(12)] + [np.array([0.125, 99999.0])]
        d_q, d_off, ml = tc.pack_queries(queries, DEV)
        blk = dc.match_topk(d_q, d_off, ml, 2, 4_096, 8).cpu().numpy()
        for qi, q in enumerate(queries):
            e = _expected(rows, q, 2)
            assert sorted(dc.find_duplicates(q, 2, with_kth=True)) == e, qi
            want = sorted(e, key=lambda h: (h[2], h[0], h[1]))[:8]
            want += [(-1, 0, NEVER)] * (8 - len(want))
            assert [tuple(int(x) for x in r) for r in blk[qi, :8]] == want and blk[qi, 8, 1] == len(e), qi
    finally:
        dc.close()
