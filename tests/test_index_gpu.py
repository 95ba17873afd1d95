"""GPU parity of the inverted index (tvz_index_kernels.h) against the oracle's restatement of
db.find_duplicates (inspector/db.py:76-94) and against the sweep kernels: lookup + delta sweep must
return exactly what a full sweep returns - bit-exact (video_id, count, kth) triples - for every
min_match the index serves (>= 1), through upserts that replace indexed rows, new rows, automatic
rebuilds, corpora larger than the candidate bitmap, and queries with more candidates than the LDS
table holds."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tvidz_amd import _lib, corpus as tc, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def dc():
    c = tc.DeviceCorpus(0)
    yield c
    c.close()


def _expected(ids, offs, keys, q, mm, excl=None):
    cnt, kth = oracle.match_kth_csr(np.asarray(q, dtype=np.float64), offs, keys, mm)
    return sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(ids))
                  if cnt[c] >= mm and (excl is None or ids[c] != excl))


def _check(dc, rows, queries, mm, excl=None, cap=None, algo=_lib.ALGO_INDEX):
    ids, offs, keys = tc.rows_to_csr(rows)
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    cap = cap or max(len(ids), 1)
    d_ex = torch.tensor(excl, dtype=torch.int32, device=DEV) if excl is not None else None
    hits, n = dc.match(d_q, d_off, max_len, mm, cap, d_exclude_ids=d_ex, algo=algo)
    torch.cuda.synchronize()
    hits, n = hits.cpu().numpy(), n.cpu().numpy()
    for qi, q in enumerate(queries):
        exp = _expected(ids, offs, keys, q, mm, None if excl is None else excl[qi])
        assert n[qi] == len(exp), (qi, int(n[qi]), len(exp))
        got = sorted(tuple(int(x) for x in h) for h in hits[qi, :min(n[qi], cap)])
        if n[qi] > cap:
            assert len(got) == cap and set(got) <= set(exp)
        else:
            assert got == exp, qi
    return n


def _check_single(dc, rows, q, mm, excl=-1):
    ids, offs, keys = tc.rows_to_csr(rows)
    got = dc.find_duplicates(q, mm, exclude_id=excl, with_kth=True)
    assert got == _expected(ids, offs, keys, q, mm, excl if excl >= 0 else None)


@pytest.mark.parametrize("C,mean_len,Q", [(300, 40, 9), (3000, 200, 24), (64, 12, 7)])
def test_index_equals_oracle_and_sweeps(dc, C, mean_len, Q):
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=C, mean_len=mean_len, dup_frac=0.05, frag_frac=0.05)
    rows = [(int(ids[r]), keys[offs[r]:offs[r + 1]].tolist()) for r in range(C)]
    dc.upload(rows)
    st = dc.index_stats()
    assert st["indexed_rows"] == C and st["delta_rows"] == 0 and st["builds"] == 1
    assert st["postings"] == dc.stats()[1] and 0 < st["distinct_keys"] <= st["postings"]
    queries = synth.synth_queries(ids, offs, keys, Q, seed=Q, mean_len=mean_len)
    queries[0] = np.concatenate([queries[0], queries[0][:7]])        # multiplicity
    queries[1] = np.array([float("nan"), -0.0, 0.0] + queries[1][:5].tolist())
    queries[2] = np.zeros(0)                                         # empty query
    excl = [int(ids[(7 * i) % C]) for i in range(Q)]
    for mm in (1, 2, 3, 5):
        _check(dc, rows, queries, mm)
        _check(dc, rows, queries, mm, excl=excl)
        _check(dc, rows, queries, mm, excl=excl, algo=_lib.ALGO_AUTO)
        _check(dc, rows, queries, mm, cap=3)                          # overflow keeps the true counts
        for q in queries[:4]:
            _check_single(dc, rows, q, mm)
            _check_single(dc, rows, q, mm, excl=excl[0])
    with pytest.raises(RuntimeError, match="min_match >= 1"):        # every row is a hit: a sweep's job
        _check(dc, rows, queries, 0)
    _check(dc, rows, queries, 0, algo=_lib.ALGO_AUTO)
    for mm in (6, 9, 40):                                             # count-only pass B + the kth fix-up walk
        _check(dc, rows, queries, mm)
        _check(dc, rows, queries, mm, excl=excl, algo=_lib.ALGO_AUTO)
        _check(dc, rows, queries, mm, cap=3)
        for q in queries[:4]:                                         # tvz_find_duplicates takes the same path (a batch of one)
            _check_single(dc, rows, q, mm)
            _check_single(dc, rows, q, mm, excl=excl[0])


def test_more_candidates_than_the_lds_table_holds(dc):
    """A small alphabet: thousands of rows share >= 2 keys with a query (parts > 1), one key is in
    half of the rows (a posting list of thousands), and a query made of that key alone."""
    C = 9000
    rng = np.random.default_rng(3)
    grid = np.arange(1, 1501) / 8.0
    rows = []
    for c in range(C):
        r = rng.choice(grid, size=int(rng.integers(5, 40)), replace=False)
        if rng.random() < 0.5:
            r = np.append(r, 777.125)
        rows.append((c + 1, r.tolist()))
    dc.upload(rows)
    queries = [rng.choice(grid, size=n, replace=False) for n in (20, 160, 600, 1400)]
    queries.append(np.array([777.125] * 3))
    queries.append(np.concatenate([grid[:300], grid[:300]]))
    for mm in (1, 2, 5):
        n = _check(dc, rows, queries, mm)
        assert n.max() > 4000                                         # well beyond 1536 table entries
    _check_single(dc, rows, queries[3], 2)
    _check_single(dc, rows, queries[4], 1)


@pytest.mark.parametrize("C", [150_000, 300_000])
def test_corpus_larger_than_the_candidate_bitmap(dc, C):
    """> 2^17 rows: rows share bits of the seen-once / seen-twice bitmaps; counts stay exact.
    (10 and 19 sub-indexes: directory entries of 48 and 64 bytes, slices of 512 entries.)"""
    rng = np.random.default_rng(5)
    alphabet = np.arange(1, 40_001) / 4.0
    lens = rng.integers(2, 6, size=C)
    offs = np.zeros(C + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    keys = rng.choice(alphabet, size=int(offs[-1]))
    ids = np.arange(1, C + 1, dtype=np.int32)
    dc.upload_csr(ids, offs, keys)
    ids2, offs2, keys2 = ids, offs, keys                              # rows may repeat a key: still a set
    queries = [rng.choice(alphabet, size=n, replace=False) for n in (50, 400, 3000)]
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    cap = 60000 * (C // 150_000)
    for mm in (1, 2, 3):
        hits, n = dc.match(d_q, d_off, max_len, mm, cap, algo=_lib.ALGO_INDEX)
        torch.cuda.synchronize()
        hits, n = hits.cpu().numpy(), n.cpu().numpy()
        for qi, q in enumerate(queries):
            exp = _expected(ids2, offs2, keys2, q, mm)
            assert n[qi] == len(exp) <= cap
            assert sorted(tuple(int(x) for x in h) for h in hits[qi, :n[qi]]) == exp


def test_build_with_every_posting_size_class_and_a_first_directory_far_too_small(dc):
    """The partitioned build: (a) nearly all keys are distinct, so the first directory (sized from
    the key count) is crowded, slices overflow and the build repeats with twice the directory until
    it fits; (b) key j = 1..130 is in exactly j rows - every posting size class (1, 2, 4 .. 32,
    whole lines) with its boundaries, min_match 1 makes every single posting count; (c) one key is
    in every row (a list of several lines in every slice-local layout)."""
    C = 3000
    rng = np.random.default_rng(11)
    rows = []
    nxt = 1000.0
    for c in range(C):
        own = nxt + np.arange(100) * 0.5                                # 100 keys nobody else has
        nxt += 100.0
        r = list(own) + [0.25]                                          # (c): in every row
        r += [float(j) for j in range(1, 131) if c < j]                 # (b): key j in rows 0 .. j-1
        rows.append((c + 1, r))
    dc.upload(rows)
    st = dc.index_stats()
    assert st["indexed_rows"] == C and st["distinct_keys"] == C * 100 + 1 + 130
    queries = [np.arange(1, 131, dtype=np.float64),                     # every class at once
               np.array([0.25, 64.0, 65.0, 33.0, 32.0]),
               np.array(rows[7][1][:100]),                              # a row's own keys
               np.array([1.0, 2.0, 3.0, 4.0, 5.0, 8.0, 9.0, 16.0, 17.0])]
    for mm in (1, 2, 3):
        _check(dc, rows, queries, mm)
    _check_single(dc, rows, queries[0], 1)
    _check_single(dc, rows, queries[1], 2)


def test_upserts_replace_indexed_rows_and_add_new_ones(dc):
    C = 1200
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=11, mean_len=60, dup_frac=0.05)
    rows = [(int(ids[r]), keys[offs[r]:offs[r + 1]].tolist()) for r in range(C)]
    dc.upload(rows)
    rng = np.random.default_rng(12)
    queries = synth.synth_queries(ids, offs, keys, 12, seed=13, mean_len=60)
    by_id = {v: i for i, (v, _) in enumerate(rows)}
    new_id = int(ids.max()) + 1
    for step in range(40):
        kind = step % 4
        if kind == 0:        # an indexed row gets the content of a query: its old postings are stale
            v = rows[int(rng.integers(0, C))][0]
            ts = queries[step % len(queries)].tolist()
        elif kind == 1:      # a brand-new video: lives in the delta table only
            v, new_id = new_id, new_id + 1
            ts = np.round(rng.uniform(0, 3000, 50), 2).tolist() + queries[0][:9].tolist()
        elif kind == 2:      # the same row again (already in the delta table): in-place swap there
            v = rows[by_id[v]][0]
            ts = queries[(step + 1) % len(queries)][:20].tolist()
        else:                # a row emptied
            v = rows[int(rng.integers(0, len(rows)))][0]
            ts = []
        dc.upsert(v, ts)
        if v in by_id:
            rows[by_id[v]] = (v, ts)
        else:
            by_id[v] = len(rows)
            rows.append((v, ts))
        if step % 5 == 4:
            _check(dc, rows, queries, 2)
            _check(dc, rows, queries, 2, algo=_lib.ALGO_TILE)          # the full sweep agrees
            _check_single(dc, rows, queries[step % len(queries)], 2, excl=rows[3][0])
    st = dc.index_stats()
    assert st["indexed_rows"] == C and 0 < st["delta_rows"] <= 40 and st["builds"] == 1
    _check(dc, rows, queries, 1)
    _check(dc, rows, queries, 5, excl=[rows[i][0] for i in range(len(queries))])
    dc.build_index()
    st = dc.index_stats()
    assert st["indexed_rows"] == len(rows) and st["delta_rows"] == 0 and st["builds"] == 2
    _check(dc, rows, queries, 2)
    _check_single(dc, rows, queries[0], 2)
    # clear: no index, no rows; the next rows are swept until there are enough for an index
    dc.clear()
    assert dc.index_stats()["indexed_rows"] == 0
    assert dc.find_duplicates(queries[0].tolist(), 1) == []
    dc.upsert(5, queries[0].tolist())
    assert dc.find_duplicates(queries[0].tolist(), 2) == [(5, len(queries[0]))]
    with pytest.raises(RuntimeError, match="no index"):
        _check(dc, [(5, queries[0].tolist())], queries[:2], 2)


def test_a_corpus_grown_by_upserts_gets_and_refreshes_its_index(dc):
    """add_timestamps only (db.py:43-64), as the service does: first index at 4096 rows, rebuilt
    whenever the delta table has grown to max(512, rows / 256) entries; matches are right at every
    stage."""
    rng = np.random.default_rng(21)
    grid = np.arange(1, 30_001) / 10.0
    rows = []
    probe = rng.choice(grid, size=30, replace=False)
    builds = []
    for v in range(1, 8500):
        ts = rng.choice(grid, size=12, replace=False).tolist()
        if v % 500 == 0:
            ts += probe[:10].tolist()
        dc.upsert(v, ts)
        rows.append((v, ts))
        if v in (100, 4095, 4096, 4097, 4500, 4607, 4608, 4609, 6000, 8191, 8193, 8499):
            builds.append((v, dc.index_stats()))
            _check_single(dc, rows, probe, 2)
            if v in (4097, 4609, 8499):
                _check(dc, rows, [probe, np.asarray(rows[10][1])], 2, algo=_lib.ALGO_AUTO)
    st = dict(builds)
    assert st[100]["builds"] == 0 and st[4095]["builds"] == 0
    assert st[4096]["builds"] == 1 and st[4096]["indexed_rows"] == 4096 and st[4096]["delta_rows"] == 0
    assert st[4097]["delta_rows"] == 1 and st[4500]["delta_rows"] == 4500 - 4096 and st[4607]["builds"] == 1
    assert st[4608]["builds"] == 2 and st[4608]["indexed_rows"] == 4608 and st[4608]["delta_rows"] == 0
    assert st[4609]["delta_rows"] == 1
    assert st[6000]["builds"] == 4 and st[6000]["indexed_rows"] == 4096 + 3 * 512
    assert st[8499]["builds"] == 9 and st[8499]["indexed_rows"] == 4096 + 8 * 512 and st[8499]["delta_rows"] == 8499 - 8192


def test_lookups_stay_exact_while_upserts_rebuild_the_index(dc):
    """Threads call find_duplicates (index lookup + delta sweep, host in / host out) and tvz_match
    while another thread upserts thousands of unrelated videos, which fills the delta table and
    rebuilds the index several times under them (a rebuild waits for the matches in flight and
    replaces the directories, postings and video-id table).  The probes' hits never change."""
    import threading
    rng = np.random.default_rng(31)
    grid = np.arange(1, 20_001) / 10.0
    probe = rng.choice(grid, size=40, replace=False)
    rows = [(v, rng.choice(grid, size=10, replace=False).tolist()) for v in range(1, 5001)]
    for v in (17, 1234, 4999):
        rows[v - 1] = (v, rows[v - 1][1][:6] + probe[:12].tolist())       # the probe's duplicates
    dc.upload(rows)
    ids, offs, keys = tc.rows_to_csr(rows)
    base = _expected(ids, offs, keys, probe, 5)
    assert [h[0] for h in base] == [17, 1234, 4999]
    stop = threading.Event()
    errs = []

    def finder():
        try:
            while not stop.is_set():
                got = dc.find_duplicates(probe, 5, with_kth=True)
                assert got == base, got
        except Exception as e:                                            # pragma: no cover
            errs.append(e)

    def batcher():
        try:
            d_q, d_off, max_len = tc.pack_queries([probe] * 8, DEV)
            while not stop.is_set():
                hits, n = dc.match(d_q, d_off, max_len, 5, 64)
                torch.cuda.synchronize()
                h, nn = hits.cpu().numpy(), n.cpu().numpy()
                for qi in range(8):
                    assert sorted(tuple(int(x) for x in r) for r in h[qi, :nn[qi]]) == base
        except Exception as e:                                            # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=finder) for _ in range(3)] + [threading.Thread(target=batcher)]
    [t.start() for t in th]
    try:
        # far from the probe's keys (> 2000.0), so none of them can become a hit
        for v in range(10_000, 10_000 + 9500):
            dc.upsert(v, (2500.0 + rng.integers(0, 100_000, size=8) / 7.0).tolist())
    finally:
        stop.set()
        [t.join(60) for t in th]
    assert not errs, errs[:1]
    st = dc.index_stats()
    assert st["builds"] >= 3 and st["indexed_rows"] > 5000
    assert dc.find_duplicates(probe, 5, with_kth=True) == base


def test_index_through_compaction_growth_long_rows_and_duplicate_ids(dc):
    """The slow paths of add_timestamps with an index in place: the arena is garbage-collected
    (every row's key offset moves), the row table and arena grow beyond their reservation, a row is
    longer than a pinned ring slot (> 8192 cuts: synchronous copy), and two rows carry the same
    video_id (the reference has no UNIQUE constraint, db.py:21-27: both are matched, the upsert
    replaces the first)."""
    rng = np.random.default_rng(41)
    grid = np.arange(1, 50_001) / 20.0
    rows = [(v, rng.choice(grid, size=int(rng.integers(3, 30)), replace=False).tolist()) for v in range(1, 301)]
    rows.append((7, rng.choice(grid, size=12, replace=False).tolist()))          # second row of video 7
    dc.upload(rows)
    q = np.asarray(rows[6][1][:8] + rows[-1][1][:8] + rows[100][1][:5])
    _check(dc, rows, [q], 2)
    _check_single(dc, rows, q, 2)
    # replace the same few rows over and over: dead arena space piles up until a compaction
    _, _, arena0 = dc.stats()
    compacted = False
    for it in range(400):
        v = 1 + (it % 5)
        ts = rng.choice(grid, size=500, replace=False).tolist()
        dc.upsert(v, ts)
        rows[v - 1] = (v, ts)
        _, live, arena = dc.stats()
        compacted = compacted or arena < arena0
        arena0 = arena
    assert compacted
    st = dc.index_stats()
    assert st["builds"] >= 2 and st["delta_rows"] <= 5          # the compaction rebuilt the index
    q2 = np.asarray(rows[2][1][:20] + rows[200][1][:6])
    for mm in (1, 2, 5):
        _check(dc, rows, [q, q2], mm)
        _check(dc, rows, [q, q2], mm, algo=_lib.ALGO_TILE)
    _check_single(dc, rows, q2, 2, excl=3)
    # a row of 9000 cuts (longer than a ring slot) and growth past the reservation
    long_ts = (np.arange(9000) * 0.25 + 5000.0).tolist()
    dc.upsert(9001, long_ts)
    rows.append((9001, long_ts))
    for v in range(20_000, 20_000 + 3000):                       # default reservation: 64 Ki rows, 2 Mi keys
        ts = rng.choice(grid, size=800, replace=False).tolist()
        dc.upsert(v, ts)
        rows.append((v, ts))
    q3 = np.asarray(long_ts[100:130] + rows[-1][1][:10])
    _check(dc, rows, [q2, q3], 2)
    _check_single(dc, rows, q3, 5)
    # upsert of a duplicated id replaces its FIRST row only (db.py:54-62 .first())
    dc.upsert(7, rows[-1][1][:15])
    rows[6] = (7, rows[-1][1][:15])
    _check(dc, rows, [q, q3], 2)
    dc.build_index()
    _check(dc, rows, [q, q3], 2)
    _check_single(dc, rows, q3, 2, excl=7)


def test_replacing_an_indexed_row_that_is_a_hit_never_hides_it(dc):
    """add_timestamps (db.py:54-62) on a row that a concurrent find_duplicates is matching: the reader
    sees the old or the new list (both hold the probe's cuts here), never neither.  The first
    replacement of an INDEXED row marks its postings dead and gives it a new delta entry; a match in
    flight captured the old delta size, so it must still find the row through the postings - the dead
    mark therefore waits (on the device) for the matches enqueued before it."""
    import threading
    rng = np.random.default_rng(51)
    grid = np.arange(1, 20_001) / 10.0
    probe = rng.choice(grid, size=30, replace=False)
    rows = [(v, rng.choice(grid, size=12, replace=False).tolist()) for v in range(1, 6001)]
    X = 4321
    variants = [probe[:14].tolist() + rows[X - 1][1][:5], probe[:12].tolist() + [5000.5, 5001.5]]
    rows[X - 1] = (X, variants[0])
    dc.upload(rows)
    stop = threading.Event()
    errs = []
    seen = {"find": 0, "batch": 0}

    def finder():
        try:
            while not stop.is_set():
                got = dc.find_duplicates(probe, 5)
                assert [h for h in got if h[0] == X] in ([(X, 14)], [(X, 12)]), got
                seen["find"] += 1
        except Exception as e:                                            # pragma: no cover
            errs.append(e)

    def batcher():
        try:
            d_q, d_off, max_len = tc.pack_queries([probe] * 16, DEV)
            while not stop.is_set():
                hits, n = dc.match(d_q, d_off, max_len, 5, 64)
                torch.cuda.synchronize()
                h, nn = hits.cpu().numpy(), n.cpu().numpy()
                for qi in range(16):
                    mine = [tuple(int(x) for x in r[:2]) for r in h[qi, :nn[qi]] if r[0] == X]
                    assert mine in ([(X, 14)], [(X, 12)]), (qi, mine)
                seen["batch"] += 1
        except Exception as e:                                            # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=finder) for _ in range(2)] + [threading.Thread(target=batcher)]
    [t.start() for t in th]
    try:
        for it in range(300):
            dc.upsert(X, variants[(it + 1) % 2])          # (first one after a build: kills the row's postings)
            if it % 3 == 2:
                dc.build_index()                          # the row is indexed again
    finally:
        stop.set()
        [t.join(60) for t in th]
    assert not errs, errs[:1]
    assert seen["find"] > 50 and seen["batch"] > 10, seen


def test_clear_does_not_disturb_matches_already_queued(dc):
    """/admin/clear-db (app.py:325-333) while matches are queued: they keep sweeping the rows they
    were launched with.  The arena is reused from offset 0 by the next add_timestamps, so that copy
    is ordered (on the device) behind every match enqueued before the clear."""
    ids, offs, keys = synth.synth_timestamp_corpus(20000, seed=77, mean_len=200)
    dc.upload_csr(ids, offs, keys)
    queries = [keys[offs[c]:offs[c + 1]].copy() for c in (0, 1, 2, 3)] * 64     # rows 0..3 are hits of their queries
    d_q, d_off, max_len = tc.pack_queries(queries, DEV)
    st = torch.cuda.Stream(DEV)
    outs = []
    for _ in range(6):                                    # ~ms of corpus sweeps queued on one stream, not waited for
        outs.append(dc.match(d_q, d_off, max_len, 5, 64, stream=st, algo=_lib.ALGO_TILE))
    dc.clear()
    dc.upsert(999_999, (np.arange(4000) * 0.5 + 9000.0).tolist())               # lands at arena offset 0
    st.synchronize()
    torch.cuda.synchronize()
    for hits, n in outs:
        h, nn = hits.cpu().numpy(), n.cpu().numpy()
        for qi in range(len(queries)):
            c = qi % 4
            got = sorted(tuple(int(x) for x in r) for r in h[qi, :nn[qi]])
            assert (int(ids[c]), int(offs[c + 1] - offs[c]), 4) in got, (qi, got[:3])
    assert dc.find_duplicates(queries[0], 2) == []
    assert dc.find_duplicates([9000.0, 9000.5, 9001.0], 3) == [(999_999, 3)]
