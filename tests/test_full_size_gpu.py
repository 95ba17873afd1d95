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
