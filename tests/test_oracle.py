"""CPU tests: pin the oracle against the reference's known answers and golden fixtures.

The fixtures under tests/golden/match_*.json were produced by executing the reference's own
inspector/db.py find_duplicates (oracle/gen_golden.py).  The scene-score half has no reference
fixture (PARITY UNPINNED, see oracle/tvz_oracle.c): its tests only check internal consistency
(C vs numpy restatement, chunking, hand-computed cases).
"""
import json
import os

import numpy as np

from oracle import oracle


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _rows(corpus):
    return [(int(v), [float(x) for x in t]) for v, t in corpus]


def test_reference_kat_test_app_66_83():
    # /root/reference/inspector/test_app.py:66-83, ids 1,2,3 in insertion order
    corpus = [(1, [1.0, 2.0, 3.0, 4.0, 5.0]), (2, [10.0, 20.0, 30.0, 40.0, 50.0])]
    for fd in (oracle.find_duplicates_py, oracle.find_duplicates_c):
        dups = fd(corpus, [10.0, 20.0, 30.0, 40.0, 50.0], min_match=5)
        assert (1, 0) not in dups and (2, 5) in dups
    corpus.append((3, [1.0, 2.0, 3.0, 4.0, 5.0]))
    for fd in (oracle.find_duplicates_py, oracle.find_duplicates_c):
        dups = fd(corpus, [1.0, 2.0, 3.0, 4.0, 5.0], min_match=5)
        assert (1, 5) in dups and (3, 5) in dups and len(dups) == 2


def test_golden_kat(golden_dir):
    g = _load(golden_dir, "match_kat.json")
    for case in g["cases"]:
        corpus = _rows(case["corpus"])
        exp = [tuple(e) for e in case["expected"]]
        for fd in (oracle.find_duplicates_py, oracle.find_duplicates_c):
            got = sorted(fd(corpus, case["query"], case["min_match"]))
            assert got == exp, case["name"]
    nc = g["nan_case"]
    corpus = [(v, [float("nan") if x is None else x for x in t]) for v, t in nc["corpus"]]
    query = [float("nan") if x is None else x for x in nc["query"]]
    for fd in (oracle.find_duplicates_py, oracle.find_duplicates_c):
        assert sorted(fd(corpus, query, nc["min_match"])) == [tuple(e) for e in nc["expected"]]


def test_golden_random(golden_dir):
    g = _load(golden_dir, "match_random.json")
    corpora = {k: _rows(v) for k, v in g["corpora"].items()}
    for case in g["cases"]:
        corpus = corpora[str(case["corpus_ref"])]
        exp = [tuple(e) for e in case["expected"]]
        assert sorted(oracle.find_duplicates_c(corpus, case["query"], case["min_match"])) == exp
    # the pure-Python restatement on a subset (it is slow by construction)
    for case in g["cases"][:12]:
        corpus = corpora[str(case["corpus_ref"])]
        assert sorted(oracle.find_duplicates_py(corpus, case["query"], case["min_match"])) == \
            [tuple(e) for e in case["expected"]]


def test_golden_streaming(golden_dir):
    g = _load(golden_dir, "match_streaming.json")
    for case in g["cases"]:
        corpus = [(v, list(t)) for v, t in _rows(case["corpus"])]
        scene, dup_ids = oracle.streaming_verdict_py(case["stream"], [(v, t) for v, t in corpus],
                                                     case["self_id"], case["min_match"])
        assert scene == case["scene_timestamps"], case["name"]
        assert sorted(dup_ids) == case["dup_ids"], case["name"]
        # batch-equivalent: kth of the deduplicated stream reproduces the early stop
        dedup = []
        for ts in case["stream"]:
            if not dedup or ts != dedup[-1]:
                dedup.append(ts)
        ids, cnt, kth = oracle.match_kth(_rows(case["corpus"]), dedup, case["min_match"])
        kstar, dups = oracle.verdict_from_kth(ids, kth, self_id=case["self_id"])
        assert dups == case["dup_ids"], case["name"]
        if kstar is None:
            assert case["scene_timestamps"] == dedup
        else:
            assert case["scene_timestamps"] == dedup[:kstar + 1]
            # at first detection every reported row has exactly min_match hits (app.py:238)
            assert all(c == case["min_match"] for _, c in case["dups"])


def test_match_kth_sorted_variant_agrees():
    rng = np.random.default_rng(5)
    corpus = [(i, sorted(set(np.round(rng.uniform(0, 50, 30), 1).tolist()))) for i in range(40)]
    q = np.round(rng.uniform(0, 50, 25), 1)
    ids, offs, keys = oracle._csr(corpus)
    for mm in (0, 1, 2, 3):
        a = oracle.match_kth_csr(q, offs, keys, mm, sorted_unique=False)
        b = oracle.match_kth_csr(q, offs, keys, mm, sorted_unique=True)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


# ------------------------------------------------------------ scene score (unpinned)

def test_luma_sad_small_exact():
    rng = np.random.default_rng(1)
    f = rng.integers(0, 256, size=(7, 9, 13), dtype=np.uint8)
    sad = oracle.luma_sad(f)
    ref = np.abs(f[1:].astype(np.int64) - f[:-1].astype(np.int64)).sum(axis=(1, 2))
    assert sad[0] == 0 and (sad[1:] == ref.astype(np.uint64)).all()
    # padded rows / frames give the same answer
    big = np.zeros((7, 11, 32), dtype=np.uint8)
    big[:, :9, :13] = f
    assert (oracle.luma_sad(big[:, :9, :13]) == sad).all()


def test_scene_select_c_matches_numpy_restatement():
    rng = np.random.default_rng(2)
    H, W = 48, 64
    # mafd values straddling the 30-level threshold and the float32 rounding of 0.3
    mafds = [0, 5, 29.999, 30.0, 30.000001, 31, 80, 82, 10, 100, 255, 0.3, 30.0000004, 60, 29]
    sad = np.array([int(round(m * H * W)) for m in mafds], dtype=np.uint64)
    sel, score, mafd, last = oracle.scene_select(sad, H, W, 0.3)
    sel_py, score_py = oracle.scene_select_py(sad, H, W, 0.3)
    assert (sel == sel_py).all() and (score == score_py).all()
    assert sel[0] == 0 and score[0] == 0.0
    assert last == mafd[-1]
    # cut <=> mafd > 30 and |mafd - prev| > 30 (modulo float32 rounding at the boundary)
    for t in range(1, len(mafds)):
        m, p = mafd[t], (mafd[t - 1] if t > 1 else 0.0)
        if min(m, abs(m - p)) > 30.001:
            assert sel[t] == 1
        if min(m, abs(m - p)) < 29.999:
            assert sel[t] == 0


def test_scene_select_float32_clip_boundary():
    # min(mafd,diff)/100 rounds to float32(0.3) = 0.30000001192...; > 0.3 (double) is TRUE
    H, W = 100, 100
    sad = np.array([0, 300000], dtype=np.uint64)  # mafd = 30.0 exactly
    sel, score, _, _ = oracle.scene_select(sad, H, W, 0.3)
    assert score[1] == float(np.float32(0.3)) and sel[1] == 1


def test_scene_select_chunking_equals_whole():
    rng = np.random.default_rng(3)
    H, W = 32, 32
    sad = rng.integers(0, 200 * H * W, size=50).astype(np.uint64)
    sad[0] = 0
    sel, score, mafd, _ = oracle.scene_select(sad, H, W, 0.3)
    s1, sc1, _, last = oracle.scene_select(sad[:20], H, W, 0.3)
    s2, sc2, _, _ = oracle.scene_select(sad[20:], H, W, 0.3, prev_mafd=last, have_prev=True)
    assert (np.concatenate([s1, s2]) == sel).all() and (np.concatenate([sc1, sc2]) == score).all()


def test_pts_time_formats():
    # %.6g (FFmpeg <= 6.x) keeps 6 significant digits; 7.x keeps 6 decimals, zeros trimmed
    assert oracle.fmt_pts_time(370, 1, 30, 0) == "12.3333"
    assert oracle.fmt_pts_time(370, 1, 30, 1) == "12.333333"
    assert oracle.fmt_pts_time(3704, 1, 30, 0) == "123.467"
    assert oracle.fmt_pts_time(37037, 1, 30, 0) == "1234.57"
    assert oracle.fmt_pts_time(30, 1, 30, 0) == "1"
    assert oracle.fmt_pts_time(30, 1, 30, 1) == "1"
    assert oracle.fmt_pts_time(0, 1, 30, 0) == "0"
    assert oracle.fmt_pts_time(0, 1, 30, 1) == "0"
    assert oracle.fmt_pts_time(15, 1, 30, 1) == "0.5"
    assert oracle.fmt_pts_time(1, 1, 1000, 1) == "0.001"
    assert oracle.fmt_pts_time(512 * 7, 1, 15360, 0) == "0.233333"
    assert oracle.pts_time_value(370, 1, 30, 0) == 12.3333


def test_host_formatter_matches_oracle():
    from tvidz_amd import scene
    rng = np.random.default_rng(4)
    for tb in [(1, 30), (1, 25), (1, 24), (1001, 30000), (1, 15360), (1, 90000), (1, 1000)]:
        for pts in [0, 1, 2, 29, 30, 31, 100, 12345] + rng.integers(0, 10**7, 300).tolist():
            for pol_c, pol_py in ((0, scene.PTS_POLICY_G6), (1, scene.PTS_POLICY_F6TRIM)):
                assert oracle.fmt_pts_time(pts, tb[0], tb[1], pol_c) == \
                    scene.format_pts_time(pts, tb, pol_py), (pts, tb, pol_py)
                assert oracle.pts_time_value(pts, tb[0], tb[1], pol_c) == \
                    scene.pts_time_value(pts, tb, pol_py)


def test_parse_showinfo_line_follows_reference_parser():
    from tvidz_amd.scene import parse_showinfo_line
    line = ("[Parsed_showinfo_1 @ 0x55d0c8] n:   3 pts:   1110 pts_time:37      duration:      1 "
            "duration_time:0.0333333 fmt:yuv420p")
    assert parse_showinfo_line(line) == 37.0
    assert parse_showinfo_line("[Parsed_showinfo_1 @ 0x1] n:   0 pts:  370 pts_time:12.3333 pos: 1") == 12.3333
    assert parse_showinfo_line("frame=  100 fps=0.0 q=-0.0 size=N/A time=00:00:03.33") is None
    assert parse_showinfo_line("[Parsed_showinfo_1 @ 0x1] config in time_base: 1/30") is None
