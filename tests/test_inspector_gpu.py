"""GPU end-to-end: Y4M clips -> feeder -> HIP scene kernels -> HIP corpus match -> verdicts,
checked against the oracle's replay of the reference loop (inspector/app.py:228-255).
BASELINE.json configs[0] shape: 10 s 480p30 clips, one a cut-shifted copy of the other."""
import json

import numpy as np
import pytest

from oracle import oracle
from tvidz_amd import db as tdb
from tvidz_amd import feeder, inspector as insp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, W, T = 480, 854, 300


def _clip(seed, cut_frames):
    """300 frames with cuts exactly at cut_frames (flat scenes + noise)."""
    rng = np.random.default_rng(seed)
    levels = [40, 120, 200, 75, 160, 230, 30, 110]
    out = np.empty((T, H, W), dtype=np.uint8)
    bounds = [0] + list(cut_frames) + [T]
    base_rng = np.random.default_rng(99)
    texture = base_rng.integers(-6, 7, size=(H, W))
    for i in range(len(bounds) - 1):
        for t in range(bounds[i], bounds[i + 1]):
            out[t] = np.clip(levels[i] + texture + rng.integers(-2, 3, size=(H, W)), 0, 255)
    return out


def _oracle_cuts(luma, tb=(1, 30)):
    sad = oracle.luma_sad(luma)
    sel, _, _, _ = oracle.scene_select(sad, luma.shape[1], luma.shape[2], 0.3)
    return [oracle.pts_time_value(int(i), tb[0], tb[1], 0) for i in np.flatnonzero(sel)]


@pytest.fixture()
def rig(tmp_path):
    store = tdb.Store(f"sqlite:///{tmp_path}/tvidz.db", device=0)   # file DB: one connection per thread
    files = {}

    def source(bucket, key, filename, unique_id):
        return feeder.Y4MReader(files[key]), None

    ins = insp.Inspector(store, device=DEV, frame_source=source, batch=64)
    yield ins, store, files, tmp_path
    store.close()


def test_feeder_delivers_every_frame(tmp_path):
    luma = _clip(1, [50, 120])
    p = str(tmp_path / "a.y4m")
    feeder.write_y4m(p, luma, chroma="420jpeg")
    got = []
    for base, d in feeder.FrameFeeder(feeder.Y4MReader(p), batch=64, device=DEV):
        assert base == sum(x.shape[0] for x in got)
        got.append(d.cpu().numpy().copy())
    assert (np.concatenate(got) == luma).all()


def test_config0_two_clips_and_a_copy(rig):
    ins, store, files, tmp = rig
    cuts_a = [45, 100, 150, 210, 260]
    a = _clip(1, cuts_a)
    b = _clip(2, [c + 7 for c in cuts_a])          # cut-shifted copy
    for name, luma in (("1700000001-a.y4m", a), ("1700000002-b.y4m", b), ("1700000003-a_copy.y4m", a)):
        files[name] = str(tmp / name)
        feeder.write_y4m(files[name], luma)
    exp_a, exp_b = _oracle_cuts(a), _oracle_cuts(b)
    assert len(exp_a) == 5 and len(exp_b) == 5
    assert [round(x * 30) for x in exp_a] == cuts_a

    ra = ins.analyze_file("videos", "1700000001-a.y4m")
    assert ra["status"] == "done" and ra["scene_cuts"] == exp_a and ra["duplicates"] == []
    assert ra["total_cuts"] == 5 and ra["progress"] == 1.0
    assert ra["original_filename"] == "1700000001-a.y4m" and ra["clean_filename"] == "a.y4m"

    # the cut-shifted copy has NO exact timestamp in common: not a duplicate for the reference
    rb = ins.analyze_file("videos", "1700000002-b.y4m")
    assert rb["status"] == "done" and rb["scene_cuts"] == exp_b and rb["duplicates"] == []

    # the exact copy is detected at its 2nd cut and analysis stops there (app.py:238-255)
    rc = ins.analyze_file("videos", "1700000003-a_copy.y4m")
    assert rc["status"] == "done" and rc["duplicates"] == ["a.y4m"]
    assert rc["scene_cuts"] == exp_a[:2] and rc["total_cuts"] == 2

    # same verdicts from the oracle's replay of the reference loop
    corpus = []
    for vid, cuts in ((1, exp_a), (2, exp_b), (3, exp_a)):
        scene_ts, dup_ids = oracle.streaming_verdict_py(cuts, corpus, vid, 2)
        assert {1: (exp_a, []), 2: (exp_b, []), 3: (exp_a[:2], [1])}[vid] == (scene_ts, dup_ids)
    vids = store.list_videos()
    assert [v["filename"] for v in vids] == ["a.y4m", "b.y4m", "a_copy.y4m"]
    assert vids[2]["duplicates"] == [vids[0]["id"]] and vids[2]["timestamps"] == exp_a[:2]
    assert ins.result_for("1700000003-a_copy.y4m")["duplicates"] == ["a.y4m"]


def test_sse_reports_done_record(rig):
    ins, store, files, tmp = rig
    files["c.y4m"] = str(tmp / "c.y4m")
    feeder.write_y4m(files["c.y4m"], _clip(3, [30, 90]))
    app = insp.create_app(ins, sse_period=0.01)
    c = app.test_client()
    ev = {"Records": [{"s3": {"bucket": {"name": "videos"}, "object": {"key": "c.y4m"}}}]}
    assert c.post("/notify", json=ev).get_json() == {"status": "Analysis started", "file": "c.y4m"}
    body = c.get("/status/stream/c.y4m").get_data(as_text=True)
    events = [json.loads(l[6:]) for l in body.split("\n\n") if l.startswith("data: ")]
    assert events[-1]["status"] == "done" and events[-1]["scene_cuts"] == [1.0, 3.0]
    assert events[-1]["duplicates"] == []


def test_concurrent_uploads(rig):
    ins, store, files, tmp = rig
    names = []
    for i in range(6):
        n = f"17000000{i:02d}-v{i % 3}.y4m"       # three distinct contents, each uploaded twice
        files[n] = str(tmp / n)
        feeder.write_y4m(files[n], _clip(10 + i % 3, [40 + 11 * (i % 3), 140, 222 + (i % 3)]))
        names.append(n)
    futs = [ins.submit("videos", n) for n in names[:3]]
    [f.result(timeout=120) for f in futs]
    futs = [ins.submit("videos", n) for n in names[3:]]
    res = [f.result(timeout=120) for f in futs]
    assert all(r["status"] == "done" for r in res)
    assert [r["duplicates"] for r in res] == [["v0.y4m"], ["v1.y4m"], ["v2.y4m"]]


def test_ten_bit_upload_end_to_end(rig):
    """A yuv420p10-style clip through reader -> feeder -> 16-bit scene kernels -> verdict."""
    ins, store, files, tmp = rig
    a8 = _clip(5, [40, 111, 190])
    a10 = (a8.astype(np.uint16) << 2) | 1                      # 10-bit samples
    for name in ("1700000050-hdr.y4m", "1700000051-hdr_copy.y4m"):
        files[name] = str(tmp / name)
        feeder.write_y4m(files[name], a10, chroma="420", bitdepth=10)
    sad = oracle.luma_sad(a10)
    sel, _, _, _ = oracle.scene_select(sad, H, W, 0.3, bitdepth=10)
    exp = [oracle.pts_time_value(int(i), 1, 30, 0) for i in np.flatnonzero(sel)]
    assert [round(x * 30) for x in exp] == [40, 111, 190]
    r1 = ins.analyze_file("videos", "1700000050-hdr.y4m")
    assert r1["status"] == "done" and r1["scene_cuts"] == exp and r1["duplicates"] == []
    r2 = ins.analyze_file("videos", "1700000051-hdr_copy.y4m")
    assert r2["status"] == "done" and r2["duplicates"] == ["hdr.y4m"] and r2["scene_cuts"] == exp[:2]


def test_opt_in_near_duplicates_flags_the_cut_shifted_copy(tmp_path):
    """configs[0]: the cut-shifted copy is NOT a duplicate for the reference (exact match) - the
    verdict stays empty - but the opt-in alignment field reports it with its shift."""
    store = tdb.Store(f"sqlite:///{tmp_path}/t.db", device=0)
    files = {}
    ins = insp.Inspector(store, device=DEV, frame_source=lambda b, k, f, u: (feeder.Y4MReader(files[k]), None),
                         batch=64, near_duplicates=True)
    try:
        cuts = [45, 100, 150, 210, 260]
        for name, luma in (("a.y4m", _clip(1, cuts)), ("b.y4m", _clip(2, [c + 7 for c in cuts]))):
            files[name] = str(tmp_path / name)
            feeder.write_y4m(files[name], luma)
        ra = ins.analyze_file("videos", "a.y4m")
        assert ra["duplicates"] == [] and ra["near_duplicates"] == []
        rb = ins.analyze_file("videos", "b.y4m")
        assert rb["status"] == "done" and rb["duplicates"] == []            # reference verdict
        assert len(rb["near_duplicates"]) == 1
        nd = rb["near_duplicates"][0]
        assert nd["filename"] == "a.y4m" and nd["jaccard"] == 1.0 and abs(nd["shift_seconds"] + 7 / 30) < 1e-9
    finally:
        store.close()


def test_container_time_base_and_real_pts_reach_the_fingerprint(tmp_path):
    """an mp4's stream time base is 1/15360 (or 1/90000) and its pts are not the
    frame index; showinfo prints pts x time_base (app.py:230).  A reader that reports such a time
    base and per-frame pts must give the SAME scene_cuts as the same frames in a Y4M at 30 fps -
    and therefore be flagged as its duplicate."""
    store = tdb.Store(f"sqlite:///{tmp_path}/t.db", device=0)
    luma = _clip(7, [33, 90, 171, 240])
    files = {"1700000060-orig.y4m": str(tmp_path / "orig.y4m")}
    feeder.write_y4m(files["1700000060-orig.y4m"], luma)

    class ContainerReader:                      # what FFmpegReader presents for a 30 fps mp4
        H, W, bitdepth, total_frames = H, W, 8, T
        time_base = (1, 15360)

        def __init__(self):
            self.t = 0

        def read_into(self, out):
            n = min(out.shape[0], T - self.t)
            out[:n] = luma[self.t:self.t + n]
            self.t += n
            return n

        def pts_of(self, n):
            return 512 * n                     # 15360 / 30

        def close(self):
            pass

    def source(bucket, key, filename, uid):
        return (ContainerReader(), None) if key.endswith(".mp4") else (feeder.Y4MReader(files[key]), None)
    ins = insp.Inspector(store, device=DEV, frame_source=source, batch=64)
    try:
        r1 = ins.analyze_file("videos", "1700000060-orig.y4m")
        exp = _oracle_cuts(luma)
        assert r1["status"] == "done" and r1["scene_cuts"] == exp and [round(x * 30) for x in exp] == [33, 90, 171, 240]
        r2 = ins.analyze_file("videos", "1700000061-remux.mp4")
        assert r2["status"] == "done", r2
        assert r2["scene_cuts"] == exp[:2] and r2["duplicates"] == ["orig.y4m"]   # same fingerprint: a duplicate
    finally:
        ins.close()
        store.close()


def test_store_bulk_load_and_audit_against_the_device_corpus(tmp_path):
    """f-2 on the real DeviceCorpus: rows written by plain SQL (another worker) are bulk-loaded into
    HBM by reload_corpus (columns -> numpy -> tvz_corpus_upload), an in-place UPDATE is found by the
    audit, and find_duplicates answers like the oracle's db.py:85-91 loop at every step."""
    store = tdb.Store(f"sqlite:///{tmp_path}/t.db", device=0)
    try:
        rng = np.random.default_rng(4)
        grid = np.arange(1, 4001) / 8.0
        rows = [(v, sorted(rng.choice(grid, size=int(rng.integers(5, 60)), replace=False).tolist())) for v in range(1, 301)]
        rows[41] = (42, [])                                        # an empty fingerprint
        s = store.SessionLocal()
        try:
            for v, ts in rows:
                s.add(tdb.VideoTimestamps(video_id=v, timestamps=ts))
            s.commit()
        finally:
            s.close()
        assert store.find_duplicates(rows[7][1], 2) == []            # the mirror has not seen them yet
        assert store.sync_if_stale() is True                       # census: rows appeared -> bulk load
        for q, mm in ((rows[7][1], 2), (rows[100][1][:9], 3), (grid[:200].tolist(), 2)):
            assert store.find_duplicates(q, mm) == sorted(oracle.find_duplicates_py(rows, q, mm))
        s = store.SessionLocal()
        try:
            s.query(tdb.VideoTimestamps).filter_by(video_id=8).first().timestamps = [0.125, 0.25, 0.375]
            s.commit()
        finally:
            s.close()
        rows[7] = (8, [0.125, 0.25, 0.375])
        assert store.sync_if_stale() is False and store.audit() == 1
        for q, mm in (([0.125, 0.25, 0.375], 2), (grid[:200].tolist(), 2)):
            assert store.find_duplicates(q, mm) == sorted(oracle.find_duplicates_py(rows, q, mm))
    finally:
        store.close()
