"""CPU tests of the host logic either side of the kernels: the db drop-in's SQL layer (same
schema and upsert rule as inspector/db.py) and the Flask routes (same cases as the reference's
inspector/test_app.py:6-64).  The device corpus is replaced by tests/fakes.OracleCorpus."""
import json
import os
import threading
import time

import numpy as np
import pytest

from tests.fakes import OracleCorpus
from tvidz_amd import db as tdb
from tvidz_amd import feeder, inspector as insp


@pytest.fixture()
def store():
    s = tdb.Store("sqlite://", corpus=OracleCorpus())
    yield s
    s.close()


@pytest.fixture()
def client(store):
    ins = insp.Inspector(store, device="cuda:0", frame_source=lambda *a: (_ for _ in ()).throw(RuntimeError("no source")))
    app = insp.create_app(ins, sse_period=0.01)
    return app.test_client(), ins


def test_schema_matches_reference():
    # inspector/db.py:12-27
    v, t = tdb.Video.__table__, tdb.VideoTimestamps.__table__
    assert v.name == "videos" and t.name == "video_timestamps"
    assert [c.name for c in v.columns] == ["id", "filename", "upload_time", "thumbnail_path", "duplicates"]
    assert [c.name for c in t.columns] == ["id", "video_id", "timestamps"]
    assert not v.c.filename.nullable and not t.c.timestamps.nullable
    assert list(t.c.video_id.foreign_keys)[0].target_fullname == "videos.id"
    from sqlalchemy.dialects import postgresql
    from sqlalchemy.schema import CreateTable
    ddl = str(CreateTable(t).compile(dialect=postgresql.dialect()))
    assert "timestamps FLOAT[] NOT NULL" in ddl
    ddl = str(CreateTable(v).compile(dialect=postgresql.dialect()))
    assert "duplicates INTEGER[]" in ddl and "filename VARCHAR NOT NULL" in ddl


def test_duplicate_detection_like_reference_test(store):
    # inspector/test_app.py:66-83 through the drop-in module functions
    v1 = store.add_video("a.mp4")
    v2 = store.add_video("b.mp4")
    store.add_timestamps(v1.id, [1.0, 2.0, 3.0, 4.0, 5.0])
    store.add_timestamps(v2.id, [10.0, 20.0, 30.0, 40.0, 50.0])
    dups = store.find_duplicates([10.0, 20.0, 30.0, 40.0, 50.0], min_match=5)
    assert (v1.id, 0) not in dups and (v2.id, 5) in dups
    v3 = store.add_video("c.mp4")
    store.add_timestamps(v3.id, [1.0, 2.0, 3.0, 4.0, 5.0])
    dups = store.find_duplicates([1.0, 2.0, 3.0, 4.0, 5.0], min_match=5)
    assert (v1.id, 5) in dups and (v3.id, 5) in dups


def test_add_timestamps_upserts_one_row_per_video(store):
    v = store.add_video("x.mp4")
    for k in range(1, 6):
        store.add_timestamps(v.id, [float(i) for i in range(k)])      # growing prefix, app.py:234
    s = store.SessionLocal()
    try:
        rows = s.query(tdb.VideoTimestamps).filter_by(video_id=v.id).all()
        assert len(rows) == 1 and rows[0].timestamps == [0.0, 1.0, 2.0, 3.0, 4.0]
    finally:
        s.close()
    assert store.corpus.rows == [(v.id, [0.0, 1.0, 2.0, 3.0, 4.0])]
    store.update_duplicates(v.id, [7, 9])
    assert store.get_video_by_id(v.id).duplicates == [7, 9]
    assert store.get_video_by_filename("x.mp4").id == v.id
    assert store.get_video_by_id(12345) is None


def test_reload_corpus_from_sql(store):
    a, b = store.add_video("a"), store.add_video("b")
    store.add_timestamps(a.id, [1.5, 2.5])
    store.add_timestamps(b.id, [3.5])
    store.corpus.clear()
    assert store.find_duplicates([1.5], 1) == []
    assert store.reload_corpus() == 2
    assert store.find_duplicates([1.5, 3.5], 1) == [(a.id, 1), (b.id, 1)]


def test_module_level_api_names(monkeypatch):
    s = tdb.init("sqlite://", corpus=OracleCorpus())
    try:
        from tvidz_amd.db import (add_timestamps, add_video, find_duplicates, get_video_by_filename,
                                  get_video_by_id, update_duplicates)
        v = add_video("m.mp4")
        add_timestamps(v.id, [1.2, 5.7, 12.3, 18.9])
        assert find_duplicates([1.2, 5.7, 12.3, 18.9], min_match=2) == [(v.id, 4)]
        update_duplicates(v.id, [1])
        assert get_video_by_id(v.id).filename == "m.mp4" and get_video_by_filename("m.mp4").id == v.id
        import inspect
        assert inspect.signature(find_duplicates).parameters["min_match"].default == 5   # db.py:76
    finally:
        s.close()
        tdb._default = None


# ---- routes: the reference's own cases (inspector/test_app.py:6-64) ----------------------

def test_status_pending(client):
    c, _ = client
    resp = c.get("/status/nonexistentfile.mp4")
    assert resp.status_code == 200 and resp.get_json()["status"] == "pending"


def test_status_stream_options(client):
    c, _ = client
    resp = c.options("/status/stream/somefile.mp4")
    assert resp.status_code == 200 and resp.headers["Access-Control-Allow-Origin"] == "*"


def test_notify_bad_event(client):
    c, _ = client
    resp = c.post("/notify", json={"foo": "bar"})
    assert resp.status_code == 400 and "error" in resp.get_json()


def test_notify_valid_event(client, monkeypatch):
    c, ins = client
    called = {}
    monkeypatch.setattr(ins, "submit", lambda bucket, key: called.update(bucket=bucket, key=key))
    event = {"Records": [{"s3": {"bucket": {"name": "videos"}, "object": {"key": "test.mp4"}}}]}
    resp = c.post("/notify", data=json.dumps(event), content_type="application/json")
    assert resp.status_code == 200
    data = resp.get_json()
    assert data["status"] == "Analysis started" and data["file"] == "test.mp4"
    assert called == {"bucket": "videos", "key": "test.mp4"}


def test_clear_db_and_build_info(client):
    c, ins = client
    v = ins.store.add_video("z.mp4")
    ins.store.add_timestamps(v.id, [1.0])
    resp = c.post("/admin/clear-db")
    assert resp.status_code == 200 and resp.get_json()["status"] == "cleared"
    assert ins.store.list_videos() == [] and ins.store.corpus.rows == []
    resp = c.get("/build-info")
    assert "build_date" in resp.get_json()["inspector"]


def test_debug_routes(client):
    c, ins = client
    r = c.post("/debug/create-test-video", json={"filename": "t.mp4", "timestamps": [1.2, 5.7]})
    assert r.get_json()["status"] == "created"
    r = c.get("/debug/videos").get_json()
    assert r["count"] == 1 and r["videos"][0]["timestamps"] == [1.2, 5.7]
    r = c.post("/debug/test-duplicate").get_json()
    assert [r["first_video_id"], 4] in r["duplicates_found"]
    assert c.get("/debug/analysis-results").get_json()["count"] == 0


def test_error_state_and_sse_stream(client):
    c, ins = client
    res = ins.analyze_file("videos", "1700000000-clip.mp4")       # frame source raises
    assert res["status"] == "error" and "no source" in res["error"]
    assert res["original_filename"] == "1700000000-clip.mp4" and res["clean_filename"] == "clip.mp4"
    assert c.get("/status/1700000000-clip.mp4").get_json()["status"] == "error"
    body = c.get("/status/stream/1700000000-clip.mp4").get_data(as_text=True)
    events = [json.loads(l[6:]) for l in body.split("\n\n") if l.startswith("data: ")]
    assert events[-1]["status"] == "error" and len(events) == 1
    # a stream on an unknown file first says pending, then follows the record to `done`
    def later():
        time.sleep(0.05)
        ins._set("k", {"status": "analyzing", "scene_cuts": [1.0], "progress": 0.5, "duplicates": [],
                       "original_filename": "new.mp4"})
        time.sleep(0.05)
        ins._set("k", {"status": "done", "scene_cuts": [1.0, 2.0], "progress": 1.0, "duplicates": ["a.mp4"],
                       "original_filename": "new.mp4"})
    threading.Thread(target=later).start()
    body = c.get("/status/stream/new.mp4").get_data(as_text=True)
    events = [json.loads(l[6:]) for l in body.split("\n\n") if l.startswith("data: ")]
    assert [e["status"] for e in events] == ["pending", "analyzing", "done"]


def test_split_filenames():
    assert insp.split_filenames("uploads/1723456789-my-video.mp4") == ("1723456789-my-video.mp4", "my-video.mp4")
    assert insp.split_filenames("plain.mp4") == ("plain.mp4", "plain.mp4")
    assert insp.split_filenames("abc-def.mp4") == ("abc-def.mp4", "abc-def.mp4")
    assert insp.split_filenames("") == ("unknown_file", "unknown_file")


def test_y4m_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    luma = rng.integers(0, 256, size=(7, 18, 22), dtype=np.uint8)
    for chroma in ("mono", "420jpeg", "444"):
        p = str(tmp_path / f"c_{chroma}.y4m")
        feeder.write_y4m(p, luma, fps=(25, 1), chroma=chroma)
        r = feeder.Y4MReader(p)
        assert (r.W, r.H, r.time_base, r.total_frames) == (22, 18, (1, 25), 7)
        got = np.stack(list(r))
        assert (got == luma).all()
        r.close()
        r = feeder.Y4MReader(p)
        buf = np.zeros((5, 18, 22), dtype=np.uint8)
        assert r.read_into(buf) == 5 and (buf == luma[:5]).all()
        assert r.read_into(buf) == 2 and (buf[:2] == luma[5:]).all()
        r.close()


def test_y4m_high_bit_depth_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    luma = rng.integers(0, 1024, size=(5, 10, 14), dtype=np.uint16)
    for chroma in ("mono", "420", "444"):
        p = str(tmp_path / f"hb_{chroma}.y4m")
        feeder.write_y4m(p, luma, fps=(24, 1), chroma=chroma, bitdepth=10)
        r = feeder.Y4MReader(p)
        assert (r.W, r.H, r.bitdepth, r.bps, r.total_frames) == (14, 10, 10, 2, 5)
        assert (np.stack(list(r)) == luma).all()
        r.close()
        r = feeder.Y4MReader(p)
        buf = np.zeros((5, 10, 14), dtype=np.int16)
        assert r.read_into(buf) == 5 and (buf.view(np.uint16) == luma).all()
        r.close()


def test_write_behind_coalesces_and_flushes(store):
    """add_timestamps_async: the device row is upserted at once, the SQL row is written behind by
    one thread that keeps only the latest prefix of a video; flush() makes the table identical to
    what per-cut commits (db.py:58-62) would have left."""
    a, b = store.add_video("a.mp4"), store.add_video("b.mp4")
    for k in range(1, 40):
        store.add_timestamps_async(a.id, [float(i) for i in range(k)])
        assert store.corpus.rows[0] == (a.id, [float(i) for i in range(k)])     # visible to the next match at once
        if k % 3 == 0:
            store.add_timestamps_async(b.id, [100.0 + i for i in range(k)])
    store.flush(a.id)
    store.flush()
    s = store.SessionLocal()
    try:
        rows = {r.video_id: r.timestamps for r in s.query(tdb.VideoTimestamps).all()}
    finally:
        s.close()
    assert rows == {a.id: [float(i) for i in range(39)], b.id: [100.0 + i for i in range(39)]}
    assert store.sync_if_stale() is False            # our own rows are not "someone else's"
    # a row added behind the store's back (another worker / plain SQL) is picked up
    s = store.SessionLocal()
    try:
        v = tdb.Video(filename="ext.mp4")
        s.add(v); s.commit()
        s.add(tdb.VideoTimestamps(video_id=v.id, timestamps=[7.5, 8.5])); s.commit()
        ext_id = v.id
    finally:
        s.close()
    assert store.find_duplicates([7.5, 8.5], 2) == []
    assert store.sync_if_stale() is True
    assert store.find_duplicates([7.5, 8.5], 2) == [(ext_id, 2)]
    store.clear()
    assert store.sync_if_stale() is False and store.find_duplicates([7.5], 1) == []


def _sql_rows(store):
    s = store.SessionLocal()
    try:
        return {r.video_id: r.timestamps for r in s.query(tdb.VideoTimestamps).all()}
    finally:
        s.close()


def test_write_behind_failure_is_retried_then_raised_once_to_its_owner(store):
    """A transient SQL error must cost nothing; a persistent one fails only the upload that owns
    the row (db.py:52-62 commits per call: one failing request, not every later one), exactly once,
    and the mirror is reloaded from SQL afterwards."""
    a, b = store.add_video("a.mp4"), store.add_video("b.mp4")
    real = store._write_timestamps_sql
    fails = {"n": 1}

    def flaky(session, vid, ts):
        if fails["n"] > 0:
            fails["n"] -= 1
            raise RuntimeError("connection reset")
        return real(session, vid, ts)
    store._write_timestamps_sql = flaky
    store.add_timestamps_async(a.id, [1.0, 2.0])
    store.flush(a.id)                                   # retried behind the scenes: no error
    assert _sql_rows(store) == {a.id: [1.0, 2.0]}
    # three failures in a row = the writer gives up on this batch
    fails["n"] = store._wb_retries
    store.add_timestamps_async(b.id, [5.0, 6.0])
    with pytest.raises(RuntimeError, match="connection reset"):
        store.flush(b.id)
    store.flush(b.id)                                   # raised once; not sticky
    store.flush()
    assert store._dirty and b.id not in _sql_rows(store)
    assert (b.id, [5.0, 6.0]) in store.corpus.rows      # the mirror is ahead of SQL ...
    store.add_timestamps_async(a.id, [1.0, 2.0, 3.0])   # ... and other uploads go on unharmed
    store.flush(a.id)
    assert store.sync_if_stale() is True                # ... until the next sync reloads it from SQL
    assert store.corpus.rows == [(a.id, [1.0, 2.0, 3.0])]
    # the owner learns about a failed write at its next call too (before the device row is touched)
    fails["n"] = store._wb_retries
    store.add_timestamps_async(b.id, [5.0])
    deadline = time.time() + 10
    while time.time() < deadline and b.id not in store._wb_errors:
        time.sleep(0.01)
    rows_before = list(store.corpus.rows)
    with pytest.raises(RuntimeError, match="connection reset"):
        store.add_timestamps_async(b.id, [5.0, 6.0])
    assert store.corpus.rows == rows_before
    store.add_timestamps_async(b.id, [5.0, 6.0])
    store.flush(b.id)
    assert _sql_rows(store)[b.id] == [5.0, 6.0]


def test_upload_error_record_when_sync_fails(store):
    """sync_if_stale runs inside analyze_file's try: a failure there is the upload's
    `status: error` record (app.py:303-315), not a worker that died without one."""
    ins = insp.Inspector(store, device="cuda:0", frame_source=lambda *a: (_ for _ in ()).throw(RuntimeError("no source")))

    def boom(min_interval=0.0):
        raise RuntimeError("census failed")
    store.sync_if_stale = boom
    res = ins.analyze_file("videos", "1700000000-x.mp4")
    assert res["status"] == "error" and "census failed" in res["error"]
    assert ins.result_for("x.mp4") is None and any(r["status"] == "error" for r in ins.analysis_results.values())
    ins.close()


def test_sync_replays_rows_of_uploads_that_raced_the_reload(store):
    """An add_timestamps_async between flush() and reload_corpus() has its HBM row upserted and
    its SQL write queued; the reload reads a table that does not hold the row yet.  The queued and
    in-flight rows are replayed into the mirror after the reload."""
    a = store.add_video("a.mp4")
    store.add_timestamps(a.id, [1.0, 2.0])
    b = store.add_video("b.mp4")
    store._wait_write_behind = lambda: None             # the race: the writer has not committed b yet
    with store._wb_cv:
        store._pending[b.id] = [8.0, 9.0]               # queued ...
        store._inflight = {777: [3.0]}                   # ... and one batch being committed
    store._dirty = True
    assert store.sync_if_stale() is True
    assert sorted(store.corpus.rows) == sorted([(a.id, [1.0, 2.0]), (b.id, [8.0, 9.0]), (777, [3.0])])
    with store._wb_cv:
        store._pending.clear()
        store._inflight = {}


def test_audit_finds_rows_updated_in_place_by_another_writer(store):
    """db.py:83 re-reads the table on every call, so the reference sees another worker's UPDATE at
    once.  Here count(*)/max(id) do not move for an in-place UPDATE; the audit pass finds it by
    content - and leaves alone the rows whose newer list is still queued for SQL."""
    a, b, c = store.add_video("a.mp4"), store.add_video("b.mp4"), store.add_video("c.mp4")
    store.add_timestamps(a.id, [1.0, 2.0, 3.0])
    store.add_timestamps(b.id, [10.0, 11.0])
    store.add_timestamps_async(c.id, [20.0, 21.0])
    store.flush()
    assert store.audit() == 0                           # everything SQL holds is what this process wrote
    s = store.SessionLocal()                            # "another worker": plain SQL on the same table
    try:
        row = s.query(tdb.VideoTimestamps).filter_by(video_id=a.id).first()
        row.timestamps = [1.0, 2.0, 4.5, 5.5]
        s.commit()
    finally:
        s.close()
    assert store.sync_if_stale() is False               # the census cannot see it ...
    assert store.find_duplicates([4.5, 5.5], 2) == []
    assert store.audit(chunk_rows=2) == 1               # ... the audit does (in chunks of two rows)
    assert store.find_duplicates([4.5, 5.5], 2) == [(a.id, 2)]
    assert store.find_duplicates([3.0], 1) == []        # the replaced content is gone from the mirror
    assert store.audit() == 0 and store.audit_repairs == 1
    # a row whose newer list is queued for SQL is this process's own: HBM stays ahead of SQL
    store.flush = lambda video_id=None: None
    with store._wb_cv:
        store._pending[b.id] = [10.0, 11.0, 12.0]
    store.corpus.upsert(b.id, [10.0, 11.0, 12.0])
    assert store.audit() == 0 and (b.id, [10.0, 11.0, 12.0]) in store.corpus.rows
    with store._wb_cv:
        store._pending.clear()


def test_audit_and_sync_take_their_locks_in_one_order(store):
    """audit() took write lock -> mirror lock, sync_if_stale mirror lock -> write lock; with
    another writer adding rows (stale census) AND updating rows in place (audit mismatch) the two
    deadlocked and the upload hung inside analyze_file.  Both run here, many times, against a table
    that keeps changing behind the store's back; every thread must come back."""
    vids = [store.add_video(f"{i}.mp4") for i in range(6)]
    for i, v in enumerate(vids):
        store.add_timestamps(v.id, [float(i), float(i) + 0.5])
    stop = threading.Event()
    errors = []

    def other_writer():
        n = 0
        while not stop.is_set():
            s = store.SessionLocal()
            try:
                with store._sql_write:
                    row = s.query(tdb.VideoTimestamps).filter_by(video_id=vids[n % 6].id).first()
                    row.timestamps = [100.0 + n, 200.0 + n]                 # in place: only the audit sees it
                    if n % 3 == 0:
                        v = tdb.Video(filename=f"ext{n}.mp4")
                        s.add(v); s.flush()
                        s.add(tdb.VideoTimestamps(video_id=v.id, timestamps=[float(n)]))   # a new row: stale census
                    s.commit()
            except Exception as e:                      # pragma: no cover - reported below
                errors.append(e)
            finally:
                s.close()
            n += 1

    def loop(fn):
        try:
            for _ in range(60):
                fn()
        except Exception as e:                          # pragma: no cover
            errors.append(e)

    w = threading.Thread(target=other_writer, daemon=True)
    t1 = threading.Thread(target=loop, args=(store.audit,), daemon=True)
    t2 = threading.Thread(target=loop, args=(store.sync_if_stale,), daemon=True)
    t3 = threading.Thread(target=loop, args=(lambda: store.add_timestamps_async(vids[0].id, [1.0, 2.0]),), daemon=True)
    for t in (w, t1, t2, t3):
        t.start()
    for t in (t1, t2, t3):
        t.join(timeout=60)
    stop.set()
    w.join(timeout=10)
    assert not any(t.is_alive() for t in (t1, t2, t3)), "audit() and sync_if_stale() deadlocked"
    assert not errors, errors
    store.flush()


def test_audit_with_several_sql_rows_per_video(store):
    """video_timestamps has no UNIQUE(video_id) (db.py:22-27) and find_duplicates reads every row
    (db.py:83-91).  A digest per VIDEO saw a mismatch on each of a video's rows in turn, on every
    pass, and its upsert replaced only the first device row (mirror B,B for SQL A,B).  Digests are per
    row id; a changed row of a multi-row video marks the mirror dirty and the reload keeps both."""
    a = store.add_video("a.mp4")
    store.add_timestamps(a.id, [1.0, 2.0])
    s = store.SessionLocal()
    try:
        s.add(tdb.VideoTimestamps(video_id=a.id, timestamps=[5.0, 6.0])); s.commit()      # an older writer's second row
    finally:
        s.close()
    assert store.sync_if_stale() is True
    assert sorted(store.corpus.rows) == [(a.id, [1.0, 2.0]), (a.id, [5.0, 6.0])]
    assert store.audit() == 0 and store.audit() == 0                  # two rows of one video: no phantom mismatch
    s = store.SessionLocal()
    try:
        s.query(tdb.VideoTimestamps).filter_by(video_id=a.id).order_by(tdb.VideoTimestamps.id.desc()).first() \
            .timestamps = [5.0, 7.0]
        s.commit()
    finally:
        s.close()
    assert store.audit() == 1 and store._dirty                         # not upserted over the FIRST row
    assert store.audit() == 0 and store.audit_repairs == 1             # ... and not found again
    assert store.sync_if_stale() is True
    assert sorted(store.corpus.rows) == [(a.id, [1.0, 2.0]), (a.id, [5.0, 7.0])]
    assert store.find_duplicates([5.0, 7.0], 2) == [(a.id, 2)]


def test_sync_leaves_a_failed_write_to_the_upload_that_owns_it(store):
    """sync_if_stale() called flush(), which raised the first write-behind error of ANY
    upload and cleared the rest - an unrelated upload went `status: error` and the owner reported
    `done` with no SQL row.  The error now waits for its owner."""
    a, b = store.add_video("a.mp4"), store.add_video("b.mp4")
    real = store._write_timestamps_sql
    fails = {"n": store._wb_retries}

    def flaky(session, vid, ts):
        if fails["n"] > 0:
            fails["n"] -= 1
            raise RuntimeError("connection reset")
        return real(session, vid, ts)
    store._write_timestamps_sql = flaky
    store.add_timestamps_async(b.id, [5.0, 6.0])
    deadline = time.time() + 10
    while time.time() < deadline and b.id not in store._wb_errors:
        time.sleep(0.01)
    assert store._dirty
    assert store.sync_if_stale() is True                # another upload's census: no error raised here
    with pytest.raises(RuntimeError, match="connection reset"):
        store.flush(b.id)                               # the owner gets it, once
    store.flush(b.id)
    store.flush(a.id)


def test_audit_thread_repairs_in_the_background(tmp_path):
    st = tdb.Store(f"sqlite:///{tmp_path}/t.db", corpus=OracleCorpus(), audit_interval=0.05)
    try:
        v = st.add_video("a.mp4")
        st.add_timestamps(v.id, [1.0, 2.0])
        s = st.SessionLocal()
        try:
            s.query(tdb.VideoTimestamps).filter_by(video_id=v.id).first().timestamps = [7.0, 8.0]
            s.commit()
        finally:
            s.close()
        deadline = time.time() + 10
        while time.time() < deadline and st.audit_repairs == 0:
            time.sleep(0.02)
        assert st.audit_repairs == 1 and st.find_duplicates([7.0, 8.0], 2) == [(v.id, 2)]
    finally:
        st.close()


def test_native_span_reads_skip_chroma_and_stop_at_partial_records(tmp_path):
    """tvz_read_records behind Y4MReader.read_into: a micro-batch per call, Y planes only (4:2:0
    chroma skipped by the record stride), the count of WHOLE frames at the end of a truncated file,
    a loud error on a corrupt FRAME header."""
    rng = np.random.default_rng(5)
    luma = rng.integers(0, 256, size=(7, 18, 34), dtype=np.uint8)
    path = str(tmp_path / "c.y4m")
    feeder.write_y4m(path, luma, fps=(25, 1), chroma="420jpeg")
    r = feeder.Y4MReader(path)
    assert r._span_at is not None and r.total_frames == 7 and r.time_base == (1, 25)
    out = np.zeros((4, 18, 34), dtype=np.uint8)
    assert r.read_into(out) == 4 and (out == luma[:4]).all()
    assert r.read_into(out) == 3 and (out[:3] == luma[4:]).all()
    assert r.read_into(out) == 0
    r.close()
    assert [f.tolist() for f in feeder.Y4MReader(path)] == [f.tolist() for f in luma]      # iteration uses it too
    size = os.path.getsize(path)
    with open(path, "r+b") as f:
        f.truncate(size - 10)                             # the last record loses part of its chroma: still 7 Y planes
    with open(path, "r+b") as f:
        f.truncate(size - (2 * 17 * 9 + 5))               # ... and now part of its Y plane: 6 frames
    r = feeder.Y4MReader(path)
    big = np.zeros((16, 18, 34), dtype=np.uint8)
    assert r.read_into(big) == 6 and (big[:6] == luma[:6]).all()
    r.close()
    with open(path, "r+b") as f:
        hdr = f.readline()
        f.seek(len(hdr) + (6 + 18 * 34 + 2 * 17 * 9))     # second record's header
        f.write(b"FRAMX\n")
    r = feeder.Y4MReader(path)
    with pytest.raises(RuntimeError, match="expected header"):
        r.read_into(big)
    r.close()


def test_a_rank_store_never_audits(tmp_path, monkeypatch):
    """with census=False (a rank of the N-rank service) the sibling ranks' rows would all look "updated by
    another writer" to the periodic audit and be upserted into this rank's shard too.  A rank's store ignores
    TVZ_AUDIT_INTERVAL and an explicit audit_interval."""
    from tests.fakes import OracleCorpus
    from tvidz_amd import db
    monkeypatch.setenv("TVZ_AUDIT_INTERVAL", "0.05")
    s = db.Store(f"sqlite:///{tmp_path}/r.db", corpus=OracleCorpus(), census=False, audit_interval=0.05)
    try:
        assert s.audit_interval == 0.0 and getattr(s, "_audit_thread", None) is None
    finally:
        s.close()
