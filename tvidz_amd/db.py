"""Drop-in for the reference's `db` module (inspector/db.py), same names and call shapes:

    from tvidz_amd.db import (add_video, add_timestamps, update_duplicates, find_duplicates,
                              get_video_by_filename, get_video_by_id)          # app.py:10

What stays the same: the `videos` / `video_timestamps` schema (db.py:12-27, byte-compatible
column names and types on Postgres), the upsert rule of add_timestamps (db.py:52-62), the
signature and result shape of find_duplicates (db.py:76: list of (video_id, match_count)).

What changes: find_duplicates no longer fetches the whole table and loops in Python
(db.py:83-91).  The table is mirrored once into HBM (tvz_corpus), add_timestamps also upserts the
device row, and the match runs in the HIP kernel of csrc/tvz_match.hip.  Results are sorted by
video_id (the reference's order is unspecified: no ORDER BY at db.py:83).

Unlike the reference there is no import-time connection/DDL (db.py:8,30): call init() — or let
the first use do it from POSTGRES_URL (db.py:7), default unchanged.

Ownership of the table: the HBM mirror follows THIS process's add_timestamps / clear calls.  The
reference re-reads `video_timestamps` on every call (db.py:83) and so also sees rows written by
other workers or by plain SQL; here such writers are detected by a cheap census
(`SELECT count(*), max(id)`, Store.sync_if_stale - the driver runs it once per upload) that reloads
the mirror when rows were added or removed behind its back.  An in-place UPDATE of an existing row
by another writer leaves count and max(id) alone; it is found by the AUDIT (Store.audit, run every
`audit_interval` seconds by a background thread when that is > 0 - TVZ_AUDIT_INTERVAL, default
off): one pass over the table in id order, chunk by chunk, comparing a digest of every row with the
digest of what this process last read or wrote there, and upserting the rows that differ.  One audit
reads what the reference reads on EVERY find_duplicates call (db.py:83), so a table shared by
several writers converges within `audit_interval` at a fraction of the reference's SQL traffic;
with one inspector process per table leave it off.
If the device upsert fails after the SQL commit the mirror is marked dirty and reloaded on the
next use.
"""
from __future__ import annotations

import os
import threading
from datetime import datetime
from typing import List, Optional, Tuple

from sqlalchemy import JSON, Column, DateTime, Float, ForeignKey, Integer, String, create_engine
from sqlalchemy.dialects.postgresql import ARRAY as PG_ARRAY
from sqlalchemy.orm import declarative_base, relationship, sessionmaker
from sqlalchemy.pool import StaticPool

DEFAULT_URL = "postgresql://tvidz:tvidz@postgres:5432/tvidz"   # db.py:7

Base = declarative_base()


class Video(Base):                                              # db.py:12-20
    __tablename__ = "videos"
    id = Column(Integer, primary_key=True)
    filename = Column(String, nullable=False)
    upload_time = Column(DateTime, default=datetime.utcnow)
    thumbnail_path = Column(String)
    duplicates = Column(PG_ARRAY(Integer).with_variant(JSON, "sqlite"), default=list)
    timestamps = relationship("VideoTimestamps", back_populates="video", uselist=False)


class VideoTimestamps(Base):                                    # db.py:22-27
    __tablename__ = "video_timestamps"
    id = Column(Integer, primary_key=True)
    video_id = Column(Integer, ForeignKey("videos.id"))
    timestamps = Column(PG_ARRAY(Float).with_variant(JSON, "sqlite"), nullable=False)
    video = relationship("Video", back_populates="timestamps")


def _digest(ts) -> int:
    """Digest of one row's timestamp list (audit): equal lists give equal digests."""
    from array import array
    return hash(array("d", ts).tobytes())       # by content: NaN and -0.0 are their bit patterns


class Store:
    """SQL persistence (same schema as the reference) + the device corpus that mirrors
    `video_timestamps`.  Thread-safe: one short session per call, like the reference."""

    def __init__(self, url: Optional[str] = None, device: int = 0, corpus=None,
                 audit_interval: Optional[float] = None, census: bool = True):
        """`census=False`: the other writers of the table are this service's own sibling ranks (one
        process per GPU, service.py): their rows live in THEIR shards, so a changed row count is not
        a reason to reload this rank's mirror; sync_if_stale then reloads only after a failed write."""
        self.url = url or os.environ.get("POSTGRES_URL", DEFAULT_URL)
        self.census = bool(census)
        kw = {}
        if self.url.startswith("sqlite") and (":memory:" in self.url or self.url in ("sqlite://", "sqlite:///")):
            kw = dict(connect_args={"check_same_thread": False}, poolclass=StaticPool)
        elif self.url.startswith("sqlite"):
            kw = dict(connect_args={"timeout": 30})
        self.engine = create_engine(self.url, **kw)
        if self.url.startswith("sqlite") and not kw.get("poolclass"):
            # the SQLite file backend (tests, probes; production is Postgres as in the reference):
            # WAL keeps readers from blocking the writer, and every write below goes through
            # _write_lock, so 64 upload threads never spin in SQLite's busy handler (its back-off
            # sleeps of up to 100 ms showed up as a bimodal +1 s in the concurrent-upload probe)
            from sqlalchemy import event

            @event.listens_for(self.engine, "connect")
            def _sqlite_pragmas(dbapi_conn, _rec):
                cur = dbapi_conn.cursor()
                cur.execute("PRAGMA journal_mode=WAL")
                cur.execute("PRAGMA synchronous=NORMAL")
                cur.close()
        self.SessionLocal = sessionmaker(bind=self.engine)
        Base.metadata.create_all(self.engine)                   # db.py:30
        if corpus is None:
            from .corpus import DeviceCorpus                    # HIP-only: raises without libtvz.so
            corpus = DeviceCorpus(device)
        self.corpus = corpus
        self._write_lock = threading.Lock()
        # SQLite allows one writer: serialise writes in-process instead of in its busy handler;
        # Postgres (MVCC) needs no such lock
        import contextlib
        self._sql_write = self._write_lock if self.url.startswith("sqlite") else contextlib.nullcontext()
        self._census = (0, 0)            # (row count, max id) of video_timestamps as this process knows it
        self._dirty = False
        # write-behind of the growing cut prefixes (add_timestamps_async): latest list per video
        self._pending = {}
        self._inflight = {}              # the batch the writer thread is committing: video_id -> list
        self._wb_cv = threading.Condition()
        self._wb_errors = {}             # video_id -> exception of its failed write, raised ONCE to its owner
        self._wb_retries = 3
        self._wb_stop = False
        self._wb_thread = None
        # HBM row + queued SQL write of add_timestamps_async happen as one step with respect to a
        # reload of the mirror (sync_if_stale): a reload in between would drop the HBM row for good
        self._mirror_lock = threading.RLock()
        # video_timestamps.id -> digest of the row SQL holds as far as this process knows (audit());
        # by PRIMARY KEY: the table has no UNIQUE(video_id), and the reference's older insert-per-cut
        # code left several rows per video, every one of which find_duplicates reads (db.py:83-91)
        self._sql_digest = {}
        self.audit_interval = float(os.environ.get("TVZ_AUDIT_INTERVAL", "0") if audit_interval is None
                                    else audit_interval)
        if not self.census:
            # A rank of the N-rank service: every row a SIBLING rank writes would look like "updated in place by
            # another writer" to the periodic audit, which would upsert it into THIS rank's shard as well - the video
            # then lives in two shards (duplicate ids in the merged top-k, totals counted twice).  The ranks are each
            # other's only writers by construction, so a rank never audits.
            self.audit_interval = 0.0
        self.audit_repairs = 0           # rows the audits found changed behind this process's back
        self._audit_stop = threading.Event()
        self._audit_thread = None
        self.reload_corpus()
        if self.audit_interval > 0:
            self._audit_thread = threading.Thread(target=self._audit_loop, name="tvz-sql-audit", daemon=True)
            self._audit_thread.start()

    # -- device mirror -------------------------------------------------------
    def reload_corpus(self) -> int:
        """Bulk load: the one `SELECT * FROM video_timestamps` the reference runs per call.
        Columns, not ORM entities, and numpy arrays instead of per-element Python floats: at 100k
        rows the load is dominated by the driver's array decoding, not by this code."""
        import numpy as np
        from sqlalchemy import select
        session = self.SessionLocal()
        try:
            fetched = session.execute(select(VideoTimestamps.id, VideoTimestamps.video_id, VideoTimestamps.timestamps)
                                      .order_by(VideoTimestamps.id)).all()
        finally:
            session.close()
        census = (len(fetched), max((int(r[0]) for r in fetched), default=0))
        live = [(int(r[1]), np.asarray(r[2] or (), dtype=np.float64)) for r in fetched if r[1] is not None]
        if hasattr(self.corpus, "upload_csr"):
            ids = np.fromiter((v for v, _ in live), dtype=np.int32, count=len(live))
            offs = np.zeros(len(live) + 1, dtype=np.int64)
            np.cumsum(np.fromiter((a.size for _, a in live), dtype=np.int64, count=len(live)), out=offs[1:])
            keys = np.concatenate([a for _, a in live]) if live else np.empty(0, dtype=np.float64)
            self.corpus.upload_csr(ids, offs, keys)
        else:                                       # (test doubles and the sharded front end take row lists)
            self.corpus.upload([(v, a.tolist()) for v, a in live])
        self._sql_digest = {int(r[0]): _digest(r[2] or ()) for r in fetched if r[1] is not None}
        self._census = census
        self._dirty = False
        return len(live)

    def sync_if_stale(self, min_interval: float = 0.0) -> bool:
        """Reload the mirror if `video_timestamps` gained or lost rows that this process did not
        write (another worker, plain SQL), or if a device upsert or a write-behind commit failed.
        `min_interval` > 0 skips the census when one ran less than that many seconds ago (the
        driver asks once per upload; a burst of uploads shares one census).  Failed write-behinds of
        OTHER uploads are not raised here: they stay with their owners (flush(video_id))."""
        from sqlalchemy import func
        import time as _time
        now = _time.monotonic()
        if not self.census and not self._dirty:
            return False
        if min_interval > 0 and not self._dirty and now - getattr(self, "_census_at", -1e9) < min_interval:
            return False
        self._census_at = now
        with self._write_lock:
            session = self.SessionLocal()
            try:
                cnt, mx = session.query(func.count(VideoTimestamps.id), func.max(VideoTimestamps.id)).one()
            finally:
                session.close()
            stale = self._dirty or (int(cnt or 0), int(mx or 0)) != self._census
        if stale:
            # our own write-behind first: its rows must be in SQL.  Only WAIT here - a failed write
            # belongs to the upload that owns the row (its next add_timestamps_async / flush(video_id)
            # raises it); collecting it here would fail whichever upload happened to find the census
            # stale and tell the owner nothing.
            self._wait_write_behind()
            with self._mirror_lock:         # no add_timestamps_async between the reload and the replay
                with self._write_lock:
                    self.reload_corpus()
                # uploads that raced the flush: their rows are queued for SQL (or being committed) but
                # were not in the table the reload read - put them back into the mirror
                with self._wb_cv:
                    replay = dict(self._inflight)
                    replay.update(self._pending)
                for vid, ts in replay.items():
                    self.corpus.upsert(int(vid), ts)
        return stale

    def audit(self, chunk_rows: int = 4096) -> int:
        """One pass over `video_timestamps` in id order: every row whose content is not what this
        process last read or wrote for that ROW (by primary key) was UPDATEd in place by another
        writer (the census of sync_if_stale cannot see that) - its HBM row is replaced by the SQL
        content.  Rows of uploads with a queued or in-flight write-behind are this process's own and
        are skipped (HBM is ahead of SQL there by design).  Returns the number of rows repaired.
        Locks: a chunk is READ under the write lock alone (uploads interleave with a long audit);
        a differing row is repaired under mirror lock -> write lock, the order sync_if_stale takes
        them in (the other order deadlocked an upload's sync against the audit thread), after
        re-reading the row: an add_timestamps in between has made it ours again."""
        repaired, last_id = 0, 0
        while True:
            suspects = []
            with self._write_lock:
                session = self.SessionLocal()
                try:
                    rows = (session.query(VideoTimestamps.id, VideoTimestamps.video_id, VideoTimestamps.timestamps)
                            .filter(VideoTimestamps.id > last_id).order_by(VideoTimestamps.id)
                            .limit(int(chunk_rows)).all())
                finally:
                    session.close()
                if not rows:
                    break
                last_id = int(rows[-1][0])
                with self._wb_cv:
                    own = set(self._pending) | set(self._inflight)
                for rid, vid, ts in rows:
                    if vid is None or int(vid) in own:
                        continue
                    if self._sql_digest.get(int(rid)) != _digest([float(x) for x in (ts or [])]):
                        suspects.append((int(rid), int(vid)))
            for rid, vid in suspects:
                repaired += self._repair_row(rid, vid)
            if len(rows) < int(chunk_rows):
                break
        self.audit_repairs += repaired
        return repaired

    def _repair_row(self, rid: int, vid: int) -> int:
        """audit(): SQL row `rid` of video `vid` differed from what this process knows.  Re-read it
        under mirror lock -> write lock and bring the mirror in line: one SQL row for the video -> its
        HBM row is upserted; several (no UNIQUE(video_id)) -> an upsert would replace only the FIRST
        device row, so the mirror is marked dirty and reloaded whole by the next sync_if_stale."""
        with self._mirror_lock:
            with self._write_lock:
                with self._wb_cv:
                    if vid in self._pending or vid in self._inflight:
                        return 0                         # an upload took the video over: HBM is ahead by design
                session = self.SessionLocal()
                try:
                    sib = (session.query(VideoTimestamps.id, VideoTimestamps.timestamps)
                           .filter_by(video_id=vid).order_by(VideoTimestamps.id).all())
                finally:
                    session.close()
                cur = {int(i): [float(x) for x in (t or [])] for i, t in sib}
                if rid not in cur or self._sql_digest.get(rid) == _digest(cur[rid]):
                    return 0                             # deleted (the census sees that) or ours again
                for i, t in cur.items():
                    self._sql_digest[i] = _digest(t)
                if len(cur) > 1:
                    self._dirty = True
                    return 1
                self.corpus.upsert(vid, cur[rid])
                return 1

    def _audit_loop(self) -> None:
        while not self._audit_stop.wait(self.audit_interval):
            try:
                self.audit()
            except Exception:               # a failed audit (connection lost ...) is retried next period
                pass

    # -- reference API -------------------------------------------------------
    def add_video(self, filename, thumbnail_path=None):         # db.py:32-41
        with self._sql_write:
            session = self.SessionLocal()
            try:
                video = Video(filename=filename, thumbnail_path=thumbnail_path)
                session.add(video)
                session.commit()
                session.refresh(video)
                session.expunge(video)
                return video
            finally:
                session.close()

    def add_timestamps(self, video_id, timestamps):             # db.py:43-64
        ts = [float(x) for x in timestamps]
        with self._write_lock:                                  # keep SQL row and HBM row in step
            session = self.SessionLocal()
            try:
                ts_row = (session.query(VideoTimestamps).filter_by(video_id=video_id)
                          .order_by(VideoTimestamps.id).first())
                if ts_row:
                    ts_row.timestamps = ts
                    session.commit()
                else:
                    ts_row = VideoTimestamps(video_id=video_id, timestamps=ts)
                    session.add(ts_row)
                    session.commit()
                    self._census = (self._census[0] + 1, max(self._census[1], int(ts_row.id)))
                self._sql_digest[int(ts_row.id)] = _digest(ts)
            finally:
                session.close()
            try:
                self.corpus.upsert(int(video_id), ts)
            except Exception:
                self._dirty = True          # SQL has the row, the mirror may not: reload on next use
                raise

    def _write_timestamps_sql(self, session, video_id, ts) -> None:
        """The SQL half of add_timestamps (db.py:52-62) inside the caller's session."""
        ts_row = (session.query(VideoTimestamps).filter_by(video_id=video_id)
                  .order_by(VideoTimestamps.id).first())
        if ts_row:
            ts_row.timestamps = ts
            session.flush()
        else:
            ts_row = VideoTimestamps(video_id=video_id, timestamps=ts)
            session.add(ts_row)
            session.flush()
            self._census = (self._census[0] + 1, max(self._census[1], int(ts_row.id)))
        self._sql_digest[int(ts_row.id)] = _digest(ts)  # (a failed commit marks the mirror dirty: reloaded)

    def add_timestamps_async(self, video_id, timestamps) -> None:
        """add_timestamps for the streaming driver: the HBM row - what the NEXT find_duplicates of
        any upload matches against - is upserted before this returns (stream-ordered, no wait);
        the SQL row is written behind by one writer thread that coalesces the growing prefixes of
        a video (only the latest list matters: db.py:58 overwrites the row) and commits many
        videos per transaction.  flush(video_id) before reporting the upload `done` makes the
        final table state identical to the reference's per-cut commits.
        A failed commit is retried with back-off; if it keeps failing the error is raised ONCE, to
        the upload that owns the row (here or in its flush) - as the reference's per-call commit
        (db.py:52-62) fails only the request it belongs to - and the mirror is reloaded from SQL
        on the next sync_if_stale."""
        ts = [float(x) for x in timestamps]
        vid = int(video_id)
        with self._wb_cv:
            err = self._wb_errors.pop(vid, None)
        if err is not None:                 # before the device row: the mirror gets nothing SQL will not hold
            raise err
        with self._mirror_lock:
            self.corpus.upsert(vid, ts)
            with self._wb_cv:
                self._pending[vid] = ts
                if self._wb_thread is None:
                    self._wb_thread = threading.Thread(target=self._write_behind, name="tvz-sql-writer", daemon=True)
                    self._wb_thread.start()
                self._wb_cv.notify_all()

    def flush(self, video_id=None) -> None:
        """Block until the write-behind has committed `video_id` (or everything).  Raises the
        error of a write that could not be committed - once: flush(video_id) the error of that
        video, flush() the first error there is (and forgets the others: their uploads are gone)."""
        with self._wb_cv:
            while True:
                if video_id is None:
                    busy = bool(self._pending or self._inflight)
                else:
                    busy = int(video_id) in self._pending or int(video_id) in self._inflight
                if not busy:
                    if video_id is not None:
                        err = self._wb_errors.pop(int(video_id), None)
                    else:
                        err = next(iter(self._wb_errors.values()), None)
                        self._wb_errors.clear()
                    if err is not None:
                        raise err
                    return
                self._wb_cv.wait(timeout=0.5)

    def _wait_write_behind(self) -> None:
        """Block until the write-behind queue is empty WITHOUT taking anybody's error (sync_if_stale)."""
        with self._wb_cv:
            while self._pending or self._inflight:
                self._wb_cv.wait(timeout=0.5)

    def _write_behind(self) -> None:
        import time as _time
        while True:
            with self._wb_cv:
                while not self._pending and not self._wb_stop:
                    self._wb_cv.wait(timeout=0.5)
                if self._wb_stop and not self._pending:
                    return
                batch, self._pending = self._pending, {}
                self._inflight = batch
            error = None
            for attempt in range(self._wb_retries):
                try:
                    with self._write_lock:
                        session = self.SessionLocal()
                        try:
                            for vid, ts in batch.items():
                                self._write_timestamps_sql(session, vid, ts)
                            session.commit()
                        finally:
                            session.close()
                    error = None
                    break
                except Exception as e:      # a transient SQL / connection error: back off and retry
                    error = e
                    _time.sleep(0.05 * (attempt + 1))
            with self._wb_cv:
                if error is not None:
                    # given up: every video of the batch that has no newer list queued gets the error
                    # (raised once, to its owner); the mirror may hold rows SQL does not - reload
                    self._dirty = True
                    for vid in batch:
                        if vid not in self._pending:
                            self._wb_errors[vid] = error
                self._inflight = {}
                self._wb_cv.notify_all()

    def update_duplicates(self, video_id, duplicate_ids):       # db.py:66-74
        with self._sql_write:
            session = self.SessionLocal()
            try:
                video = session.query(Video).filter_by(id=video_id).first()
                if video:
                    video.duplicates = [int(d) for d in duplicate_ids]
                    session.commit()
            finally:
                session.close()

    def find_duplicates(self, new_timestamps, min_match=5) -> List[Tuple[int, int]]:   # db.py:76-94
        return self.corpus.find_duplicates(new_timestamps, min_match)

    def find_duplicates_kth(self, new_timestamps, min_match=5, exclude_id=-1):
        """(video_id, match_count, kth): kth = index of the min_match-th matching query element;
        one call replaces the per-prefix loop of app.py:231-255 (see include/tvz.h)."""
        return self.corpus.find_duplicates(new_timestamps, min_match, exclude_id=exclude_id,
                                           with_kth=True)

    def get_video_by_id(self, video_id):                        # db.py:96-102
        session = self.SessionLocal()
        try:
            v = session.query(Video).filter_by(id=video_id).first()
            if v is not None:
                session.expunge(v)
            return v
        finally:
            session.close()

    def get_video_by_filename(self, filename):                  # db.py:104-109
        session = self.SessionLocal()
        try:
            v = session.query(Video).filter_by(filename=filename).first()
            if v is not None:
                session.expunge(v)
            return v
        finally:
            session.close()

    def clear(self):                                            # app.py:325-333
        try:
            self.flush()
        except Exception:
            pass
        with self._write_lock:
            session = self.SessionLocal()
            try:
                session.query(VideoTimestamps).delete()
                session.query(Video).delete()
                session.commit()
            finally:
                session.close()
            self.corpus.clear()
            self._census = (0, 0)
            self._sql_digest = {}

    def list_videos(self):                                      # app.py:347-366
        session = self.SessionLocal()
        try:
            out = []
            for video in session.query(Video).order_by(Video.id).all():
                ts = (session.query(VideoTimestamps).filter_by(video_id=video.id)
                      .order_by(VideoTimestamps.id).first())
                out.append({"id": video.id, "filename": video.filename,
                            "upload_time": video.upload_time.isoformat() if video.upload_time else None,
                            "duplicates": video.duplicates,
                            "timestamps": ts.timestamps if ts else []})
            return out
        finally:
            session.close()

    def close(self):
        try:
            self.flush()
        except Exception:
            pass
        with self._wb_cv:
            self._wb_stop = True
            self._wb_cv.notify_all()
        if self._wb_thread is not None:
            self._wb_thread.join(timeout=5)
        self._audit_stop.set()
        if self._audit_thread is not None:
            self._audit_thread.join(timeout=5)
        self.corpus.close()
        self.engine.dispose()


def create_schema(url: Optional[str] = None) -> None:
    """db.py:30 (`Base.metadata.create_all`) on its own: the parent of the N-rank service makes the
    tables once, before the rank processes open the same database side by side."""
    url = url or os.environ.get("POSTGRES_URL", DEFAULT_URL)
    eng = create_engine(url)
    try:
        Base.metadata.create_all(eng)
    finally:
        eng.dispose()


def video_filenames(url: Optional[str] = None) -> dict:
    """{video id: file name} of `videos` (the N-rank service routes by file name: service.owner_rank)."""
    from sqlalchemy import select
    url = url or os.environ.get("POSTGRES_URL", DEFAULT_URL)
    eng = create_engine(url)
    try:
        with eng.connect() as c:
            return {int(i): f for i, f in c.execute(select(Video.id, Video.filename))}
    finally:
        eng.dispose()


# ---- module-level functions, as the reference exports them (app.py:10) --------------------
_default: Optional[Store] = None
_default_lock = threading.Lock()


def init(url: Optional[str] = None, device: int = 0, corpus=None) -> Store:
    global _default
    with _default_lock:
        if _default is not None:
            _default.close()
        _default = Store(url, device, corpus)
        return _default


def store() -> Store:
    global _default
    if _default is None:
        with _default_lock:
            if _default is None:
                _default = Store()
    return _default


def add_video(filename, thumbnail_path=None):
    return store().add_video(filename, thumbnail_path)


def add_timestamps(video_id, timestamps):
    return store().add_timestamps(video_id, timestamps)


def update_duplicates(video_id, duplicate_ids):
    return store().update_duplicates(video_id, duplicate_ids)


def find_duplicates(new_timestamps, min_match=5):
    return store().find_duplicates(new_timestamps, min_match)


def get_video_by_id(video_id):
    return store().get_video_by_id(video_id)


def get_video_by_filename(filename):
    return store().get_video_by_filename(filename)
