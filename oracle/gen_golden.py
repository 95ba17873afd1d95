#!/usr/bin/env python3
"""Generate tests/golden/match_*.json by EXECUTING the reference matcher.

Runs only in the build container (needs /root/reference).  The reference's
inspector/db.py is imported unmodified; three harness steps make that possible
without Postgres (none edits the reference):
  1. POSTGRES_URL=sqlite://      so create_engine needs no psycopg2 (db.py:7-8)
  2. MetaData.create_all is a no-op while `import db` runs (db.py:30 would try
     to render PG ARRAY columns on SQLite)
  3. db.SessionLocal is replaced by a fake whose .query(VideoTimestamps).all()
     returns row objects with .video_id / .timestamps (what db.py:83-91 reads)

Only DATA is written: corpus, query, min_match, and the reference's output.
The streaming cases replay the loop of inspector/app.py:228-255 around the
reference's find_duplicates (upsert-then-match on every new prefix, drop self,
stop at first hit).

Usage: python oracle/gen_golden.py     (rewrites tests/golden/match_*.json)
"""
import json
import math
import os
import random
import sys

REF = "/root/reference/inspector"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference_db():
    os.environ["POSTGRES_URL"] = "sqlite://"
    sys.dont_write_bytecode = True
    import sqlalchemy
    orig = sqlalchemy.MetaData.create_all
    sqlalchemy.MetaData.create_all = lambda self, *a, **k: None
    sys.path.insert(0, REF)
    try:
        import db  # the reference module, unmodified
    finally:
        sqlalchemy.MetaData.create_all = orig
        sys.path.remove(REF)
    return db


class Row:
    def __init__(self, video_id, timestamps):
        self.video_id = video_id
        self.timestamps = timestamps


class FakeSession:
    rows = []

    def query(self, model):
        return self

    def all(self):
        return list(FakeSession.rows)

    def close(self):
        pass


def ref_find(db, corpus, query, min_match):
    FakeSession.rows = [Row(v, list(t)) for v, t in corpus]
    return sorted([int(a), int(b)] for a, b in db.find_duplicates(list(query), min_match=min_match))


def ref_streaming(db, corpus, stream, self_id, min_match):
    """app.py:228-255 replay; corpus is a list of [vid, ts] (self row appended if absent)."""
    rows = [Row(v, list(t)) for v, t in corpus]
    me = next((r for r in rows if r.video_id == self_id), None)
    scene = []
    for ts in stream:
        if not scene or ts != scene[-1]:
            scene.append(ts)
            if me is None:
                me = Row(self_id, [])
                rows.append(me)
            me.timestamps = list(scene)
            FakeSession.rows = rows
            dups = db.find_duplicates(scene, min_match=min_match)
            dups = [d for d in dups if d[0] != self_id]
            if dups:
                return scene, sorted(int(d[0]) for d in dups), sorted([int(a), int(b)] for a, b in dups)
    return scene, [], []


def g6(x):
    return float("%.6g" % x)


def synth_video(rng, fps=None, n=None, dur=None):
    fps = fps or rng.choice([24, 25, 30])
    dur = dur or rng.uniform(60, 600)
    nframes = int(dur * fps)
    n = n or max(5, min(80, int(rng.gauss(40, 8))))
    n = min(n, nframes - 1)
    frames = sorted(rng.sample(range(1, nframes), n))
    return [g6((1.0 / fps) * f) for f in frames]


def main():
    db = load_reference_db()
    FakeSessionFactory = lambda: FakeSession()
    db.SessionLocal = FakeSessionFactory
    os.makedirs(OUT, exist_ok=True)
    nan = float("nan")

    # ---- known answers and hand-made edge cases -------------------------
    kat = []

    def add(name, corpus, query, mm):
        kat.append({"name": name, "corpus": corpus, "query": query, "min_match": mm,
                    "expected": ref_find(db, corpus, query, mm)})

    c2 = [[1, [1.0, 2.0, 3.0, 4.0, 5.0]], [2, [10.0, 20.0, 30.0, 40.0, 50.0]]]
    add("test_app.py:66-77 disjoint", c2, [10.0, 20.0, 30.0, 40.0, 50.0], 5)
    c3 = c2 + [[3, [1.0, 2.0, 3.0, 4.0, 5.0]]]
    add("test_app.py:78-83 third identical", c3, [1.0, 2.0, 3.0, 4.0, 5.0], 5)
    add("app.py:399-408 debug scenario", [[1, [1.2, 5.7, 12.3, 18.9]]], [1.2, 5.7, 12.3, 18.9], 2)
    add("app.py:372 default vector", [[7, [1.2, 5.7, 12.3, 18.9, 25.1]]], [1.2, 5.7], 2)
    # sample rows of docs/tvidz-detailed-guide.md:1268-1272 ("Partial match with video 1")
    doc = [[1, [1.2, 5.7, 12.3, 18.9, 25.1]], [2, [2.1, 8.4, 15.7, 22.1, 28.9]], [3, [1.2, 5.7, 12.3]]]
    add("guide.md:1268-1272 sample rows, query = video 3", doc, [1.2, 5.7, 12.3], 2)
    add("guide.md:1268-1272 sample rows, query = video 1, default min_match", doc, [1.2, 5.7, 12.3, 18.9, 25.1], 5)
    add("guide.md:1268-1272 sample rows, query = video 2", doc, [2.1, 8.4, 15.7, 22.1, 28.9], 3)
    add("query multiplicity counts", [[1, [1.2, 9.9]]], [1.2, 1.2], 2)
    add("candidate multiplicity does not", [[1, [1.2, 1.2, 1.2]]], [1.2], 2)
    add("candidate multiplicity min1", [[1, [1.2, 1.2, 1.2]]], [1.2], 1)
    add("exactness", [[1, [1.2, 5.7]]], [1.2000001, 5.7], 2)
    add("exactness ulp", [[1, [0.1 + 0.2, 5.7]]], [0.3, 5.7], 2)
    add("min_match 0 returns all", c3, [], 0)
    add("min_match 0 nonempty", c3, [1.0, 99.0], 0)
    add("empty query min1", c3, [], 1)
    add("empty candidate", [[1, []], [2, [3.0]]], [3.0], 1)
    add("empty candidate min0", [[1, []], [2, [3.0]]], [3.0], 0)
    add("empty corpus", [], [1.0, 2.0], 1)
    add("negative zero", [[1, [0.0, 1.0]], [2, [-0.0, 2.0]]], [-0.0, 0.0], 1)
    add("infinities", [[1, [math.inf, 1.0]], [2, [-math.inf]]], [math.inf, -math.inf, 1.0], 1)
    add("int-valued and large", [[1, [1e15, 3.0, 1e-300]], [2, [4503599627370497.0]]],
        [1e15, 1e-300, 4503599627370497.0], 1)
    add("unsorted candidate", [[5, [9.5, 1.5, 7.25, 3.0]]], [3.0, 9.5, 4.0], 2)
    add("duplicate video_id rows", [[4, [1.0, 2.0]], [4, [2.0, 3.0]]], [2.0, 3.0], 1)
    add("negative min_match", c3, [1.0], -3)
    add("subnormal", [[1, [5e-324, 1.0]]], [5e-324, 0.0], 1)
    kat_nan = {"name": "nan never matches (distinct objects)", "corpus_has_nan_at": [[0, 1]],
               "corpus": [[1, [1.0, None, 2.0]]], "query": [None, 1.0], "query_has_nan_at": [0],
               "min_match": 1}
    FakeSession.rows = [Row(1, [1.0, float("nan"), 2.0])]
    kat_nan["expected"] = sorted([int(a), int(b)] for a, b in
                                 db.find_duplicates([float("nan"), 1.0], min_match=1))
    with open(os.path.join(OUT, "match_kat.json"), "w") as f:
        json.dump({"source": "reference inspector/db.py find_duplicates executed by oracle/gen_golden.py",
                   "cases": kat, "nan_case": kat_nan}, f, indent=1)

    # ---- seeded random corpora -----------------------------------------
    rng = random.Random(20250815)
    rnd = []
    for case in range(6):
        C = [40, 120, 300, 300, 64, 17][case]
        corpus = []
        for v in range(C):
            vid = 1000 + v if case != 3 else rng.randrange(1, 10**6)
            corpus.append([vid, synth_video(rng, fps=30 if case < 2 else None,
                                             dur=120 if case < 2 else None)])
        # make some exact copies, prefixes and shuffled rows
        for _ in range(max(1, C // 20)):
            a, b = rng.randrange(C), rng.randrange(C)
            if a != b:
                corpus[b][1] = list(corpus[a][1])
        for _ in range(max(1, C // 20)):
            a, b = rng.randrange(C), rng.randrange(C)
            if a != b:
                corpus[b][1] = list(corpus[a][1][: max(1, len(corpus[a][1]) // 2)])
        rng.shuffle(corpus[rng.randrange(C)][1])
        queries = []
        src = corpus[rng.randrange(C)][1]
        queries.append(list(src))
        queries.append(list(src[: len(src) // 3]))
        queries.append(synth_video(rng, fps=30, dur=120))
        q = list(src[:10]) + synth_video(rng)[:10]
        rng.shuffle(q)
        queries.append(q + q[:3])  # multiplicity
        for qi, q in enumerate(queries):
            for mm in ([1, 2, 5] if qi < 2 else [2, 3]):
                rnd.append({"name": f"rand{case}_q{qi}_mm{mm}", "corpus_ref": case, "query": q,
                            "min_match": mm, "expected": None})
        rnd.append({"corpus_def": case, "corpus": corpus})
    # resolve expected with the reference
    corpora = {e["corpus_def"]: e["corpus"] for e in rnd if "corpus_def" in e}
    for e in rnd:
        if "corpus_ref" in e:
            e["expected"] = ref_find(db, corpora[e["corpus_ref"]], e["query"], e["min_match"])
    with open(os.path.join(OUT, "match_random.json"), "w") as f:
        json.dump({"source": "reference inspector/db.py find_duplicates executed by oracle/gen_golden.py; seed 20250815",
                   "corpora": {str(k): v for k, v in corpora.items()},
                   "cases": [e for e in rnd if "corpus_ref" in e]}, f)

    # ---- streaming verdict cases (app.py:228-255 replay) -----------------
    rng = random.Random(1234)
    stream_cases = []
    for case in range(10):
        C = 150
        corpus = [[v + 1, synth_video(rng, fps=30, dur=90)] for v in range(C)]
        self_id = C + 1
        kind = case % 5
        if kind == 0:      # exact duplicate of an existing video
            stream = list(corpus[rng.randrange(C)][1])
        elif kind == 1:    # fresh video (accidental min_match=2 hits expected on a 30fps/90s grid)
            stream = synth_video(rng, fps=30, dur=90)
        elif kind == 2:    # consecutive repeated timestamps in the stream (app.py:231 drops them)
            base = list(corpus[rng.randrange(C)][1])
            stream = []
            for x in base:
                stream += [x] * rng.choice([1, 1, 2, 3])
        elif kind == 3:    # self row already present (re-analysis), duplicate of two others
            src = list(corpus[3][1])
            corpus[77][1] = list(src)
            corpus.append([self_id, [0.5]])
            stream = src
        else:              # no duplicates at all: disjoint timeline, higher min_match
            stream = [g6(1000 + 0.37 * i) for i in range(30)]
        mm = 2 if kind != 4 else 3
        scene, dup_ids, dups = ref_streaming(db, corpus, stream, self_id, mm)
        stream_cases.append({"name": f"stream{case}_kind{kind}", "corpus": corpus, "stream": stream,
                             "self_id": self_id, "min_match": mm, "scene_timestamps": scene,
                             "dup_ids": dup_ids, "dups": dups})
    with open(os.path.join(OUT, "match_streaming.json"), "w") as f:
        json.dump({"source": "inspector/app.py:228-255 loop replayed around the reference's db.find_duplicates; seed 1234",
                   "cases": stream_cases}, f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
