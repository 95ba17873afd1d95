"""`python -m tvidz_amd.service --ranks 2` on a CPU box: the parent spawns two rank processes (gloo; the
shards and the matcher backend are the oracle through `--parts tests.fakes:cpu_rank_parts`, the scene
half is a cut list carried in the file name), and the reference's surface is driven through the FRONT
over HTTP: POST /notify (inspector/app.py:31-44), GET /status/<f> (:46-62), the SSE stream (:64-115).
What is the product's and under test: the launcher, the routing by file name, the relays, RankCorpus's
tick exchange between the two processes, db.Store on one shared SQLite file, Inspector._after_cuts
(app.py:234-255).  Scenario of tests/test_config4_gpu.py in small: library videos ingested first, then
a burst of unique uploads, copies of library videos (flagged at their 2nd cut, whichever rank holds the
original), a twin pair, and 20 copies of one clip that tie at kth 1 with k = 4 (no upload may fail
because ties exceed k).  Expected records = the oracle's replay of app.py:228-255."""
import json
import os
import threading
import time

import pytest
import requests

from oracle import oracle
from tvidz_amd import service

PORT = 5600 + os.getpid() % 300


def _key(name, cuts, stamp=1700000000):
    return f"videos/{stamp}-{name}__{'_'.join(str(int(round(c * 10))) for c in cuts)}.mp4"


def _notify(base, key):
    r = requests.post(f"{base}/notify", json={"Records": [{"s3": {"bucket": {"name": "videos"}, "object": {"key": key}}}]},
                      timeout=30)
    assert r.status_code == 200 and r.json() == {"status": "Analysis started", "file": key}


def _wait_done(base, filename, timeout=60):
    deadline = time.time() + timeout
    while time.time() < deadline:
        rec = requests.get(f"{base}/status/{filename}", timeout=30).json()
        if rec.get("status") in ("done", "error"):
            return rec
        time.sleep(0.05)
    raise AssertionError(f"{filename} never finished")


def _sse_last(base, filename, out):
    with requests.get(f"{base}/status/stream/{filename}", stream=True, timeout=(10, 60)) as r:
        assert r.headers["Content-Type"].startswith("text/event-stream")
        assert r.headers["Access-Control-Allow-Origin"] == "*"
        last = None
        for line in r.iter_lines():
            if line.startswith(b"data: "):
                last = json.loads(line[6:])
                if last.get("status") in ("done", "error"):
                    break
        out[filename] = last


@pytest.fixture(scope="module")
def svc(tmp_path_factory):
    db_url = f"sqlite:///{tmp_path_factory.mktemp('svc')}/t.db"
    s = service.RankService(2, db_url, base_port=PORT, backend="gloo", parts="tests.fakes:cpu_rank_parts", k=4, cap=64,
                            workers=8, tick_s=0.002, ready_timeout=300,
                            env={"PYTHONPATH": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))})
    front = service.create_front(s.urls)
    from werkzeug.serving import make_server
    srv = make_server("127.0.0.1", PORT, front, threaded=True)
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    yield s, f"http://127.0.0.1:{PORT}"
    srv.shutdown()
    s.stop()


def test_two_rank_service_through_the_front(svc):
    s, base = svc
    assert requests.post(f"{base}/notify", json={"nope": 1}, timeout=10).status_code == 400     # app.py:39-40
    assert requests.get(f"{base}/status/never-seen.mp4", timeout=10).json() == {"status": "pending"}
    # ---- library: six videos, ingested one after the other; they land on both ranks by file name
    lib, per_rank, i = {}, [0, 0], 0
    while len(lib) < 6:                               # names chosen so that each rank owns three of them
        cuts = [10.0 * len(lib) + 1.5, 10.0 * len(lib) + 3.0, 10.0 * len(lib) + 4.5, 10.0 * len(lib) + 7.0]
        name, i = f"lib{len(lib)}v{i}", i + 1
        r = service.owner_rank(service.clean_name(_key(name, cuts)), 2)
        if per_rank[r] < 3:
            per_rank[r] += 1
            lib[name] = cuts
    table = []                                        # (video id, cuts) as SQL will hold them, in id order
    for name, cuts in lib.items():
        key = _key(name, cuts)
        _notify(base, key)
        rec = _wait_done(base, key.split("/")[-1])
        assert rec["status"] == "done" and rec["scene_cuts"] == cuts and rec["duplicates"] == []
        table.append((len(table) + 1, cuts))
    info = requests.get(f"{base}/ranks", timeout=10).json()["ranks"]
    assert [r["rank"] for r in info] == [0, 1] and sum(r["rows"] for r in info) == 6
    assert all(r["rows"] >= 1 for r in info), info    # the names route to both ranks
    owners = {name: service.owner_rank(service.clean_name(_key(name, c)), 2) for name, c in lib.items()}
    assert sorted(set(owners.values())) == [0, 1]
    # ---- the burst, all at once: uniques, copies of library videos, one twin pair
    uploads = {}
    for i in range(6):
        uploads[f"uniq{i}"] = [200.0 + 10 * i + 0.5, 200.0 + 10 * i + 2.5, 200.0 + 10 * i + 6.0]
    for i in (0, 2, 3, 5):
        src = list(lib)[i]
        uploads[f"copy_of_{src}"] = list(lib[src])
    keys = {n: _key(n, c, stamp=1700000100) for n, c in uploads.items()}
    sse = {}
    watchers = [threading.Thread(target=_sse_last, args=(base, k.split("/")[-1], sse)) for k in keys.values()]
    [w.start() for w in watchers]
    th = [threading.Thread(target=_notify, args=(base, k)) for k in keys.values()]
    [t.start() for t in th]
    [t.join(60) for t in th]
    [w.join(90) for w in watchers]
    for name, cuts in uploads.items():
        fn = keys[name].split("/")[-1]
        rec = _wait_done(base, fn)
        assert rec == sse[fn], (name, rec, sse.get(fn))           # the stream's last event IS the record
        assert rec["status"] == "done" and rec["original_filename"] == fn
        exp_ts, exp_dups = oracle.streaming_verdict_py(cuts, table, -1, 2)
        if name.startswith("copy_of_"):
            src = name[len("copy_of_"):]
            # flagged at its 2nd cut, the list truncated there, the original's CLEAN name reported (app.py:241-245)
            assert rec["scene_cuts"] == list(exp_ts) == cuts[:2] and rec["total_cuts"] == 2
            assert rec["duplicates"] == [service.clean_name(_key(src, lib[src]))]
        else:
            assert rec["scene_cuts"] == cuts and rec["duplicates"] == [] and not exp_dups
    # ---- 20 copies of one clip, one after the other: the LAST one ties with 19 rows at kth 1 with k = 4
    clip = [900.5, 901.5, 903.0, 904.0]
    for j in range(20):
        key = _key(f"tie{j}", clip, stamp=1700000200)
        _notify(base, key)
        rec = _wait_done(base, key.split("/")[-1])
        assert rec["status"] == "done", rec                      # never "more rows share the verdict's prefix than ..."
        assert len(rec["duplicates"]) == j and rec["total_cuts"] == (2 if j else 4)
    info = requests.get(f"{base}/ranks", timeout=10).json()["ranks"]
    assert sum(r["exact_asks"] for r in info) >= 1 and all(r["broken"] is None for r in info)
    assert s.dead() == []
    # the SQL table both ranks wrote: one row per upload, the duplicate ids of the last tie upload are the 19 others
    import sqlalchemy as sa
    eng = sa.create_engine(s_db(svc))
    with eng.connect() as c:
        n_videos = c.execute(sa.text("select count(*) from videos")).scalar()
        n_rows = c.execute(sa.text("select count(*) from video_timestamps")).scalar()
    assert n_videos == n_rows == 6 + len(uploads) + 20


def s_db(svc):
    return svc[0].procs[0].args[svc[0].procs[0].args.index("--db") + 1]


def test_a_dead_rank_is_noticed(svc):
    s, base = svc
    assert s.dead() == []
    s.procs[1].terminate()
    s.procs[1].wait(timeout=20)
    assert s.dead() == [1]                           # main()'s monitor thread stops the service on this


def test_rank_processes_notice_a_dead_launcher_under_a_subreaper(tmp_path):
    """the child watchdog compared os.getppid() with 1 - wrong both ways: a launcher that IS pid 1 (the
    reference's deployment: inspector/entrypoint.sh ends with `exec python app.py`) made every rank leave at once,
    and under a subreaper an orphan is re-parented to a pid other than 1 and never noticed.  Here the launcher runs
    under a subreaper (this harness), is killed, and its rank must leave with the watchdog's exit code 4."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = PORT + 400
    harness = f"""
import ctypes, os, signal, subprocess, sys, time, json, urllib.request
ctypes.CDLL(None).prctl(36, 1, 0, 0, 0)                       # PR_SET_CHILD_SUBREAPER
env = dict(os.environ, PYTHONPATH={root!r})
L = subprocess.Popen([sys.executable, "-m", "tvidz_amd.service", "--ranks", "1", "--port", "{port}", "--db",
                      "sqlite:///{tmp_path}/t.db", "--parts", "tests.fakes:cpu_rank_parts", "--k", "4", "--cap", "64"], env=env)
pid, deadline = None, time.time() + 240
while pid is None and time.time() < deadline:
    try:
        pid = json.load(urllib.request.urlopen("http://127.0.0.1:{port + 1}/rank-info", timeout=2))["pid"]
    except Exception:
        time.sleep(0.2)
assert pid, "the rank never came up"
assert L.poll() is None, "the launcher left although its rank's parent is alive (the pid-1 comparison)"
os.kill(L.pid, signal.SIGKILL)
L.wait()
deadline = time.time() + 20
while time.time() < deadline:
    got, status = os.waitpid(pid, os.WNOHANG)                 # re-parented to this subreaper, not to pid 1
    if got == pid:
        print("CHILD_EXIT", os.WEXITSTATUS(status) if os.WIFEXITED(status) else -1)
        break
    time.sleep(0.1)
else:
    os.kill(pid, signal.SIGKILL)
    print("CHILD_NEVER_LEFT")
"""
    out = subprocess.run([sys.executable, "-c", harness], capture_output=True, text=True, timeout=300)
    assert "CHILD_EXIT 4" in out.stdout, (out.stdout, out.stderr[-2000:])


def test_parent_gone_compares_with_the_launchers_pid_not_with_one():
    assert service.parent_gone(os.getppid()) is False          # also when os.getppid() == 1: a pid-1 launcher is alive
    assert service.parent_gone(os.getppid() + 1) is True
