#!/usr/bin/env python3
"""End-to-end service throughput proxy for BASELINE.json configs[4] (concurrent uploads through the
driver): N distinct synthetic 1080p clips as mono Y4M files in RAM -> Inspector.submit (Y4M reader
-> pinned ring -> H2D -> HIP scene kernels -> HIP corpus match per micro-batch -> SQL upserts).
No decoder in the loop (the host-side ffmpeg decode is outside the hot path); PCIe-inclusive.
   python profiles/e2e_service.py [n_uploads] [frames_per_clip] [workers] [batch] [slot_MiB] [switch] [H] [W] [shards]
H W default to 1080p; 2160 3840 = configs[4]'s 4K uploads.  shards > 0: the table in that many shards
(service.ShardedCorpus), every ask through the tick-batched sharded match.  shards = -1: what ONE rank of
`python -m tvidz_amd.service --ranks N` runs - service.RankCorpus over RcclShardedMatcher (tvz_match_sharded
with a one-rank communicator: the collective tick exchange degenerates to this process, everything else is
the N-rank path)."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import db as tdb, feeder, inspector as insp, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
WORKERS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 256
SLOT_MB = int(sys.argv[5]) if len(sys.argv) > 5 else 64
SWITCH = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
if SWITCH > 0:
    sys.setswitchinterval(SWITCH)
H = int(sys.argv[7]) if len(sys.argv) > 7 else 1080
W = int(sys.argv[8]) if len(sys.argv) > 8 else 1920
SHARDS = int(sys.argv[9]) if len(sys.argv) > 9 else 0
root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (N * T * H * W * 1.2) else None
tmp = tempfile.mkdtemp(prefix="tvz_e2e_", dir=root)
try:
    files = {}
    for i in range(N):
        frames, _ = synth.synth_luma(T, H, W, device="cuda:0", seed=100 + i, min_scene=20, max_scene=90)
        name = f"17000000{i:02d}-clip{i}.y4m"
        files[name] = os.path.join(tmp, name)
        feeder.write_y4m(files[name], frames.cpu().numpy(), fps=(30, 1), chroma="mono")
        del frames
    corpus = None
    if SHARDS > 0:
        from tvidz_amd import service
        corpus = service.ShardedCorpus(0, n_shards=SHARDS, k=16)
    elif SHARDS < 0:
        from tvidz_amd import corpus as tc, service, sharded
        shard = tc.DeviceCorpus(0)
        comm = sharded.make_comm(0)
        corpus = service.RankCorpus(shard, sharded.RcclShardedMatcher(shard, comm, k=16, cap=4096, priority=-1), xdev="cpu")   # asks are exchanged on the host
    store = tdb.Store(f"sqlite:///{tmp}/tvidz.db", device=0, corpus=corpus, census=SHARDS >= 0)
    ids, offs, keys = synth.synth_timestamp_corpus(5000, seed=1)
    lib_rows = [(int(ids[c]) + 100000, keys[offs[c]:offs[c + 1]].tolist()) for c in range(len(ids))]
    store.corpus.upload(lib_rows)                         # a 5k-video corpus to match against

    def source(bucket, key, filename, unique_id):
        return feeder.Y4MReader(files[key]), None

    ins = insp.Inspector(store, device="cuda:0", frame_source=source, batch=BATCH, max_workers=WORKERS,
                         slot_bytes=SLOT_MB << 20, profile=True)
    # warm-up pass: a long-running service has its pinned/device slots cached by the allocators
    [f.result() for f in [ins.submit("videos", k) for k in files]]
    store.clear()
    store.corpus.upload(lib_rows)
    ins.phase_seconds.clear()
    scored0 = ins.frames_scored
    t0 = time.perf_counter()
    futs = [ins.submit("videos", k) for k in files]
    res = [f.result() for f in futs]
    dt = time.perf_counter() - t0
    assert all(r["status"] == "done" for r in res), [r.get("error") for r in res]
    frames_done = ins.frames_scored - scored0   # an upload stops at its duplicate verdict (inspector/app.py:249-255)
    print(json.dumps({"uploads": N, "frames_per_clip": T, "height": H, "width": W, "shards": SHARDS, "workers": WORKERS, "batch": BATCH, "wall_s": round(dt, 3),
                      "frames_per_s": round(frames_done / dt), "GBps_luma": round(frames_done * H * W / dt / 1e9, 2),
                      "frames_scored": frames_done, "frames_submitted": N * T, "submitted_frames_per_s": round(N * T / dt),
                      "slot_MiB": SLOT_MB, "switch_interval": sys.getswitchinterval(),
                      "cuts_total": sum(r["total_cuts"] for r in res), "dups_total": sum(len(r["duplicates"]) for r in res),
                      "phase_thread_seconds": {k: (round(v, 3) if isinstance(v, float) else v)
                                               for k, v in sorted(ins.phase_seconds.items()) if not k.startswith("n_")},
                      "tick": ({"ticks": corpus.batcher.ticks, "asks": corpus.batcher.asks, "exact_asks": corpus.exact_asks,
                                "tick_wall_s": round(corpus.tick_host_s, 4),
                                "us_per_tick": round(corpus.tick_host_s * 1e6 / max(corpus.batcher.ticks, 1), 1)} if SHARDS > 0 else
                               {"ticks": corpus.ticks, "busy_ticks": corpus.busy_ticks, "exact_asks": corpus.exact_asks,
                                "tick_wall_s": round(corpus.tick_host_s, 4),
                                "us_per_busy_tick": round(corpus.tick_host_s * 1e6 / max(corpus.busy_ticks, 1), 1)} if SHARDS < 0 else None),
                      "tmp": "shm" if root else "disk"}))
    store.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
