#!/usr/bin/env python3
"""What a tick of the sharded service costs with NOTHING else running: one handle
(tvz_find_duplicates), service.ShardedCorpus (8 shards, one tvz_match_topk_shards call per tick) and
service.RankCorpus over RcclShardedMatcher at world size 1 (what one rank of `python -m tvidz_amd.service`
runs), 5k-row corpus, asks of ~200 timestamps: 200 asks one after the other (latency of a lone ask) and
64 threads x 8 asks at once (asks share ticks).   python profiles/tick_cost.py"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, service, sharded, synth  # noqa: E402

ids, offs, keys = synth.synth_timestamp_corpus(5000, seed=1)
rows = [(int(ids[c]), keys[offs[c]:offs[c + 1]].tolist()) for c in range(len(ids))]
queries = synth.synth_queries(ids, offs, keys, 512, seed=3)


def measure(name, corpus, stats):
    corpus.upload(rows)
    for q in queries[:20]:
        corpus.find_duplicates(q, 2, exclude_id=-1, with_kth=True)
    lat = []
    for q in queries[:200]:
        t = time.perf_counter()
        corpus.find_duplicates(q, 2, exclude_id=-1, with_kth=True)
        lat.append(time.perf_counter() - t)
    before = stats()
    t0 = time.perf_counter()

    def worker(i):
        for j in range(8):
            corpus.find_duplicates(queries[(i * 8 + j) % 512], 2, exclude_id=-1, with_kth=True)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(64)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    after = stats()
    out = {"form": name, "lone_ask_us": {"median": round(float(np.median(lat)) * 1e6, 1), "p90": round(float(np.percentile(lat, 90)) * 1e6, 1)},
           "burst_512_asks_64_threads": {"wall_ms": round(wall * 1e3, 2), "asks_per_s": round(512 / wall)}}
    if after:
        ticks = after[0] - before[0]
        out["burst_512_asks_64_threads"].update({"ticks": ticks, "asks_per_tick": round(512 / max(ticks, 1), 1),
                                                 "tick_wall_us": round((after[1] - before[1]) * 1e6 / max(ticks, 1), 1)})
    print(json.dumps(out))
    corpus.close()


measure("one handle (tvz_find_duplicates)", tc.DeviceCorpus(0), lambda: None)
sc = service.ShardedCorpus(0, n_shards=8, k=16)
measure("ShardedCorpus, 8 shards", sc, lambda: (sc.batcher.ticks, sc.tick_host_s))
shard = tc.DeviceCorpus(0)
comm = sharded.make_comm(0)
rc = service.RankCorpus(shard, sharded.RcclShardedMatcher(shard, comm, k=16, cap=4096, priority=-1), xdev="cpu")
measure("RankCorpus over RcclShardedMatcher, world size 1", rc, lambda: (rc.busy_ticks, rc.tick_host_s))
comm.close()
