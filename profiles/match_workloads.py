#!/usr/bin/env python3
"""Matcher workloads for rocprofv3 passes (kernel trace or PMC), one process per workload so a
counter row can be attributed by kernel name alone:
   python3 profiles/match_workloads.py <workload> [reps]
     join   : C=100k x Q=4096, min_match 2   -> ts_match_join_kernel (+ prep, build)   tag C100000_Q4096
     q1_100k: C=100k x Q=1                    -> ts_match_q1_kernel                     tag C100000_Q1
     q1_5k  : C=5k   x Q=1 + find_duplicates  -> ts_match_q1_kernel (both output modes) tag C5000_Q1
     tile   : C=100k x Q=64, min_match 2, forced LDS tile kernel                       tag C100000_Q64
     topk   : C=100k x Q=4096 match_topk (sweep + select top-k) + merge                tag C100000_Q4096
     shard8 : rank 0's 1/8 shard (12.5k rows) x Q=4096, match_topk + merge              tag C12500_Q4096
     index  : C=100k x Q=4096, inverted-index lookup (ts_match_index_kernel)           tag C100000_Q4096
     index1 : C=100k x Q=1 lookup + find_duplicates latency                            tag C100000_Q1
     index1_5k : the same at C=5k                                                      tag C5000_Q1
     rebuild: C=100k, `reps` x tvz_corpus_build_index after the upload (the upload's own build takes a
              second pass at the size its count revealed; a rebuild is one pass) + one batch of 64 queries
   topk / shard8 take the index too (AUTO); `join`, `tile`, `q1_*` force the sweep kernels.
Prints one JSON line with the event-timed median of the call."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "join"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
MM = int(sys.argv[3]) if len(sys.argv) > 3 else 2
# shape of the lookup that keeps the top-k (topk / shard8): 0 = the library's choice, 0x400 = one wave per query
# (tvz.h TVZ_ALGO_WAVE), 0x800 = never (TVZ_ALGO_NO_WAVE), 0x100 / 0x200 = two queries per block always / never
SHAPE = int(os.environ.get("TVZ_SHAPE", "0"), 0)
dev = torch.device("cuda:0")
C, Q, algo = {"join": (100000, 4096, _lib.ALGO_JOIN), "q1_100k": (100000, 1, _lib.ALGO_Q1),
              "q1_5k": (5000, 1, _lib.ALGO_Q1), "tile": (100000, 64, _lib.ALGO_TILE),
              "topk": (100000, 4096, _lib.ALGO_AUTO), "shard8": (100000, 4096, _lib.ALGO_AUTO),
              "index": (100000, 4096, _lib.ALGO_INDEX), "index1": (100000, 1, _lib.ALGO_INDEX),
              "index1_5k": (5000, 1, _lib.ALGO_INDEX), "rebuild": (100000, 64, _lib.ALGO_INDEX)}[which]
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, max(Q, 64), seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
if which == "shard8":          # rank 0's shard of an 8-way sharded corpus, the whole batch of queries
    from tvidz_amd import sharded
    dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
else:
    dc.upload_csr(ids, offs, keys)
rebuild_ms = []
if which == "rebuild":
    torch.cuda.synchronize()
    for _ in range(reps):
        t = time.perf_counter()
        dc.build_index()
        rebuild_ms.append((time.perf_counter() - t) * 1e3)
    reps = 3
d_q, d_off, ml = tc.pack_queries(queries[:Q], dev)
# the batched workloads ROTATE through 8 distinct query batches (counters and timings of
# a loop that replays one batch are warm-cache figures)
NB = 8 if Q >= 64 else 1
batches = [(d_q, d_off, ml)] + [tc.pack_queries(synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b), dev)
                                for b in range(1, NB)]
ml = max(b[2] for b in batches)
turn = {"i": 0}


def nxt():
    b = batches[turn["i"] % NB]
    turn["i"] += 1
    return b


CAP = 16384 if Q > 1 else 4096
hits = torch.empty((Q, CAP, 3), dtype=torch.int32, device=dev)
n = torch.empty(Q, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(dev)
ts = []
if which in ("topk", "shard8"):
    ws = torch.empty(tc.workspace_bytes(Q, ml, CAP, 16), dtype=torch.uint8, device=dev)
    out = torch.empty((Q, 17, 3), dtype=torch.int32, device=dev)
    def call():
        b = nxt()
        blk = dc.match_topk(b[0], b[1], ml, MM, CAP, 16, out=out, workspace=ws, stream=st, algo=SHAPE)
        tc.topk_merge(blk.view(1, Q, 17, 3), 16, stream=st)
else:
    ws = torch.empty(tc.workspace_bytes(Q, ml), dtype=torch.uint8, device=dev)

    def call():
        b = nxt()
        dc.match(b[0], b[1], ml, MM, CAP, out_hits=hits, out_n=n, stream=st, workspace=ws, algo=algo)
for r in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); call(); b.record(st)
    st.synchronize()
    ts.append(a.elapsed_time(b))
res = {"workload": which, "shape": hex(SHAPE), "C": C, "Q": Q, "min_match": MM, "median_ms": round(float(np.median(ts[2:])), 4),
       "first_call_ms_cold": round(ts[0], 4), "distinct_batches": NB,
       "hits": int(n.sum().item()) if which not in ("topk", "shard8") else None}
if which in ("q1_5k", "index1", "index1_5k"):     # find_duplicates takes the index when there is one
    lat = []
    for i in range(200):
        t = time.perf_counter()
        dc.find_duplicates(queries[i % 64], 2)
        lat.append(time.perf_counter() - t)
    res["find_duplicates_us"] = round(float(np.median(lat[20:])) * 1e6, 1)
    res["find_duplicates_p10_p90_us"] = [round(float(np.percentile(lat[20:], p)) * 1e6, 1) for p in (10, 90)]
if rebuild_ms:
    res["build_index_call_ms"] = {"median": round(float(np.median(rebuild_ms)), 3), "min": round(min(rebuild_ms), 3),
                                  "note": "host to host: drains readers, builds, reads the status back twice"}
res["index"] = dc.index_stats()
rows, nkeys, _ = dc.stats()
res["corpus_image_bytes"] = 16 * rows + 8 * nkeys
print(json.dumps(res))
dc.close()
