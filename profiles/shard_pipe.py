#!/usr/bin/env python3
"""The pipelined sharded batch (tvz_match_sharded with a one-rank communicator, 2 and 3 batches in flight, 8 query
batches rotating) on rank 0's 1/N shard of the 100k-video table, per shape of the lookup that keeps the top-k:
   [TVZ_LIB=variants/libtvz_x.so] python3 profiles/shard_pipe.py [n_shards] [steps]
What a GPU of configs[3] sustains per batch when the next batches' lookups fill the tails of this one's."""
import json
import os
import sys
import time
from collections import deque

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, sharded, synth  # noqa: E402

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
C, Q = 100000, 4096
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, NS))
batches = [tc.pack_queries(synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b), dev) for b in range(8)]
ml = max(b[2] for b in batches)
comm = sharded.make_comm(0)
res = {}
shapes = {"auto": 0, "block": _lib.ALGO_NO_WAVE}      # (auto: RcclShardedMatcher adds ALGO_PREFER_WAVE)
if dc.index_stats()["indexed_rows"] <= 16384:
    shapes["wave"] = _lib.ALGO_WAVE
for name, fl in shapes.items():
    for depth in (2, 3):
        sm = sharded.RcclShardedMatcher(dc, comm, k=16, cap=16384, n_streams=depth, algo=fl)
        for i in range(2 * depth + 2):
            sm.match_topk(batches[i % 8][0], batches[i % 8][1], ml, 2)
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            inflight = deque()
            for i in range(STEPS):
                b = batches[i % 8]
                inflight.append(sm.submit(b[0], b[1], ml, 2, inputs_ready=True))
                if len(inflight) >= depth:
                    sm.finish(inflight.popleft(), host=True)
            while inflight:
                sm.finish(inflight.popleft(), host=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e6 / STEPS
            best = dt if best is None else min(best, dt)
        res[f"{name}_depth{depth}_us"] = round(best, 1)
print(json.dumps({"lib": os.environ.get("TVZ_LIB", "product"), "shards": NS, "index": dc.index_stats(), "us_per_batch": res}))
comm.close()
dc.close()
