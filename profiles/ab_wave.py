#!/usr/bin/env python3
"""A/B of the lookup that keeps the top-k on rank 0's 1/8 shard of the 100k-video table (one sub-index),
Q = 4096, 8 query batches rotating: block per query pair (round 4's shape), block per query, ONE WAVE per
query (ts_match_wq_topk_kernel).  Event-timed per call, interleaved A/B/C rounds in one process.
   python3 profiles/ab_wave.py [reps] [min_match] [n_shards]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, sharded, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MM = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
C, Q = 100000, 4096
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, NS))
batches = [tc.pack_queries(synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b), dev) for b in range(8)]
ml = max(b[2] for b in batches)
CAP = 16384
ws = torch.empty(tc.workspace_bytes(Q, ml, CAP, 16), dtype=torch.uint8, device=dev)
out = torch.empty((Q, 17, 3), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(dev)
shapes = {"block_pair": _lib.ALGO_PAIR, "block_one": _lib.ALGO_NO_PAIR | _lib.ALGO_NO_WAVE, "wave": _lib.ALGO_WAVE}
if dc.index_stats()["indexed_rows"] > 16384:
    shapes.pop("wave")
ts = {k: [] for k in shapes}
ref = None
for r in range(reps):
    for name, fl in shapes.items():
        b = batches[r % 8]
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        dc.match_topk(b[0], b[1], ml, MM, CAP, 16, out=out, workspace=ws, stream=st, algo=fl)
        e.record(st)
        st.synchronize()
        ts[name].append(a.elapsed_time(e))
        if r == 0:
            if ref is None:
                ref = out.clone()
            else:
                assert (out == ref).all(), name
print(json.dumps({"workload": f"rank 0 of {NS} shards of {C} videos x {Q} queries, min_match {MM}, max_len {ml}",
                  "index": dc.index_stats(),
                  "median_us": {k: round(float(np.median(v[4:])) * 1e3, 1) for k, v in ts.items()},
                  "p10_us": {k: round(float(np.percentile(v[4:], 10)) * 1e3, 1) for k, v in ts.items()}}))
dc.close()
