#!/usr/bin/env python3
"""The lookup with the top-k in its epilogue (tvz_match_topk on an indexed corpus) against the unfused
pipeline (tvz_match -> tvz_topk_shard) on the same handle: event-timed ms per call, Q queries against
the whole C-row corpus and against rank 0's 1/8 shard.  Batches ROTATE through `nb` distinct query
sets (a timed loop that replays one batch measures a warm cache).
    python profiles/topk_probe.py [Q] [C] [nb]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, sharded, synth  # noqa: E402

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K, CAP = 16, 16384
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
batches = []
for b in range(NB):
    qs = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b)
    batches.append(tc.pack_queries(qs, dev))
max_len = max(b[2] for b in batches)
st = torch.cuda.Stream(dev)


def timed(fn, reps=40, skip=8):
    ts = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(i); b.record(st)
        st.synchronize()
        ts.append(a.elapsed_time(b))
    ts = ts[skip:]
    return {"median_ms": round(float(np.median(ts)), 4), "p10": round(float(np.percentile(ts, 10)), 4),
            "p90": round(float(np.percentile(ts, 90)), 4), "first_ms": round(ts[0], 4)}


out = {"Q": Q, "C": C, "k": K, "cap": CAP, "distinct_batches": NB, "rows": []}
for N in (1, 8):
    s = sharded.shard_csr(ids, offs, keys, 0, N)
    dc = tc.DeviceCorpus(0)
    dc.upload_csr(*s)
    ws = torch.empty(tc.workspace_bytes(Q, max_len, CAP, K), dtype=torch.uint8, device=dev)
    hits = torch.empty((Q, CAP, 3), dtype=torch.int32, device=dev)
    n_h = torch.empty(Q, dtype=torch.int32, device=dev)
    blk = torch.empty((Q, K + 1, 3), dtype=torch.int32, device=dev)

    def fused(i):
        d_q, d_off, _ = batches[i % NB]
        dc.match_topk(d_q, d_off, max_len, 2, CAP, K, out=blk, workspace=ws, stream=st)

    def fused_same(i):
        d_q, d_off, _ = batches[0]
        dc.match_topk(d_q, d_off, max_len, 2, CAP, K, out=blk, workspace=ws, stream=st)

    def unfused(i):
        d_q, d_off, _ = batches[i % NB]
        dc.match(d_q, d_off, max_len, 2, CAP, out_hits=hits, out_n=n_h, stream=st, workspace=ws)
        tc.topk_shard(hits, n_h, K, stream=st)

    def lookup_only(i):
        d_q, d_off, _ = batches[i % NB]
        dc.match(d_q, d_off, max_len, 2, CAP, out_hits=hits, out_n=n_h, stream=st, workspace=ws)

    row = {"n_shards": N, "rows": int(len(s[0])),
           "fused_rotating": timed(fused), "fused_one_batch_replayed": timed(fused_same),
           "unfused_match_then_topk": timed(unfused), "unfused_lookup_alone": timed(lookup_only)}
    # same answer
    d_q, d_off, _ = batches[1]
    a = dc.match_topk(d_q, d_off, max_len, 2, CAP, K, workspace=ws).clone()
    h2, n2 = dc.match(d_q, d_off, max_len, 2, CAP)
    b2 = tc.topk_shard(h2, n2, K)
    torch.cuda.synchronize()
    row["identical_blocks"] = bool((a == b2).all().item())
    out["rows"].append(row)
    dc.close()
print(json.dumps(out))
