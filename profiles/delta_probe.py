#!/usr/bin/env python3
"""What the delta table costs a batched match: C=100k indexed rows + D rows upserted since the build
(below the rebuild trigger), Q queries, event-timed tvz_match (AUTO).   python profiles/delta_probe.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, synth  # noqa: E402

dev = torch.device("cuda:0")
C = 100000
ids, offs, keys = synth.synth_timestamp_corpus(C + 12000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids[:C], offs[:C + 1], keys[:offs[C]], 4096, seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
dc.reserve(C + 20000, int(offs[-1]) + 1000000)
dc.upload_csr(ids[:C], offs[:C + 1], keys[:offs[C]])
st = torch.cuda.Stream(dev)
out = []
done = 0
for D in (0, 64, 1024, 6000, 12000):
    for c in range(C + done, C + D):
        dc.upsert(int(ids[c]), keys[offs[c]:offs[c + 1]])
    done = D
    stats = dc.index_stats()
    for Q in (256, 4096):
        d_q, d_off, ml = tc.pack_queries(queries[:Q], dev)
        ws = torch.empty(tc.workspace_bytes(Q, ml), dtype=torch.uint8, device=dev)
        hits = torch.empty((Q, 16384, 3), dtype=torch.int32, device=dev)
        n = torch.empty(Q, dtype=torch.int32, device=dev)
        ts = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            dc.match(d_q, d_off, ml, 2, 16384, out_hits=hits, out_n=n, stream=st, workspace=ws)
            b.record(st)
            st.synchronize()
            ts.append(a.elapsed_time(b))
        out.append({"delta_rows": stats["delta_rows"], "indexed_rows": stats["indexed_rows"], "Q": Q,
                    "ms": round(float(np.median(ts[2:])), 4)})
print(json.dumps(out))
