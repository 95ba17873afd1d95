#!/usr/bin/env python3
"""Timing probe for the corpus matcher in one process: python profiles/tune_match.py [C] [Q]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, synth  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
dc.upload_csr(ids, offs, keys)
d_q, d_off, max_len = tc.pack_queries(queries, dev)
hits = torch.empty((Q, 1024, 3), dtype=torch.int32, device=dev)
n = torch.empty(Q, dtype=torch.int32, device=dev)
for mm in (2, 3, 5, 1000, 2):
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dc.match(d_q, d_off, max_len, mm, 1024, out_hits=hits, out_n=n)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    med = float(np.median(ts[2:]))
    print(json.dumps({"C": C, "Q": Q, "min_match": mm, "median_ms": round(med, 4),
                      "Gpairs_s": round(Q * C / med / 1e6, 3), "hits_total": int(n.sum().item())}))
