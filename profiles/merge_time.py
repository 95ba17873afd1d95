#!/usr/bin/env python3
"""Kernel time of the merge of N gathered per-rank blocks [N][Q][k+1][3] (tvz_topk_merge), N = 1, 2, 4, 8, Q = 4096, k = 16:
run under `rocprofv3 --kernel-trace --stats` and read ts_topk_merge_sorted_kernel<G>'s rows (G = N rounded up to a power of
two); the script itself prints event-timed medians of 20 back-to-back calls (launch overhead amortised).
   [TVZ_LIB=variants/libtvz_x.so] python3 profiles/merge_time.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, sharded, synth  # noqa: E402

dev = torch.device("cuda:0")
C, Q, K = 100000, 4096, 16
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
d_q, d_off, ml = tc.pack_queries(synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1), dev)
blocks = []
for r in range(8):
    dc = tc.DeviceCorpus(0)
    dc.upload_csr(*sharded.shard_csr(ids, offs, keys, r, 8))
    blocks.append(dc.match_topk(d_q, d_off, ml, 2, 16384, K).clone())
    dc.close()
res = {}
for N in (1, 2, 4, 8):
    g = torch.stack(blocks[:N]).contiguous()
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out = tc.topk_merge(g, K)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    res[N] = round(float(np.median(ts[2:])) * 1e3, 2)
print(json.dumps({"lib": os.environ.get("TVZ_LIB", "product"), "Q": Q, "k": K, "us_per_merge_back_to_back": res}))
