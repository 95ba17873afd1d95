#!/usr/bin/env python3
"""Hits per query on rank 0's 1/8 shard of the bench corpus (the totals row of tvz_match_topk): how many candidates the lookup's
slot phases see per query.   python3 profiles/shard_hits_hist.py"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tvidz_amd import _lib, corpus as tc, sharded, synth
dev = torch.device("cuda:0")
C, Q = 100000, 4096
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
res = {}
for b in range(2):
    d_q, d_off, ml = tc.pack_queries(synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b), dev)
    for mm in (2,):
        out = dc.match_topk(d_q, d_off, ml, mm, 16384, 16).cpu().numpy()
        tot = np.abs(out[:, 16, 1])
        res[f"batch{b}_mm{mm}"] = {"mean": float(tot.mean()), "p50": float(np.percentile(tot, 50)), "p90": float(np.percentile(tot, 90)), "p99": float(np.percentile(tot, 99)), "max": int(tot.max()),
                                  "share_over_512": float((tot > 512).mean()), "share_over_256": float((tot > 256).mean()), "share_over_1024": float((tot > 1024).mean()), "share_le_64": float((tot <= 64).mean())}
print(json.dumps(res))
