#!/usr/bin/env python3
"""tvz_topk_merge of R gathered blocks alone (Q = 4096, k = 16), for a kernel trace:  python profiles/merge_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, sharded, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, 4096, seed=synth.CORPUS_SEED + 1)
d_q, d_off, ml = tc.pack_queries(queries, dev)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
blk = dc.match_topk(d_q, d_off, ml, 2, 16384, 16).clone()
for R in (1, 2, 4, 8, 16):
    g = blk.unsqueeze(0).repeat(R, 1, 1, 1).contiguous()
    for _ in range(reps):
        tc.topk_merge(g, 16)
    torch.cuda.synchronize()
dc.close()
