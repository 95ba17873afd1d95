#!/usr/bin/env python3
"""Tile vs join vs per-query sweep for SMALL row tables (the delta table of an indexed corpus) under
LARGE batches: which sweep should follow the index lookup.   python profiles/ab_small_corpus.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, synth
dev = torch.device("cuda:0")
ids_all, offs_all, keys_all = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries_all = synth.synth_queries(ids_all, offs_all, keys_all, 4096, seed=synth.CORPUS_SEED + 1)
for C in (64, 256, 1024, 4096, 12000):
    ids, offs, keys = ids_all[:C], offs_all[:C + 1], keys_all[:offs_all[C]]
    dc = tc.DeviceCorpus(0); dc.upload_csr(ids, offs, keys)
    for Q in (64, 256, 1024, 4096):
        d_q, d_off, max_len = tc.pack_queries(queries_all[:Q], dev)
        hits = torch.empty((Q, 1024, 3), dtype=torch.int32, device=dev); n = torch.empty(Q, dtype=torch.int32, device=dev)
        ws = torch.empty(tc.workspace_bytes(Q, max_len), dtype=torch.uint8, device=dev)
        res = {}
        modes = [_lib.ALGO_TILE, _lib.ALGO_JOIN, _lib.ALGO_Q1]
        for rnd in range(6):
            for mode in modes:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); dc.match(d_q, d_off, max_len, 2, 1024, out_hits=hits, out_n=n, workspace=ws, algo=mode); b.record()
                torch.cuda.synchronize()
                if rnd: res.setdefault(mode, []).append(a.elapsed_time(b))
        print(json.dumps({"rows": C, "Q": Q, "tile_ms": round(float(np.median(res[_lib.ALGO_TILE])), 4),
                          "join_ms": round(float(np.median(res[_lib.ALGO_JOIN])), 4),
                          "q1_ms": round(float(np.median(res[_lib.ALGO_Q1])), 4)}))
    dc.close()
