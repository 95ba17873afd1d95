#!/usr/bin/env python3
"""A/B in one process: LDS tile kernel vs hash join vs per-query sweep over corpus sizes and batch
sizes; the kernel is chosen PER CALL (tvz.h TVZ_ALGO_*)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, synth
lib = _lib.load(); dev = torch.device("cuda:0")
for C in (5000, 20000, 50000, 100000):
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
    dc = tc.DeviceCorpus(0); dc.upload_csr(ids, offs, keys)
    for Q in (16, 64, 256, 1024):
        queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
        d_q, d_off, max_len = tc.pack_queries(queries, dev)
        hits = torch.empty((Q, 1024, 3), dtype=torch.int32, device=dev); n = torch.empty(Q, dtype=torch.int32, device=dev)
        ws = torch.empty(tc.workspace_bytes(Q, max_len), dtype=torch.uint8, device=dev)
        res = {}
        modes = [_lib.ALGO_TILE, _lib.ALGO_JOIN] + ([_lib.ALGO_Q1] if Q <= 64 else [])
        for rnd in range(6):
            for mode in modes:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); dc.match(d_q, d_off, max_len, 2, 1024, out_hits=hits, out_n=n, workspace=ws, algo=mode); b.record()
                torch.cuda.synchronize()
                if rnd: res.setdefault(mode, []).append(a.elapsed_time(b))
        t0, t2 = float(np.median(res[_lib.ALGO_TILE])), float(np.median(res[_lib.ALGO_JOIN]))
        row = {"C": C, "Q": Q, "tile_ms": round(t0, 4), "join_ms": round(t2, 4),
               "tile_Gpairs": round(Q * C / t0 / 1e6, 1), "join_Gpairs": round(Q * C / t2 / 1e6, 1)}
        if _lib.ALGO_Q1 in res:
            t1 = float(np.median(res[_lib.ALGO_Q1]))
            row.update(q1_ms=round(t1, 4), q1_Gpairs=round(Q * C / t1 / 1e6, 1))
        print(json.dumps(row))
    dc.close()
