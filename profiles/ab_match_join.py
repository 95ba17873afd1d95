#!/usr/bin/env python3
"""A/B in one process: LDS tile kernel (0) vs hash join (2) over corpus sizes and batch sizes."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, synth
lib = _lib.load(); dev = torch.device("cuda:0")
for C in (5000, 20000, 50000, 100000):
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
    dc = tc.DeviceCorpus(0); dc.upload_csr(ids, offs, keys)
    for Q in (64, 256, 1024):
        queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
        d_q, d_off, max_len = tc.pack_queries(queries, dev)
        hits = torch.empty((Q, 1024, 3), dtype=torch.int32, device=dev); n = torch.empty(Q, dtype=torch.int32, device=dev)
        res = {}
        for rnd in range(6):
            for mode in (0, 2):
                _lib.check(lib.tvz_match_set_tuning(mode))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); dc.match(d_q, d_off, max_len, 2, 1024, out_hits=hits, out_n=n); b.record()
                torch.cuda.synchronize()
                if rnd: res.setdefault(mode, []).append(a.elapsed_time(b))
        _lib.check(lib.tvz_match_set_tuning(1))
        t0, t2 = float(np.median(res[0])), float(np.median(res[2]))
        print(json.dumps({"C": C, "Q": Q, "tile_ms": round(t0, 4), "join_ms": round(t2, 4),
                          "tile_Gpairs": round(Q * C / t0 / 1e6, 1), "join_Gpairs": round(Q * C / t2 / 1e6, 1)}))
    dc.close()
