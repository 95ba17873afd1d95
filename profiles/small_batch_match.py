import os, sys, json, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tvidz_amd import corpus as tc, synth
dev = torch.device("cuda:0")
for C in (5000, 100000):
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
    queries = synth.synth_queries(ids, offs, keys, 64, seed=synth.CORPUS_SEED + 1)
    dc = tc.DeviceCorpus(0); dc.upload_csr(ids, offs, keys)
    for Q in (1, 4, 16, 64):
        d_q, d_off, ml = tc.pack_queries(queries[:Q], dev)
        hits = torch.empty((Q, 1024, 3), dtype=torch.int32, device=dev); n = torch.empty(Q, dtype=torch.int32, device=dev)
        ts = []
        for r in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dc.match(d_q, d_off, ml, 2, 1024, out_hits=hits, out_n=n); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        lat = []
        for i in range(30):
            t = time.perf_counter(); dc.find_duplicates(queries[i % 64], 2); lat.append(time.perf_counter() - t)
        print(json.dumps({"C": C, "Q": Q, "kernel_us": round(float(np.median(ts[5:])) * 1e3, 1), "find_duplicates_us": round(float(np.median(lat)) * 1e6, 1)}))
    dc.close()
