#!/usr/bin/env python3
"""Index at scale: 1 M rows x ~50 keys (62 sub-indexes).  Upload + build time, one-query and batched
lookups compared with the forced sweep on the device (same hits), find_duplicates latency.
   python profiles/scale_probe.py [rows] [mean_len]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(9)
lens = np.clip(rng.normal(L, L / 8, C).round().astype(np.int64), 4, None)
offs = np.zeros(C + 1, dtype=np.int64); offs[1:] = np.cumsum(lens)
fps = rng.choice([24.0, 25.0, 30.0], size=C)
keys = np.empty(int(offs[-1]), dtype=np.float64)
frames = rng.integers(1, 7200 * 30, size=int(offs[-1]))
keys[:] = np.round(frames / np.repeat(fps, lens), 4)
ids = np.arange(1, C + 1, dtype=np.int32)
dev = torch.device("cuda:0")
dc = tc.DeviceCorpus(0)
t = time.perf_counter(); dc.upload_csr(ids, offs, keys); torch.cuda.synchronize(); up = time.perf_counter() - t
t = time.perf_counter(); dc.build_index(); build_first = time.perf_counter() - t   # sizes the shadow generation's buffers (hipMalloc)
t = time.perf_counter(); dc.build_index(); build_b = time.perf_counter() - t
t = time.perf_counter(); dc.build_index(); build = time.perf_counter() - t
st = dc.index_stats()
res = {"rows": C, "keys": int(offs[-1]), "upload_s": round(up, 3), "rebuild_index_s": round(build, 4), "rebuild_into_fresh_buffers_s": round(build_first, 4), "second_rebuild_s": round(build_b, 4), "index_stats": st}
queries = [keys[offs[r]:offs[r + 1]].copy() for r in rng.integers(0, C, 64)]
d_q, d_off, ml = tc.pack_queries(queries, dev)
cap = 65536
for name, algo in (("index", _lib.ALGO_INDEX), ("sweep", _lib.ALGO_TILE)):
    hits, n = dc.match(d_q, d_off, ml, 2, cap, algo=algo); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hits, n = dc.match(d_q, d_off, ml, 2, cap, algo=algo); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    h, nn = hits.cpu().numpy(), n.cpu().numpy()
    res[name] = {"ms_64_queries": round(float(np.median(ts)), 3), "hits": int(nn.sum())}
    res[name + "_sets"] = [sorted(map(tuple, h[q, :nn[q]].tolist())) for q in range(64)]
res["identical_hits"] = res.pop("index_sets") == res.pop("sweep_sets")
lat = []
for i in range(40):
    t = time.perf_counter(); dc.find_duplicates(queries[i % 64], 2); lat.append(time.perf_counter() - t)
res["find_duplicates_us"] = round(float(np.median(lat[5:])) * 1e6, 1)
print(json.dumps(res))
dc.close()
