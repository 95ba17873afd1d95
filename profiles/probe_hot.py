import os, sys, json
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from tvidz_amd import _lib, corpus as tc, synth, sharded
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, 4096, seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
st = torch.cuda.Stream(dev)
out = {}
for name, qs in (("distinct", queries), ("same_query_4096x", [queries[0]] * 4096), ("64_queries_repeated", [queries[i % 64] for i in range(4096)])):
    d_q, d_off, ml = tc.pack_queries(qs, dev)
    hits = torch.empty((4096, 16384, 3), dtype=torch.int32, device=dev); n = torch.empty(4096, dtype=torch.int32, device=dev)
    ws = torch.empty(tc.workspace_bytes(4096, ml), dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); dc.match(d_q, d_off, ml, 2, 16384, out_hits=hits, out_n=n, stream=st, workspace=ws, algo=_lib.ALGO_INDEX); b.record(st)
        st.synchronize(); ts.append(a.elapsed_time(b))
    out[name] = round(float(np.median(ts[2:])) * 1e3, 1)
print(json.dumps(out))
