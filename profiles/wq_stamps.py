#!/usr/bin/env python3
"""Per-phase cycles of ts_match_wq_topk_kernel (every wave, s_memtime), diagnostic build:
   bash profiles/variant_build.sh stamp -DTVZ_IX_STAMP
   TVZ_ALLOW_DIAGNOSTIC=1 TVZ_LIB=$PWD/variants/libtvz_stamp.so python3 profiles/wq_stamps.py [min_match]
Read the SHARES, not the length (the stamps serialise what the real kernel overlaps); with the launch's wall
time the sum of the waves' lifetimes gives the average number of waves resident."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, sharded, synth
MM = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, 4096, seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
d_q, d_off, ml = tc.pack_queries(queries, dev)
ws = torch.empty(tc.workspace_bytes(4096, ml, 16384, 16), dtype=torch.uint8, device=dev)
lib = _lib.load(); out = (C.c_ulonglong * 16)()
st = torch.cuda.Stream(dev)
def run():
    dc.match_topk(d_q, d_off, ml, MM, 16384, 16, workspace=ws, algo=_lib.ALGO_WAVE, stream=st)
for _ in range(3): run()
torch.cuda.synchronize()
lib.tvz_debug_ix_stamps(out)
R, ts = 5, []
for _ in range(R):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); run(); b.record(st); st.synchronize(); ts.append(a.elapsed_time(b))
lib.tvz_debug_ix_stamps(out)
v = np.array(list(out), dtype=np.float64); waves = v[15]
names = ["lds set-up", "probe + layout", "posting loads issued", "pass A (touch)", "rank", "slot rows + ivid issue", "pass B", "emit + keep",
         "final top-k + store"]
tot = v[:9].sum()
ms = float(np.median(ts))
print(json.dumps({"workload": f"shard8 wave kernel, min_match {MM}", "waves": int(waves), "call_ms_with_stamps": round(ms, 4),
                  "cycles_per_wave": round(tot / waves), "postings_per_query": round(v[14] / waves),
                  "avg_waves_resident_if_stamps_tick_at_2.4GHz": round(tot / R / (ms * 1e-3 * 2.4e9), 1),
                  "per_phase_cycles_per_wave": {k: round(x / waves) for k, x in zip(names, v[:9])},
                  "share": {k: round(x / tot, 3) for k, x in zip(names, v[:9])}}))
