#!/usr/bin/env python3
"""Per-phase cycles of ts_match_index_kernel (wave 0 of every block, s_memtime), diagnostic build:
   bash profiles/variant_build.sh stamp -DTVZ_IX_STAMP
   TVZ_LIB=$PWD/variants/libtvz_stamp.so python3 profiles/ix_stamps.py [index|shard8] [topk]
`topk`: the lookup that keeps the per-shard top-k itself (tvz_match_topk) instead of tvz_match."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, corpus as tc, sharded, synth
which = sys.argv[1] if len(sys.argv) > 1 else "index"
fused = len(sys.argv) > 2 and sys.argv[2] == "topk"
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, 4096, seed=synth.CORPUS_SEED + 1)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*(sharded.shard_csr(ids, offs, keys, 0, 8) if which == "shard8" else (ids, offs, keys)))
d_q, d_off, ml = tc.pack_queries(queries, dev)
hits = torch.empty((4096, 16384, 3), dtype=torch.int32, device=dev); n = torch.empty(4096, dtype=torch.int32, device=dev)
ws = torch.empty(tc.workspace_bytes(4096, ml, 16384, 16), dtype=torch.uint8, device=dev)
lib = _lib.load(); out = (C.c_ulonglong * 16)()
def run():
    if fused: dc.match_topk(d_q, d_off, ml, 2, 16384, 16, workspace=ws)
    else: dc.match(d_q, d_off, ml, 2, 16384, out_hits=hits, out_n=n, workspace=ws)
for _ in range(3): run()
torch.cuda.synchronize()
lib.tvz_debug_ix_stamps(out)
R = 5
for _ in range(R): run()
torch.cuda.synchronize()
lib.tvz_debug_ix_stamps(out)
v = np.array(list(out), dtype=np.float64); blocks = v[15]
names = ["probe", "compact", "passA", "passA_barrier", "rank+elist", "elist_barrier", "ivid+passB", "passB_barrier",
         "emit_scan+barrier", "write+resets (fused: top-k keep)", "final_topk"]
tot = v[:11].sum()
print(json.dumps({"workload": which + ("+topk" if fused else ""), "blocks": int(blocks), "cycles_per_block": round(tot / blocks),
                  "per_phase_cycles_per_block": {k: round(x / blocks) for k, x in zip(names, v[:11])},
                  "share": {k: round(x / tot, 3) for k, x in zip(names, v[:11])}}))
