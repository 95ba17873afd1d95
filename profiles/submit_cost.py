#!/usr/bin/env python3
"""Host time of the calls one sharded batch is made of (rank 0's 1/8 shard, Q queries), nothing waiting on the
device: each call is timed in a loop of `reps` with a device sync every 8 calls so the queue never fills.
   python profiles/submit_cost.py [Q] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, sharded, synth  # noqa: E402

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
d_q, d_off, ml = tc.pack_queries(queries, dev)
comm = sharded.make_comm(0)
dc = tc.DeviceCorpus(0)
dc.upload_csr(*sharded.shard_csr(ids, offs, keys, 0, 8))
sm = sharded.RcclShardedMatcher(dc, comm, k=16, cap=16384, n_streams=3)
st = torch.cuda.Stream(dev)
ws = torch.empty(tc.workspace_bytes(Q, ml, 16384, 16, 1), dtype=torch.uint8, device=dev)
out = torch.empty((Q, 17, 3), dtype=torch.int32, device=dev)
mo = (torch.empty((Q, 16, 3), dtype=torch.int32, device=dev), torch.empty(Q, dtype=torch.int32, device=dev))
ev = torch.cuda.Event()
sm_ws = torch.empty(tc.workspace_bytes(Q, ml, 16384, 16, 1), dtype=torch.uint8, device=dev)


def timed(name, fn):
    for _ in range(16):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(REPS):
        a = time.perf_counter()
        fn()
        tot += time.perf_counter() - a
        if i % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{name:58s} {tot * 1e6 / REPS:7.2f} us")


timed("workspace_bytes (ctypes, no device work)", lambda: tc.workspace_bytes(Q, ml, 16384, 16, 1))
timed("event record on a side stream (torch)", lambda: ev.record(st))
timed("tvz_match_topk (fused lookup: one launch)", lambda: dc.match_topk(d_q, d_off, ml, 2, 16384, 16, out=out, workspace=ws, stream=st))
timed("tvz_topk_merge of one block (one launch)", lambda: tc.topk_merge(out.view(1, Q, 17, 3), 16, stream=st))
timed("tvz_match_sharded (lookup + all-gather + merge)", lambda: comm.match_sharded(dc, d_q, d_off, ml, 2, 16384, 16, None, workspace=sm_ws, stream=st, out=mo))
timed("RcclShardedMatcher.submit(inputs_ready)", lambda: sm.submit(d_q, d_off, ml, 2, inputs_ready=True))
timed("RcclShardedMatcher.submit (waits for the caller's stream)", lambda: sm.submit(d_q, d_off, ml, 2))
dc.close()
comm.close()
