#!/usr/bin/env python3
"""The sharded match pipeline of ONE rank's 1/N shard under a kernel trace: which kernels a batch is
made of and how the two alternating streams overlap.   python profiles/shard_trace.py [N] [Q] [steps] [streams] [nowait|-] [depth] [algo flags]
(run under rocprofv3 --kernel-trace; profiles/shard_timeline.py reads the trace).  depth = batches in flight
(default 2: submit the next, then finish the previous)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import corpus as tc, sharded, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
N_STREAMS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
NO_WAIT = len(sys.argv) > 5 and sys.argv[5] == "nowait"      # probe only: drop the wait on the caller's stream
DEPTH = int(sys.argv[6]) if len(sys.argv) > 6 else 2
if NO_WAIT:
    torch.cuda.Stream.wait_stream = lambda self, other: None
dev = torch.device("cuda:0")
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1)
d_q, d_off, max_len = tc.pack_queries(queries, dev)
comm = sharded.make_comm(0)
s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, keys, 0, N)
dc = tc.DeviceCorpus(0)
dc.upload_csr(s_ids, s_offs, s_keys)
ALGO = int(sys.argv[7], 0) if len(sys.argv) > 7 else 0     # 0x100: two queries per block at any batch size, 0x200: never
sm = sharded.RcclShardedMatcher(dc, comm, k=16, cap=16384, n_streams=N_STREAMS, algo=ALGO)
for _ in range(3):
    sm.match_topk(d_q, d_off, max_len, 2)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
from collections import deque
inflight = deque()
t_submit = t_finish = 0.0
for _ in range(STEPS):
    a = time.perf_counter()
    inflight.append(sm.submit(d_q, d_off, max_len, 2, inputs_ready=True))
    b = time.perf_counter()
    t_submit += b - a
    if len(inflight) >= DEPTH:
        sm.finish(inflight.popleft(), host=True)
        t_finish += time.perf_counter() - b
while inflight:
    sm.finish(inflight.popleft(), host=True)
torch.cuda.synchronize()
print(f"N={N} Q={Q} streams={N_STREAMS} depth={DEPTH} nowait={NO_WAIT}: {(time.perf_counter() - t0) * 1e6 / STEPS:.1f} us per batch "
      f"(host: {t_submit * 1e6 / STEPS:.1f} us in submit, {t_finish * 1e6 / STEPS:.1f} us waiting in finish)")
dc.close()
comm.close()
