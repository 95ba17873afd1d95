"""Child process of tests/test_comm_gpu.py: creates the libtvz RCCL communicator BEFORE its first
GPU call (a process that has touched the GPU must not be re-exec'd, so this is a fresh python),
runs the sharded match through the C ABI at world size 1 and prints the merged result as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (import only: no GPU call yet)

from tvidz_amd import corpus as tc, sharded, synth  # noqa: E402

uid = tc.Comm.unique_id()                      # ncclGetUniqueId: no GPU involved
comm = tc.Comm(uid, 1, 0, 0)                   # ncclCommInitRank: the first GPU call of this process
C, Q, k, mm = 3000, 12, 16, 2
ids, offs, keys = synth.synth_timestamp_corpus(C, seed=31, mean_len=60, dup_frac=0.03)
queries = synth.synth_queries(ids, offs, keys, Q, seed=4, mean_len=60)
dev = torch.device("cuda:0")
dc = tc.DeviceCorpus(0)
dc.upload_csr(ids, offs, keys)
d_q, d_off, max_len = tc.pack_queries(queries, dev)
excl = torch.tensor([int(ids[(5 * i) % C]) for i in range(Q)], dtype=torch.int32, device=dev)
merged, totals = comm.match_sharded(dc, d_q, d_off, max_len, mm, 64, k, d_exclude_ids=excl)
torch.cuda.synchronize()
out = {"merged": merged.cpu().tolist(), "totals": totals.cpu().tolist()}
# the pipelined form: three batches in flight over two streams, each equal to the plain call
sm = sharded.RcclShardedMatcher(dc, comm, k=k, cap=64)
tickets = [sm.submit(d_q, d_off, max_len, mm, excl) for _ in range(3)]
same = True
for t in tickets:
    m2, t2 = sm.finish(t)
    torch.cuda.synchronize()
    same = same and torch.equal(m2, merged) and torch.equal(t2, totals)
out["pipelined_equal"] = bool(same)
# the streaming form (bench.py): no wait on the caller's stream, results read after a host-side wait;
# two different batches alternate, every ticket must carry ITS batch's answer
queries_b = synth.synth_queries(ids, offs, keys, Q, seed=9, mean_len=60)
d_qb, d_offb, max_len_b = tc.pack_queries(queries_b, dev)
merged_b, totals_b = comm.match_sharded(dc, d_qb, d_offb, max_len_b, mm, 64, k, d_exclude_ids=excl)
torch.cuda.synchronize()
batches = [(d_q, d_off, max_len, merged, totals), (d_qb, d_offb, max_len_b, merged_b, totals_b)]
stream_ok = not torch.equal(merged, merged_b)
ticket, want = None, None
for i in range(7):
    b = batches[i % 2]
    nxt = sm.submit(b[0], b[1], b[2], mm, excl, inputs_ready=True)
    if ticket is not None:
        m4, t4 = sm.finish(ticket, host=True)
        stream_ok = stream_ok and torch.equal(m4, want[3]) and torch.equal(t4, want[4])
    ticket, want = nxt, b
m4, t4 = sm.finish(ticket, host=True)
stream_ok = stream_ok and torch.equal(m4, want[3]) and torch.equal(t4, want[4])
out["streaming_equal"] = bool(stream_ok)
# a truncated shard list is signalled through the all-gather by a negative total
m3, t3 = comm.match_sharded(dc, d_q, d_off, max_len, 0, 50, 8)
torch.cuda.synchronize()
out["overflow_totals"] = t3.cpu().tolist()
comm.close()
dc.close()
print("RESULT " + json.dumps(out))
