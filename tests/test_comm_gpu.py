"""The collective of the sharded corpus match BEHIND THE C ABI (tvz_comm_init / tvz_match_sharded:
local sweep -> per-shard top-k -> ncclAllGather -> merge), run on the GPU at world size 1 in a
FRESH child process that creates the communicator before its first GPU call.  The result must
equal the oracle's answer over the whole corpus (the same expectation as
test_match_gpu.py::test_sharded_pipeline_on_one_gpu).  The merge logic for world sizes 2 and 3 is
covered on CPU by the gloo tests (tests/test_sharded_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

from oracle import oracle
from tvidz_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEVER = 0x7FFFFFFF


def test_match_sharded_through_rccl_world_size_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_child.py")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    C, Q, k, mm = 3000, 12, 16, 2
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=31, mean_len=60, dup_frac=0.03)
    queries = synth.synth_queries(ids, offs, keys, Q, seed=4, mean_len=60)
    excl = [int(ids[(5 * i) % C]) for i in range(Q)]
    for qi, q in enumerate(queries):
        cnt, kth = oracle.match_kth_csr(q, offs, keys, mm)
        rows = [(int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(C)
                if cnt[c] >= mm and ids[c] != excl[qi]]
        assert res["totals"][qi] == len(rows)
        exp = sorted(rows, key=lambda h: (h[2], h[0], h[1]))[:k]
        exp += [(-1, 0, NEVER)] * (k - len(exp))
        assert [tuple(r) for r in res["merged"][qi]] == exp, qi
    assert res["pipelined_equal"] is True
    assert res["streaming_equal"] is True          # submit(inputs_ready=True) + finish(host=True), alternating batches
    assert res["overflow_totals"] == [-C] * Q          # min_match 0: every row hits, cap 50 overflows
