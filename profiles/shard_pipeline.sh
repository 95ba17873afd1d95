#!/bin/bash
# The sharded pipeline from plain C against the same loop from Python (run on the GPU box from the repo root):
#   bash profiles/shard_pipeline.sh [N=8] [Q=4096]
N=${1:-8}; Q=${2:-4096}
REPO=$(pwd)
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$REPO")
from tvidz_amd import sharded, synth
ids, offs, keys = synth.synth_timestamp_corpus(100000, seed=synth.CORPUS_SEED)
queries = synth.synth_queries(ids, offs, keys, $Q, seed=synth.CORPUS_SEED + 1)
s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, keys, 0, $N)
lens = np.array([len(q) for q in queries], dtype=np.int64)
qoffs = np.zeros(len(queries) + 1, dtype=np.int64); np.cumsum(lens, out=qoffs[1:])
qkeys = np.concatenate([np.asarray(q, dtype=np.float64) for q in queries])
with open("/tmp/shard_pipeline.bin", "wb") as f:
    np.array([len(s_ids), len(s_keys), len(queries), len(qkeys), int(lens.max())], dtype=np.int64).tofile(f)
    np.asarray(s_ids, dtype=np.int32).tofile(f); np.asarray(s_offs, dtype=np.int64).tofile(f)
    np.asarray(s_keys, dtype=np.float64).tofile(f); qoffs.tofile(f); qkeys.tofile(f)
PY
gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude profiles/shard_pipeline.c -o /tmp/shard_pipeline -Ltvidz_amd -ltvz \
    -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$REPO/tvidz_amd -Wl,-rpath,/opt/rocm/lib || exit 1
for sh in 0 0x800; do for d in 1 2 3 4; do echo -n "shape $sh: "; /tmp/shard_pipeline /tmp/shard_pipeline.bin $d 300 $sh 2>/dev/null | tail -1; done; done
for d in 2 3; do python3 profiles/shard_trace.py $N $Q 300 $d - $d 2>/dev/null | tail -1; done
