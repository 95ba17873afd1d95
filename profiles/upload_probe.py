import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tvidz_amd import corpus as tc, synth
t=time.perf_counter(); ids, offs, keys = synth.synth_timestamp_corpus(100000); print("synth", time.perf_counter()-t)
dc = tc.DeviceCorpus(0)
for _ in range(2):
    t=time.perf_counter(); dc.upload_csr(ids, offs, keys); print("upload 100k rows / %d keys: %.3f s" % (len(keys), time.perf_counter()-t))
t=time.perf_counter()
for i in range(200): dc.upsert(10**6+i, keys[:150+i])
print("200 upserts: %.3f ms each" % ((time.perf_counter()-t)*1e3/200))
