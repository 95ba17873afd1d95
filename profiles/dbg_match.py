import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tvidz_amd import corpus as tc, _lib
dc = tc.DeviceCorpus(0)
dc.upload([(1, [1.0, 2.0, 3.0, 4.0, 5.0]), (2, [10.0, 20.0, 30.0, 40.0, 50.0])])
q = [10.0, 20.0, 30.0, 40.0, 50.0]
for mm in (0, 1, 2, 3, 5, 6):
    print("find_duplicates mm", mm, dc.find_duplicates(q, mm, with_kth=True))
d_q, d_off, ml = tc.pack_queries([np.array(q)], "cuda:0")
for algo in (1, 2, 3):
    for mm in (1, 2, 5):
        h, n = dc.match(d_q, d_off, ml, mm, 4, algo=algo)
        torch.cuda.synchronize()
        print("match algo", algo, "mm", mm, n.tolist(), h[0, :max(int(n[0]), 0)].tolist())
