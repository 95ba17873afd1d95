#!/usr/bin/env python3
"""Throughput of the generic (pitched / padded rows) scene path vs the flat path."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import scene
dev = torch.device("cuda:0")
T, H, W = 2048, 1080, 1920
for name, pitch, hpad in (("flat", W, 0), ("pitch2048", 2048, 0), ("pitch1984_hpad8", 1984, 8)):
    big = torch.randint(0, 256, (T, H + hpad, pitch), dtype=torch.uint8, device=dev)
    view = big[:, :H, :W]
    sc = scene.SceneScorer(H, W, T, dev)
    ts = []
    for r in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sc.score_batch(view, carry=False); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    med = float(np.median(ts[2:]))
    print(json.dumps({"layout": name, "ms": round(med, 3), "GBps_algorithmic": round((T - 1) * H * W / med / 1e6, 1),
                      "fps": round(T / med * 1e3)}))
    del big, view, sc
