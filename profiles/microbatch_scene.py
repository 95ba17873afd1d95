#!/usr/bin/env python3
"""Scene path on micro-batches (the streaming driver's unit) and PCIe-inclusive feed rate.
   python profiles/microbatch_scene.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, scene  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
dev = torch.device("cuda:0")
lib = _lib.load()
big = torch.randint(0, 256, (2048, H, W), dtype=torch.uint8, device=dev)
for T in (32, 64, 128, 256, 512, 1024, 2048):
    frames = big[:T]
    res = {}
    variants = [("auto", (0, 0, 1))] + [(f"U{u}tc{tc}", (u, tc, 1)) for u in (1, 2, 4, 8) for tc in (8, 16, 32, 64, 128, 256)]
    for name, tune in variants:
        sc = scene.SceneScorer(H, W, T, dev)
        ts = []
        for r in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); sc.score_batch(frames, carry=False, shape=_lib.shape(*tune)); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[name] = round(float(np.median(ts[5:])) * 1e3, 1)
    best = min(res, key=res.get)
    print(json.dumps({"T": T, "auto_us": res["auto"], "best": best, "best_us": res[best],
                      "best_GBps": round((T - 1) * H * W / res[best] / 1e3, 1),
                      "top": sorted(res.items(), key=lambda kv: kv[1])[:5]}))

# PCIe-inclusive: pinned host frames -> HBM -> score, double buffered on two streams
T = 256
host = [torch.randint(0, 256, (T, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
devb = [torch.empty((T, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
sc = scene.SceneScorer(H, W, T, dev)
copy_s = torch.cuda.Stream()
n_batches = 40
torch.cuda.synchronize()
t0 = time.perf_counter()
evs = [None, None]
for i in range(n_batches):
    s = i & 1
    with torch.cuda.stream(copy_s):
        if evs[s] is not None:
            copy_s.wait_event(evs[s])
        devb[s].copy_(host[s], non_blocking=True)
        ce = torch.cuda.Event(); ce.record(copy_s)
    torch.cuda.current_stream().wait_event(ce)
    sc.score_batch(devb[s], carry=False)
    evs[s] = torch.cuda.Event(); evs[s].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"pcie_inclusive_fps": round(n_batches * T / dt), "GBps_h2d": round(n_batches * T * H * W / dt / 1e9, 1),
                  "batch": T, "note": "pinned host luma -> H2D -> score, double buffered"}))
