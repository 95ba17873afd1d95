#!/usr/bin/env python3
"""A/B sweep of the luma-SAD kernel shape in ONE process (interleaved rounds, guide rule 24).
   python profiles/tune_scene.py [T] [rounds]   -> table of median/min ms and GB/s per variant."""
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvidz_amd import _lib, scene  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
H, W = 1080, 1920
dev = torch.device("cuda:0")
lib = _lib.load()
frames = torch.randint(0, 256, (T, H, W), dtype=torch.uint8, device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "tc":      # the time-chunk sweep at the widest strip: wave-count quantisation
    variants = [(8, tc, 1) for tc in (128, 192, 256, 320, 384, 448, 512, 640, 832, 1280, 2560)]
else:
    variants = [(U, tc, nt) for U, tc, nt in itertools.product((1, 2, 4, 8), (64, 128, 256, 512), (0, 1))]
sc = scene.SceneScorer(H, W, T, dev)       # workspace covers every shape
times = {v: [] for v in variants}
ref = None
for r in range(ROUNDS):
    for v in variants:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sad, _, _, _ = sc.score_batch(frames, carry=False, shape=_lib.shape(*v))   # per-call shape
        b.record()
        torch.cuda.synchronize()
        if r == 0:
            s = sad.clone()
            if ref is None:
                ref = s
            assert torch.equal(ref, s), v
        else:
            times[v].append(a.elapsed_time(b))
rows = []
for v, t in times.items():
    med, mn = float(np.median(t)), float(np.min(t))
    rows.append({"U": v[0], "tc": v[1], "nt": v[2], "median_ms": round(med, 4), "min_ms": round(mn, 4),
                 "GBps_median": round((T - 1) * H * W / med / 1e6, 1)})
rows.sort(key=lambda r: r["median_ms"])
for r in rows:
    print(json.dumps(r))
