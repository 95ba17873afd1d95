"""Seeded synthetic workloads of the shapes BASELINE.json names.

  synth_luma(...)             uint8 [T,H,W] luma with known scene cuts, generated with torch on
                              the requested device (20.7 GB for 10k x 1080p stays in HBM)
  synth_timestamp_corpus(...) CSR timestamp corpus (C videos x ~200 cuts) + query videos

Data only: nothing here is on the measured path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

FRAME_SEED = 20250815
CORPUS_SEED = 1234


def synth_luma(T: int, H: int, W: int, device="cpu", seed: int = FRAME_SEED,
               min_scene: int = 30, max_scene: int = 300, adversarial: bool = True,
               out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, List[int]]:
    """Scenes of U{min_scene..max_scene} frames; a scene is a smooth base image (gradient +
    blocks, mean level U[40,215]) panned slowly, plus U{-2..2} noise per frame.  Consecutive
    scenes differ by > 40 mean levels, so a true cut has mafd >> 30 and a non-cut mafd < 5.
    With adversarial=True a back-to-back cut pair, a slow fade and a one-frame flash are
    inserted.  Returns (frames, ground-truth cut frame indices of the *scene layout*; the
    oracle decides what the scene filter selects)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    frames = out if out is not None else torch.empty((T, H, W), dtype=torch.uint8, device=dev)
    assert frames.shape == (T, H, W) and frames.dtype == torch.uint8
    yy = torch.linspace(0, 1, H, device=dev).view(H, 1)
    margin = 64
    xx = torch.linspace(0, 1, W + margin, device=dev).view(1, W + margin)
    cuts: List[int] = []
    t = 0
    prev_level = None
    scene_no = 0
    while t < T:
        n = int(rng.integers(min_scene, max_scene + 1))
        special = None
        if adversarial and scene_no in (3, 7, 11):
            special = {3: "b2b", 7: "fade", 11: "flash"}[scene_no]
            if special == "b2b":
                n = 1  # a one-frame scene: cuts at t and t+1
        n = min(n, T - t)
        level = float(rng.uniform(40, 215))
        while prev_level is not None and abs(level - prev_level) <= 45:
            level = float(rng.uniform(40, 215))
        gx, gy = rng.uniform(-30, 30, size=2)
        base = level + gx * (xx - 0.5) + gy * (yy - 0.5)
        for _ in range(4):  # blocks
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W + margin))
            hh, ww = int(rng.integers(H // 16 + 1, H // 3 + 2)), int(rng.integers(W // 16 + 1, W // 3 + 2))
            base[y0:y0 + hh, x0:x0 + ww] += float(rng.uniform(-25, 25))
        base = base.clamp_(3, 252)
        pan_speed = float(rng.uniform(0, margin / max(n, 1)))
        if t > 0:
            cuts.append(t)
        for s in range(0, n, 64):  # micro-batches keep peak memory small
            m = min(64, n - s)
            offs = [int(pan_speed * (s + i)) % margin for i in range(m)]
            stack = torch.stack([base[:, o:o + W] for o in offs])
            if special == "fade":
                # triangle wave, exactly 10 levels per frame: mafd ~ 10, never a cut
                ramp = torch.arange(s, s + m, device=dev, dtype=torch.float32).view(m, 1, 1)
                tri = (ramp % 20 - 10).abs() * 10.0
                stack = stack + tri * (1.0 if level < 128 else -1.0)
            noise = torch.randint(-2, 3, (m, H, W), device=dev, generator=g, dtype=torch.int16)
            frames[t + s:t + s + m] = (stack + noise).clamp_(0, 255).to(torch.uint8)
        if special == "flash" and n > 10:
            frames[t + 5] = 250
        prev_level = level
        t += n
        scene_no += 1
    return frames, cuts


def _round_sig6(x: np.ndarray) -> np.ndarray:
    """Vectorised stand-in for float('%.6g' % x) on positive values (synthetic data only; the
    exact formatter is scene.format_pts_time)."""
    e = np.floor(np.log10(np.maximum(x, 1e-300)))
    scale = np.power(10.0, 5 - e)
    return np.round(x * scale) / scale


def synth_timestamp_corpus(C: int, seed: int = CORPUS_SEED, mean_len: float = 200.0,
                           dup_frac: float = 0.01, frag_frac: float = 0.01, first_id: int = 1):
    """-> (ids int32[C], offsets int64[C+1], keys float64[n]).  Per video: L ~ round(N(mean,mean/10))
    clipped to [mean/4, 2*mean]; cut frames = sorted sample without replacement of 1..dur*fps,
    dur ~ U[600,7200] s, fps in {24,25,30}; ts = 6-significant-digit (1/fps)*frame.  dup_frac of
    the videos are exact copies of another video, frag_frac are prefixes of another."""
    rng = np.random.default_rng(seed)
    lo, hi = max(2, int(mean_len / 4)), int(mean_len * 2)
    lens = np.clip(np.round(rng.normal(mean_len, mean_len / 10.0, size=C)), lo, hi).astype(np.int64)
    fps = rng.choice(np.array([24, 25, 30]), size=C)
    dur = rng.uniform(600, 7200, size=C)
    rows: List[np.ndarray] = []
    for c in range(C):
        nfr = int(dur[c] * fps[c])
        L = int(min(lens[c], nfr - 1))
        fr = np.sort(rng.choice(nfr - 1, size=L, replace=False) + 1)
        rows.append(_round_sig6((1.0 / fps[c]) * fr))
    n_dup, n_frag = int(C * dup_frac), int(C * frag_frac)
    if C > 1:
        for _ in range(n_dup):
            a, b = rng.integers(0, C, size=2)
            if a != b:
                rows[b] = rows[a].copy()
        for _ in range(n_frag):
            a, b = rng.integers(0, C, size=2)
            if a != b:
                rows[b] = rows[a][: max(2, len(rows[a]) // 2)].copy()
    lens = np.array([len(r) for r in rows], dtype=np.int64)
    offs = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    keys = np.concatenate(rows) if rows else np.zeros(0)
    ids = np.arange(first_id, first_id + C, dtype=np.int32)
    return ids, offs, keys.astype(np.float64)


def synth_queries(ids, offs, keys, Q: int, seed: int = CORPUS_SEED + 1, mean_len: float = 200.0):
    """Q query videos: half are copies of corpus videos (true duplicates), half are fresh videos
    on the same frame grids (accidental min_match=2 collisions are expected and are part of
    parity).  -> list of float64 arrays."""
    rng = np.random.default_rng(seed)
    C = len(ids)
    out = []
    fresh_ids, fresh_offs, fresh_keys = synth_timestamp_corpus(max(1, Q - Q // 2), seed=seed + 7,
                                                              mean_len=mean_len, dup_frac=0, frag_frac=0)
    fi = 0
    for q in range(Q):
        if q % 2 == 0 and C > 0:
            c = int(rng.integers(0, C))
            out.append(keys[offs[c]:offs[c + 1]].copy())
        else:
            out.append(fresh_keys[fresh_offs[fi]:fresh_offs[fi + 1]].copy())
            fi += 1
    return out
