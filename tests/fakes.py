"""Test-only stand-ins (never imported by the product)."""
import threading

import numpy as np
import torch

from oracle import oracle

NEVER = 0x7FFFFFFF


class OracleCorpus:
    """Mimics tvidz_amd.corpus.DeviceCorpus on the CPU so the SQL/route layer can be tested
    without a GPU.  Backed by the oracle: test infrastructure, not a product fallback."""

    def __init__(self):
        self.rows = []
        self.lock = threading.Lock()

    def upload(self, rows):
        with self.lock:
            self.rows = [(int(v), list(t)) for v, t in rows]

    def upsert(self, video_id, timestamps):
        with self.lock:
            for i, (v, _) in enumerate(self.rows):
                if v == video_id:
                    self.rows[i] = (v, list(timestamps))
                    return
            self.rows.append((int(video_id), list(timestamps)))

    def clear(self):
        with self.lock:
            self.rows = []

    def stats(self):
        return len(self.rows), sum(len(t) for _, t in self.rows), 0

    def find_duplicates(self, new_timestamps, min_match=5, exclude_id=-1, with_kth=False):
        with self.lock:
            rows = list(self.rows)
        if not rows:
            return []
        ids, cnt, kth = oracle.match_kth(rows, list(new_timestamps), min_match)
        out = sorted((int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(rows))
                     if cnt[c] >= min_match and ids[c] != exclude_id)
        return out if with_kth else [(a, b) for a, b, _ in out]

    def align(self, timestamps, eps=0.1, max_offset=60.0):
        with self.lock:
            rows = list(self.rows)
        import numpy as np
        return np.array(oracle.align_py(rows, list(timestamps), eps, max_offset), dtype=np.int32).reshape(-1, 5)

    def close(self):
        pass


class OracleBackend:
    """Test stand-in for sharded.HipBackend (same method shapes, CPU tensors)."""

    def __init__(self, ids=None, offs=None, keys=None, live=None):
        """Static CSR shard, or `live` = an OracleCorpus whose rows are read at every match."""
        self._static = (ids, offs, keys)
        self.live = live

    @property
    def ids(self):
        return self._csr()[0]

    @property
    def offs(self):
        return self._csr()[1]

    @property
    def keys(self):
        return self._csr()[2]

    def _csr(self):
        if self.live is None:
            return self._static
        with self.live.lock:
            rows = list(self.live.rows)
        ids = np.array([v for v, _ in rows], dtype=np.int32)
        lens = np.array([len(t) for _, t in rows], dtype=np.int64)
        offs = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        keys = np.array([x for _, t in rows for x in t], dtype=np.float64)
        return ids, offs, keys

    def match(self, d_q, d_off, max_len, min_match, cap, d_excl):
        q_all, off = d_q.numpy(), d_off.numpy()
        Q = len(off) - 1
        ids, offs, keys = self._csr()
        hits = torch.zeros((Q, cap, 3), dtype=torch.int32)
        n = torch.zeros(Q, dtype=torch.int32)
        for qi in range(Q):
            q = q_all[off[qi]:off[qi + 1]]
            cnt, kth = oracle.match_kth_csr(q, offs, keys, min_match) if len(ids) else ([], [])
            rows = [(int(ids[c]), int(cnt[c]), int(kth[c])) for c in range(len(ids))
                    if cnt[c] >= min_match and (d_excl is None or ids[c] != int(d_excl[qi]))]
            rows = rows[::-1]  # unspecified order, like the atomic appends of the HIP kernel
            n[qi] = len(rows)
            for j, r in enumerate(rows[:cap]):
                hits[qi, j] = torch.tensor(r, dtype=torch.int32)
        return hits, n

    @staticmethod
    def _best(ent, k):
        ent = sorted(ent, key=lambda h: (h[2], h[0], h[1]))[:k]
        return ent + [(-1, 0, NEVER)] * (k - len(ent))

    def topk_shard(self, hits, hits_n, k):
        Q, cap, _ = hits.shape
        out = torch.empty((Q, k + 1, 3), dtype=torch.int32)
        for q in range(Q):
            m = min(int(hits_n[q]), cap)
            ent = [tuple(int(x) for x in e) for e in hits[q, :m]]
            n = int(hits_n[q])
            out[q] = torch.tensor(self._best(ent, k) + [(-1, -n if n > cap else n, NEVER)], dtype=torch.int32)
        return out

    def topk_merge(self, gathered, k):
        R, Q, k1, _ = gathered.shape
        out = torch.empty((Q, k, 3), dtype=torch.int32)
        totals = torch.zeros(Q, dtype=torch.int32)
        for q in range(Q):
            ent = []
            for r in range(R):
                ent += [tuple(int(x) for x in e) for e in gathered[r, q, :k] if int(e[0]) >= 0]
                totals[q] += abs(int(gathered[r, q, k, 1]))
            if any(int(gathered[r, q, k, 1]) < 0 for r in range(R)):
                totals[q] = -totals[q]
            out[q] = torch.tensor(self._best(ent, k), dtype=torch.int32)
        return out, totals




# ---- the N-rank service on a CPU box (tests/test_service_launch_cpu.py) --------------------------
class CutReader:
    """What the CPU driver's frame source returns: the cut timestamps of an upload, delivered in
    micro-batches (the GPU driver gets them from the scene kernels; that half has its own -m gpu tests)."""

    def __init__(self, cuts, per_batch=3, frames=300):
        self.cuts, self.per_batch, self.total_frames = list(cuts), int(per_batch), int(frames)
        self.H, self.W, self.time_base, self.bitdepth = 8, 8, (1, 30), 8
        self.closed = False

    def close(self):
        self.closed = True


def cut_inspector(store, **kw):
    """inspector.Inspector with the scene half replaced: the reference's per-cut body (app.py:231-255,
    Inspector._after_cuts) runs unchanged over the reader's cut list."""
    from tvidz_amd import inspector as insp

    class CutListInspector(insp.Inspector):
        def _run(self, analysis_key, video_id, reader):
            scene_timestamps, dups = [], []
            try:
                for b in range(0, len(reader.cuts), reader.per_batch):
                    grew = False
                    for ts in reader.cuts[b:b + reader.per_batch]:
                        if not scene_timestamps or ts != scene_timestamps[-1]:        # app.py:231
                            scene_timestamps.append(ts)
                            grew = True
                    frames_done = min(reader.total_frames, (b + reader.per_batch) * 10)
                    if grew:
                        scene_timestamps, stop = self._after_cuts(analysis_key, video_id, scene_timestamps,
                                                                  frames_done, reader.total_frames, dups)
                        if stop:
                            break
                    self._progress(analysis_key, scene_timestamps, frames_done, reader.total_frames, dups)
            finally:
                self.store.flush(video_id)
            return scene_timestamps, dups
    return CutListInspector(store, **kw)


def cuts_of_key(key: str):
    """The synthetic upload named by an S3 key: `.../<digits>-<name>__<c0>_<c1>_...` carries its cut list
    (tenths of seconds) in the name, so every rank process derives the same clip from the key alone."""
    name = key.split("/")[-1]
    body = name.rsplit(".", 1)[0].split("__", 1)[1]
    return [int(x) / 10.0 for x in body.split("_")]


def cpu_rank_parts(rank, world, group, a):
    """service.py `--parts tests.fakes:cpu_rank_parts`: a rank made of the oracle (shard + matcher
    backend) on gloo; the tick exchange, the store, the driver body and the routes are the product's."""
    from tvidz_amd import sharded
    shard = OracleCorpus()
    matcher = sharded.ShardedMatcher(OracleBackend(live=shard), k=a.k, cap=max(a.cap, a.k), group=group)
    return dict(shard=shard, matcher=matcher, xdev="cpu",
                inspector=lambda store: cut_inspector(
                    store, device="cuda:0", max_workers=a.workers,
                    frame_source=lambda bucket, key, filename, uid: (CutReader(cuts_of_key(key)), None)))


# ---- the N-rank service on the GPU at world size 1 (tests/test_service_gpu.py) --------------------
class KeyedClipReader:
    """A procedural clip named by its S3 key `<stamp>-<name>__<pts0>__<c0>_<c1>_...`: flat scenes whose
    level changes by >= 45 at the listed frame numbers (what the scene filter selects) + 2-bit noise;
    the reader protocol of tvidz_amd.feeder (H, W, time_base, total_frames, bitdepth, read_into, pts_of)."""
    H, W, T = 96, 128, 64
    LEVELS = [40, 130, 220, 85, 175, 30, 120, 210]

    def __init__(self, key):
        body = key.split("/")[-1].rsplit(".", 1)[0].split("__")
        self.pts0 = int(body[1])
        self.cuts = [int(x) for x in body[2].split("_")]
        self.time_base, self.total_frames, self.bitdepth = (1, 30), self.T, 8
        self.t, self.closed = 0, False
        self.noise = np.random.default_rng(4).integers(0, 4, size=(self.H, self.W), dtype=np.uint8)

    def read_into(self, out):
        n = 0
        while n < out.shape[0] and self.t < self.T and not self.closed:
            level = self.LEVELS[sum(1 for c in self.cuts if c <= self.t) % len(self.LEVELS)]
            np.add(self.noise, self.t & 3, out=out[n])
            np.bitwise_and(out[n], 3, out=out[n])
            out[n] += level
            self.t += 1
            n += 1
        return n

    def pts_of(self, n):
        return self.pts0 + n

    def close(self):
        self.closed = True


def gpu_rank_parts(rank, world, group, a):
    """service.py `--parts tests.fakes:gpu_rank_parts`: the PRODUCT's rank (DeviceCorpus, RCCL matcher behind
    the C ABI, the real driver with the HIP scene kernels) - only the frame source is synthetic."""
    from tvidz_amd import service
    parts = service._hip_parts(rank, world, group, a)
    from tvidz_amd.inspector import Inspector
    parts["inspector"] = lambda store: Inspector(store, device=f"cuda:{a.device}", max_workers=a.workers, batch=32,
                                                 frame_source=lambda b, key, f, u: (KeyedClipReader(key), None))
    return parts
