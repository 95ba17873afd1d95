"""Test-only stand-ins (never imported by the product)."""
import threading

from oracle import oracle


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
