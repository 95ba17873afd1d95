"""tvidz_amd — MI355X-native core of the tvidz inspector hot path.

Scene-cut fingerprint extraction (scene.py) and timestamp-corpus matching (corpus.py, db.py,
sharded.py) behind the reference's own call shapes; compute lives in hand-written HIP kernels
(csrc/) reached through the C ABI of include/tvz.h.  There is no CPU fallback: without
libtvz.so every entry point raises.
"""
__version__ = "0.1.0"
