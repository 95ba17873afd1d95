"""ctypes binding of libtvz.so (the C ABI declared in include/tvz.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails,
a RuntimeError is raised.  The reference surfaces every failure of this path as an
exception caught at inspector/app.py:303; so does this binding.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libtvz.so")

KTH_NEVER = 0x7FFFFFFF

# name -> (restype, argtypes); mirrors include/tvz.h one to one
SIGNATURES = {
    "tvz_version": (C.c_int, []),
    "tvz_last_error": (C.c_char_p, []),
    "tvz_scene_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "tvz_luma_sad_u8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvz_scene_select": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "tvz_scene_scores_u8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                      C.c_int64, C.c_void_p, C.c_double, C.c_int32, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "tvz_scene_scores_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                       C.c_int64, C.c_void_p, C.c_double, C.c_int32, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "tvz_corpus_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "tvz_corpus_destroy": (C.c_int, [C.c_void_p]),
    "tvz_corpus_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_int64]),
    "tvz_corpus_upsert": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "tvz_corpus_clear": (C.c_int, [C.c_void_p]),
    "tvz_corpus_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "tvz_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                            C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvz_find_duplicates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_int64)]),
    "tvz_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                           C.c_void_p, C.c_void_p]),
    "tvz_topk_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p]),
    "tvz_topk_merge": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "tvz_align": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                            C.c_void_p]),
    # not part of the stable ABI (kernel-shape A/B knob)
    "tvz_scene_set_tuning": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "tvz_match_set_tuning": (C.c_int, [C.c_int]),
}

_lock = threading.Lock()
_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7); libtvz.so NEEDs libamdhip64.so.7.  If libtvz.so were loaded first the loader
    would take /opt/rocm's copy and torch would later map a SECOND runtime (its NEEDED name is
    the unversioned file name, so no SONAME match), and launches through one of them fail with
    "no ROCm-capable device".  Loading torch's copy first makes libtvz.so bind to it."""
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libtvz.so; RuntimeError if it was not built (python -m tvidz_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(SO_PATH):
                raise RuntimeError(
                    f"{SO_PATH} is missing: the HIP extension is not built and there is no CPU "
                    "fallback. Run `python -m tvidz_amd.build` (needs hipcc).")
            _preload_torch_hip_runtime()
            lib = C.CDLL(SO_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().tvz_last_error()
        raise RuntimeError(f"libtvz error {rc}: {(msg or b'').decode(errors='replace')}")
