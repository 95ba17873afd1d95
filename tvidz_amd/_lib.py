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
# TVZ_LIB selects another build of the same ABI (profiles/variant_*.sh time -D variants this way
# instead of copying them over the product library)
SO_PATH = os.environ.get("TVZ_LIB") or os.path.join(_HERE, "libtvz.so")

KTH_NEVER = 0x7FFFFFFF
VERSION = 400

# name -> (restype, argtypes); mirrors include/tvz.h one to one
_P = C.c_void_p
SIGNATURES = {
    "tvz_version": (C.c_int, []),
    "tvz_last_error": (C.c_char_p, []),
    "tvz_scene_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "tvz_scene_state_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "tvz_scene_state_reset": (C.c_int, [_P, _P]),
    "tvz_luma_sad_u8": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                  _P, _P, C.c_size_t, _P]),
    "tvz_scene_select": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, _P,
                                   _P, _P, _P, _P]),
    "tvz_scene_scores_u8": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P,
                                      C.c_int32, C.c_double, _P, _P, _P, _P, _P, C.c_int32, _P,
                                      C.c_size_t, C.c_uint32, _P]),
    "tvz_scene_scores_u16": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P,
                                       C.c_int32, C.c_double, _P, _P, _P, _P, _P, C.c_int32, _P,
                                       C.c_size_t, C.c_uint32, _P]),
    "tvz_corpus_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "tvz_corpus_destroy": (C.c_int, [_P]),
    "tvz_corpus_reserve": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "tvz_corpus_upload": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64]),
    "tvz_corpus_upsert": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "tvz_corpus_clear": (C.c_int, [_P]),
    "tvz_corpus_build_index": (C.c_int, [_P]),
    "tvz_corpus_index_stats": (C.c_int, [_P] + [C.POINTER(C.c_int64)] * 5),
    "tvz_corpus_bucket_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "tvz_corpus_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "tvz_match_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tvz_match_workspace_bytes_long": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "tvz_match": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P,
                            C.c_size_t, C.c_int32, _P]),
    "tvz_match_topk": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32,
                                 _P, _P, C.c_size_t, C.c_int32, _P]),
    "tvz_find_duplicates": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P,
                                      C.POINTER(C.c_int64)]),
    "tvz_topk": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "tvz_topk_shard": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "tvz_topk_merge": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "tvz_match_topk_shards": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32,
                                        C.c_int32, _P, _P, _P, _P, C.c_size_t, C.c_int32, _P]),
    "tvz_comm_unique_id": (C.c_int, [_P]),
    "tvz_comm_init": (C.c_int, [C.POINTER(C.c_void_p), _P, C.c_int32, C.c_int32, C.c_int32]),
    "tvz_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tvz_comm_destroy": (C.c_int, [_P]),
    "tvz_match_sharded": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32,
                                    C.c_int32, _P, _P, _P, C.c_size_t, C.c_int32, _P]),
    "tvz_align": (C.c_int, [_P, _P, C.c_int32, C.c_double, C.c_double, _P, _P]),
    "tvz_read_records": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P,
                                   C.POINTER(C.c_int64)]),
    "tvz_read_stream": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, C.POINTER(C.c_int64)]),
}

# per-call selectors of include/tvz.h
ALGO_AUTO, ALGO_Q1, ALGO_TILE, ALGO_JOIN, ALGO_INDEX = 0, 1, 2, 3, 4
ALGO_PAIR, ALGO_NO_PAIR = 0x100, 0x200      # OR-ed into `algo` of the top-k calls: two queries per lookup block always / never
ALGO_WAVE, ALGO_NO_WAVE = 0x400, 0x800      # ... one WAVE per query on a handle of one sub-index: required / never
ALGO_PREFER_WAVE = 0x1000                   # ... wherever it fits (a stream of batches in flight: fewer instructions per batch)
SHAPE_AUTO = 0
SHAPE_NO_NT = 1 << 30
UNIQUE_ID_BYTES = 128


def shape(U: int = 0, tc: int = 0, nt: bool = True) -> int:
    """TVZ_SHAPE(U, tc) [| TVZ_SHAPE_NO_NT]: per-call kernel-shape override of the scene kernels."""
    s = (int(U) & 0xFF) | (int(tc) << 8)
    if not nt:
        s |= SHAPE_NO_NT
    return s


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
            lib.tvz_version.restype, lib.tvz_version.argtypes = C.c_int, []
            # A stale library behind a newer binding (or the reverse) shifts arguments silently: round
            # 2's only host crash (SIGSEGV inside tvz_find_duplicates, gpurun_out/r2_t2.log) was a run
            # of the ABI-v2 binding against a library built from the half-converted sources.  Refuse.
            if lib.tvz_version() == -VERSION and os.environ.get("TVZ_ALLOW_DIAGNOSTIC") != "1":
                raise RuntimeError(f"{SO_PATH} is a DIAGNOSTIC build (tvz_version() = {lib.tvz_version()}: compiled with "
                                   "a TVZ_IX_* / TVZ_DIAGNOSTIC define; some of them return wrong results on purpose). "
                                   "The product binding refuses it; profile scripts set TVZ_ALLOW_DIAGNOSTIC=1.")
            if abs(lib.tvz_version()) != VERSION:
                raise RuntimeError(f"{SO_PATH} is version {lib.tvz_version()}, this binding expects {VERSION}: "
                                   "rebuild it (python -m tvidz_amd.build --force)")
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
