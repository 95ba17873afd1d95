"""Build libtvz.so (HIP, gfx950 only) in-tree with hipcc.

    python -m tvidz_amd.build [--force]

The shared object lands next to this file (tvidz_amd/libtvz.so): git-ignored, but it
travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
SO = os.path.join(HERE, "libtvz.so")
SOURCES = ["tvz_api.hip", "tvz_scene.hip", "tvz_match.hip", "tvz_comm.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtvz.so cannot be built on this machine")
    return exe


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    so_m = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "tvz.h")]
    return any(os.path.getmtime(d) > so_m for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    # TVZ_CXXFLAGS reaches the product compile only together with TVZ_DIAGNOSTIC=1, and the library it makes then
    # says so (-DTVZ_DIAGNOSTIC: tvz_version() negated, refused by the binding): a stray environment variable cannot
    # turn tvidz_amd/libtvz.so into one of the diagnostic builds of profiles/variant_build.sh
    extra = []
    if os.environ.get("TVZ_DIAGNOSTIC") == "1":
        extra = ["-DTVZ_DIAGNOSTIC=1"] + os.environ.get("TVZ_CXXFLAGS", "").split()
    elif os.environ.get("TVZ_CXXFLAGS"):
        print("tvidz_amd.build: TVZ_CXXFLAGS ignored (set TVZ_DIAGNOSTIC=1 for a diagnostic build)", file=sys.stderr)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}",
           "-o", SO + ".tmp"] + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
