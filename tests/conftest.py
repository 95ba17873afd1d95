import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build libtvz.so (hipcc cross-compiles gfx950 without a GPU) and the CPU oracle if their
    sources are newer than the in-tree binaries.  Failure is not hidden: the tests that need the
    library then fail loudly in _lib.load()."""
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        try:
            from tvidz_amd import build
            build.build()
        except Exception as e:  # pragma: no cover
            print(f"[conftest] building libtvz.so failed: {e}", file=sys.stderr)
    try:
        from oracle import oracle
        oracle.build()
    except Exception as e:  # pragma: no cover
        print(f"[conftest] building the oracle failed: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
