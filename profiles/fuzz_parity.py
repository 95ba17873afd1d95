#!/usr/bin/env python3
"""Randomised differential soak: HIP path vs the oracle for N seconds (default 60).
   python profiles/fuzz_parity.py [seconds] [seed]
The cases live in tests/fuzz_cases.py (a seeded slice of them runs inside `pytest -m gpu`:
tests/test_fuzz_gpu.py).  Prints a summary line; exits non-zero at the first mismatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz_cases  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
try:
    stats = fuzz_cases.run(SECONDS, SEED, progress=True)
except AssertionError as e:
    print(e.args[0] if e.args else e)
    sys.exit(1)
print(json.dumps({"seconds": SECONDS, "seed": SEED, **stats, "result": "no mismatch"}))
