"""The C ABI without Python or torch in the process: compile tests/c_abi_smoke.c with gcc against
include/tvz.h + libtvz.so + the system HIP runtime and run it on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path):
    from tvidz_amd import build
    build.build()
    exe = str(tmp_path / "c_abi_smoke")
    rocm = "/opt/rocm"
    cmd = ["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
           f"{ROOT}/tests/c_abi_smoke.c", "-o", exe, f"-L{ROOT}/tvidz_amd", "-ltvz", f"-L{rocm}/lib",
           "-lamdhip64", "-lm", f"-Wl,-rpath,{ROOT}/tvidz_amd", f"-Wl,-rpath,{rocm}/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "c_abi_smoke OK" in out.stdout
    print(out.stdout)
    # the sharded forms from plain C: two shard handles always; the one-rank RCCL path unless no librccl.so loads
    assert "tvz_match_sharded through a one-rank RCCL communicator OK" in out.stdout or "no usable librccl.so" in out.stdout


def test_rebuilds_do_not_stall_lookups(tmp_path):
    """db.py:83: a reader never waits for a writer.  100k rows, 40k upserts (two background rebuilds
    of the index, ~1 ms of GPU work each), a thread timing every tvz_find_duplicates call from C (no
    GIL in the picture).  Every result is right; lookups START AND FINISH while a rebuild is running
    (none could if readers were drained for rebuilds, as they were in round 2: 5 ms at 100k rows);
    the 99th percentile stays at the quiet level.  The maximum is split: of the
    lookups that OVERLAP a rebuild window (the only ones a rebuild could stall) it is asserted
    (<= 1 ms: a lookup that shares the GPU with the build kernels takes 100-200 us); of the others it
    is reported - a shared host stalls lookups and upserts alike now and then, rebuild or not."""
    import json
    from tvidz_amd import build
    build.build()
    exe = str(tmp_path / "rebuild_latency")
    cmd = ["gcc", "-O1", "-pthread", f"-I{ROOT}/include", f"{ROOT}/tests/rebuild_latency.c", "-o", exe,
           f"-L{ROOT}/tvidz_amd", "-ltvz", "-lm", f"-Wl,-rpath,{ROOT}/tvidz_amd", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    res = json.loads(out.stdout.strip().splitlines()[-1])
    print(res)
    assert res["wrong_results"] == 0
    assert res["rebuilds"] >= 2 and res["indexed_rows"] > 100000, res
    assert res["lookups_during_upserts"] > 1000, res
    assert res["min_lookups_inside_one_rebuild"] >= 3, res      # readers ran THROUGH every rebuild (sharing the GPU with it)
    assert res["p99_us"] < 300.0, res
    assert res["rebuild_windows"] >= 2 and res["lookups_overlapping_rebuilds"] >= 6, res
    assert res["max_us_inside_rebuild"] <= 1000.0, res          # no reader was stalled by a rebuild
