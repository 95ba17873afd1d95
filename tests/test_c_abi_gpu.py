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
