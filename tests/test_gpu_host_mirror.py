"""Compiles and runs the C++ host mirror's test program (tests/host/test_host.cpp): the reference's Go tests
for the hot path replayed through semadb_amd/host/semadb_host.hpp -> C ABI -> HIP."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    exe = os.path.join(tmp, "test_host")
    libdir = os.path.join(ROOT, "semadb_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "host", "test_host.cpp"), "-o", exe,
           "-L" + libdir, "-lsemadb_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_host_mirror_compiles(tmp_path):
    """CPU: the header and its test program compile and link against the C ABI."""
    _build(str(tmp_path))


@pytest.mark.gpu
def test_host_mirror_reference_tests(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL HOST TESTS PASSED" in out.stdout
