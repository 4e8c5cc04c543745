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


@pytest.mark.gpu
def test_no_exception_leaves_the_c_abi(tmp_path):
    """tests/host/test_faults.cpp: operator new throws at the k-th allocation made inside libsemadb_amd.so, k swept over
    load / insert_batch / delete_batch / (filtered) search_batch / compact / cluster_search_batch: every call returns a
    status with a message, the process lives, the index answers as before (or says it must be reloaded)."""
    exe = os.path.join(str(tmp_path), "test_faults")
    libdir = os.path.join(ROOT, "semadb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "host", "test_faults.cpp"), "-o", exe,
                           "-L" + libdir, "-lsemadb_amd", "-ldl", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(out.stdout[-6000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL FAULT-INJECTION TESTS PASSED" in out.stdout
