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


def _device_faults_cmd(exe):
    libdir = os.path.join(ROOT, "semadb_amd")
    return ["g++", "-std=c++17", "-O1", "-pthread", "-rdynamic", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
            os.path.join(ROOT, "tests", "host", "test_device_faults.cpp"), "-o", exe, "-L" + libdir, "-lsemadb_amd",
            "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]


@pytest.mark.gpu
def test_device_memory_faults_end_in_a_status(tmp_path):
    """tests/host/test_device_faults.cpp: hipMalloc / hipMallocAsync / hipHostMalloc fail at the k-th call made inside
    libsemadb_amd.so, k swept over load / search_batch (plain, filtered, bitset, quantized) / attach_pq / insert_batch
    (with table growth) / delete_batch / compact (plain and quantized) / cluster_search_batch: SDB_ERR_DEVICE with a
    message (or the call goes on without an optional cache), the index answers as before or says it must be reloaded,
    and device memory is back at its baseline when the handles are gone (CONTRIBUTING.md:150, manager.go:231-240)."""
    exe = os.path.join(str(tmp_path), "test_device_faults")
    subprocess.check_call(_device_faults_cmd(exe))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=1500)
    print(out.stdout[-8000:], out.stderr[-2000:])
    assert out.returncode == 0 and "all device-memory faults ended in a status" in out.stdout
    for section in ("sdb_index_load", "search_batch (first call)", "search_batch (filtered)", "sdb_index_attach_pq",
                    "search_batch (quantized)", "sdb_index_insert_batch", "sdb_index_delete_batch", "sdb_index_compact",
                    "sdb_index_compact (quantized)", "sdb_cluster_search_batch"):
        assert section in out.stdout, section
    assert "did not come back" not in out.stdout
