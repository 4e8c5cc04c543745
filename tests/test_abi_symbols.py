"""CPU-only: the C-ABI library loads and exports every symbol include/semadb_amd.h declares, the
Python signature table covers the same set, and calls fail loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "semadb_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_functions():
    names = declared_functions()
    assert "sdb_index_search_batch" in names and "sdb_distance_batch" in names and len(names) >= 20


def test_library_exports_every_declared_symbol():
    from semadb_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = C.CDLL(_lib.SO_PATH)
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, "not exported: %s" % missing


def test_header_compiles_as_c99(tmp_path):
    """cgo hands the header to a C compiler, not a C++ one"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "semadb_amd.h"\nint main(void) { sdb_index_params p; (void)p; return SDB_ABI_VERSION - 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-fsyntax-only", str(src)])


def test_python_signature_table_matches_header():
    from semadb_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_abi_version_and_error_channel():
    from semadb_amd import _lib
    L = _lib.lib()
    assert L.sdb_abi_version() == 2
    out = C.c_uint32(0)
    assert L.sdb_shard_limit(10, 8, 75, C.byref(out)) == 0 and out.value == 10  # actions.go:291-299
    assert L.sdb_shard_limit(10, 0, 75, C.byref(out)) != 0
    assert L.sdb_last_error()


def test_no_cpu_fallback_without_gpu():
    """On a machine without an MI355X every compute entry point must fail, never compute on the CPU."""
    import numpy as np
    from semadb_amd import _lib, distance, vamana
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.SemaDBError):
        distance.distance_batch("euclidean", np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32))
    with pytest.raises(_lib.SemaDBError):
        vamana.NewIndexVamana("x", vamana.IndexVectorVamanaParameters(4, "euclidean"))


def test_product_path_never_touches_the_oracle():
    """semadb_amd/ must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "semadb_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "sdb_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_every_entry_point_ends_exceptions():
    """CONTRIBUTING.md:150 (no panics) at the C boundary: every extern "C" function of the library is a
    function-try-block that ends in SDB_API_CATCH, so that no C++ exception (std::bad_alloc from a multi-GB host
    staging vector, std::system_error from a thread) unwinds through cgo into std::terminate.  The run-time proof is
    tests/host/test_faults.cpp (operator-new fault injection, on the GPU)."""
    import glob
    csrc = os.path.join(ROOT, "semadb_amd", "csrc")
    text = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc"))))
    trivial = {"sdb_last_error", "sdb_abi_version"}  # a pointer to a thread-local buffer / a constant
    unguarded = []
    for name in declared_functions():
        if name in trivial:
            continue
        m = re.search(r'^(?:extern "C" )?int\s+%s\(' % name + r"[^;{]*?\)\s*(try)?\s*\{", text, flags=re.M | re.S)
        assert m, "no definition of %s found" % name
        if not m.group(1) or 'SDB_API_CATCH("%s")' % name not in text:
            unguarded.append(name)
    assert not unguarded, "entry points that let exceptions out: %s" % unguarded


def test_fault_injection_program_compiles(tmp_path):
    import subprocess
    libdir = os.path.join(ROOT, "semadb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "host", "test_faults.cpp"), "-o",
                           str(tmp_path / "test_faults"), "-L" + libdir, "-lsemadb_amd", "-ldl", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])


def test_design_register_table_is_generated_and_design_stays_short():
    """DESIGN.md's per-kernel register figures are the tool's output for the BUILT library, not typed
    (tools/kernel_table.py --design between the kernel_table markers), and the file stays under 60 KB: earlier rounds'
    measurements live in HISTORY.md."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), "--design"], capture_output=True,
                         text=True, check=True).stdout.strip()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    begin, end = "<!-- kernel_table:begin", "<!-- kernel_table:end -->"
    i, j = design.index(begin), design.index(end) + len(end)
    assert design[i:j].strip() == out, "DESIGN.md's register table is stale: python3 tools/kernel_table.py --design"
    assert len(design.encode()) <= 60 * 1024, "DESIGN.md has %d bytes" % len(design.encode())


def test_bench_traffic_constants_are_the_committed_profiles():
    """bench.py labels roofline.traffic and config.c4.*.traffic_over_algorithmic as profile-derived: the constants it
    reads (profiles/pmc_traffic.json, profiles/pmc_c4.json) must be the figures of the profile files they name."""
    import json
    import re
    prof = os.path.join(ROOT, "profiles")
    t = json.load(open(os.path.join(prof, "pmc_traffic.json")))
    src = re.match(r"profiles/(\S+\.md)", t["source"]).group(1)
    md = open(os.path.join(prof, src)).read()
    read_b = float(re.search(r"corrected HBM read bytes / launch[^|]*\| ([0-9.e+]+) \|", md).group(1))
    write_b = float(re.search(r"WRITE_SIZE / launch \(KB\) -> bytes \| [0-9.]+ -> ([0-9.e+]+) \|", md).group(1))
    assert abs(t["read_bytes"] - read_b) <= 1e-3 * read_b and abs(t["write_bytes"] - write_b) <= 1e-2 * write_b
    assert abs(t["hbm_bytes_per_launch"] - (read_b + write_b)) <= 1e-3 * read_b
    c4 = json.load(open(os.path.join(prof, "pmc_c4.json")))
    for M in (8, 192):
        name = re.search(r"profiles/(\S*pmc_c4_M%d\.json)" % M, c4["source"]) or re.search(r"(r\d+\w*_pmc_c4_M%d\.json)" % M, c4["source"])
        rec = json.load(open(os.path.join(prof, os.path.basename(name.group(1)))))
        runs = [v for k, v in rec.items() if k.startswith("k_greedy_search")][0]
        mean = sum(r["traffic_over_codes_and_edges"] for r in runs) / len(runs)
        assert abs(c4["M=%d" % M]["traffic_over_codes_and_edges"] - mean) < 0.005, (M, mean)


def test_device_fault_injection_program_compiles(tmp_path):
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_host_mirror import _device_faults_cmd
    subprocess.check_call(_device_faults_cmd(str(tmp_path / "test_device_faults")))


def test_hot_kernels_have_no_scratch():
    """tools/kernel_table.py reads the register / scratch figures from the built library's gfx950 code objects: NO
    kernel of the library may spill a VGPR or carry scratch memory (round 6: all 948), the kernels a BASELINE
    configuration launches in a timed region spill at most 64 SGPRs into VGPR lanes (filtered walks 128), any kernel
    at most 160."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " 0 with scratch" in out.stdout
    import re
    m = re.search(r"most SGPRs spilled: (\d+) on a timed path, (\d+) anywhere", out.stdout)
    assert m and int(m.group(1)) <= 64 and int(m.group(2)) <= 160, out.stdout[-500:]


def test_bench_repeats_the_reference_schedule_record_as_it_is():
    """config.reference_schedule_graph is labelled profile-derived: what bench.py copies must be the committed file's
    figures, and the file must say that its prefix equalled the oracle's graph."""
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_refsched_1m.json")))
    assert rec["prefix"]["equal_to_oracle_edge_for_edge"] is True and rec["prefix"]["rows"] >= 20000
    assert rec["rows_reached"] == 1000000 and "1000000 x 384" in rec["workload"] and "latent:24" in rec["workload"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.path.join(ROOT, "profiles", "r06_refsched_1m.json")' in src
    assert "profile-derived, not measured in this run: profiles/r06_refsched_1m.json" in src
    for key in ("qps", "recall_at_10", "kernel_ms_avg", "frac_of_hbm_peak", "mean_n_dist"):
        assert key in rec["reference_schedule_graph"] and '"%s": g["%s"]' % (key, key) in src
