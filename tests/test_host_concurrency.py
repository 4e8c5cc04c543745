"""The product's concurrent host code under ThreadSanitizer and AddressSanitizer + UBSan, on the CPU box.

The reference's contract for the hot path is concurrency: one goroutine per request calls IndexVamana.Search under the
shard's RLock (shard/index/search.go:53-87, shard/cache/manager.go:159-181), ClusterNode.SearchPoints fans a request out
to every shard at once (cluster/actions.go:316-351), and its own suite runs under Go's race detector
(.vscode/tasks.json:7).  The drop-in's equivalents -- the lock-free SearchBatcher, IndexVamana / GpuFanout
(semadb_amd/host/semadb_host.hpp) and the ticket turnstile / ring slots / group rendezvous of the shard exchange
(semadb_amd/csrc/turnstile.h, the very code cluster.hip compiles) -- are driven here by tests/host/test_concurrency.cpp
against a CPU stand-in of the C ABI (tests/host/mock_sdb.cpp) that answers every query with a function of the query, so
each answer is checked against the request that asked for it.

clang's runtimes are used (the image's g++ 11 libtsan does not know pthread_cond_clockwait, which libstdc++'s
condition_variable::wait_for calls, and reports every timed wait as a double lock)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")

SANITIZERS = {
    "thread": (["-fsanitize=thread"], "WARNING: ThreadSanitizer"),
    "address": (["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "ERROR: AddressSanitizer"),
}


def _build(kind, tmp_path):
    if CLANG is None:
        pytest.skip("no clang++ with sanitizer runtimes in this image")
    exe = str(tmp_path / ("conc_" + kind))
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-pthread"] + SANITIZERS[kind][0] + [
        os.path.join(HOST, "test_concurrency.cpp"), os.path.join(HOST, "mock_sdb.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    return exe


def _env():
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1"
    env["ASAN_OPTIONS"] = "detect_leaks=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    return env


@pytest.mark.parametrize("kind", ["thread", "address"])
def test_host_concurrency_under_sanitizer(kind, tmp_path):
    exe = _build(kind, tmp_path)
    marker = SANITIZERS[kind][1]
    # the sanitizer of this build is awake: a seeded race / heap overrun must be reported
    seeded = subprocess.run([exe, "seeded_bugs"], capture_output=True, text=True, timeout=120, env=_env())
    assert marker in seeded.stderr, "the %s sanitizer did not report a seeded bug:\n%s" % (kind, seeded.stderr[-2000:])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=_env())
    tail = out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert out.returncode == 0, tail
    assert "0 failures" in out.stdout, tail
    for word in ("ThreadSanitizer", "AddressSanitizer", "LeakSanitizer", "runtime error"):
        assert word not in out.stderr, "sanitizer report:\n" + tail
    for name in ("batcher_mixed", "batcher_backpressure", "batcher_window_race", "batcher_pageable",
                 "batcher_poll_and_reissue", "batcher_destructor", "turnstile_order", "turnstile_missing_ticket",
                 "view_mutex_writer_gets_its_turn",
                 "fanout", "exchange_order", "exchange_missing_rank", "vamana_readers_and_writers"):
        assert "ok   " + name in out.stdout, tail


def test_index_uses_the_tested_view_lock():
    src = open(os.path.join(ROOT, "semadb_amd", "csrc", "index.h")).read()
    assert '#include "view_mutex.h"' in src and "class ViewMutex" not in src
    assert "mutable sdb::ViewMutex view_mu;" in src


def test_cluster_source_uses_the_tested_turnstile():
    """cluster.hip must run the order code the stress drives, not a copy of it"""
    src = open(os.path.join(ROOT, "semadb_amd", "csrc", "cluster.hip")).read()
    assert '#include "turnstile.h"' in src
    assert "struct sdb_cluster : sdb::OrderState" in src
    for gone in ("struct Turn {", "static sdb_cluster::Slot *take_slot("):
        assert gone not in src, "cluster.hip grew its own copy of " + gone
    for used in ("Turn turn{c, ticket}", "take_slot(c, c->ring, lk)", "c->group->arrive(", "c->group->withdraw(",
                 "skip_unentered(c, ticket)"):
        assert used in src, used
