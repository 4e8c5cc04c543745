"""Host-memory serving path A/B on one GPU (round 6): blocking calls timed from C, and the micro-batcher at 2/3/4/6
batches in flight, each with the in-place (zero-copy) host path and with staging forced (SDB_TUNE_NO_ZERO_COPY).
Usage: python tools/host_path_ab.py [rows]   -> one JSON line on stdout"""
import ctypes as C
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    a = types.SimpleNamespace(metric="cosine", search_size=75, degree_bound=64, alpha=1.2)
    dev = torch.device("cuda:0")
    d, nq, k, L, nb = 384, 1024, 10, 75, 20
    base = bench.gen_rows(rows, d, 20250620, "latent:24", dev)
    queries = bench.gen_rows(nb * nq, d, 20250621, "latent:24", dev).view(nb, nq, d)
    ix, build_s = bench.build_index(a, base, 0)
    out = {"rows": rows, "build_s": round(build_s, 2)}
    # device-resident reference rates
    outs = [ix.search_batch(queries[b], k, L) for b in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(3):
        for b in range(nb):
            ix.search_batch(queries[b], k, L)
    torch.cuda.synchronize()
    out["device_qps"] = round(3 * nb * nq / (time.perf_counter() - t0))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(3):
        for b in range(nb):
            with torch.cuda.stream(s1 if b % 2 else s2):
                ix.search_batch(queries[b], k, L)
    torch.cuda.synchronize()
    out["device_two_streams_qps"] = round(3 * nb * nq / (time.perf_counter() - t0))
    hb = C.CDLL(os.path.join(ROOT, "semadb_amd", "libsemadb_hostbench.so"))
    hb.sdb_hostbench_blocking.restype = C.c_int
    hb.sdb_hostbench_blocking.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    hb.sdb_hostbench_batcher.restype = C.c_int
    hb.sdb_hostbench_batcher.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    q_np = np.ascontiguousarray(queries.cpu().numpy().reshape(nb * nq, d))
    ref_ids = outs[0][0].cpu().numpy().view(np.uint64)
    for no_zc in (0, 1):
        ix.set_tuning("no_zero_copy", no_zc)
        tag = "staged" if no_zc else "in_place"
        for pinned in (1, 0):
            qps, ms = C.c_double(0), C.c_double(0)
            first = np.zeros((nq, k), dtype=np.uint64)
            fc = np.zeros(nq, dtype=np.uint32)
            rc = hb.sdb_hostbench_blocking(ix._h, d, q_np.ctypes.data, nb, nq, k, L, 5, pinned, first.ctypes.data, fc.ctypes.data,
                                           C.byref(qps), C.byref(ms))
            out["blocking_%s_%s" % (tag, "pinned" if pinned else "pageable")] = {
                "qps": round(qps.value), "ms": round(ms.value, 4), "rc": rc, "same_ids": bool(np.array_equal(first, ref_ids))}
        for workers in (2, 3, 4, 6):
            qps, batches, served = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            p50, p99 = C.c_double(0), C.c_double(0)
            first = np.zeros((nb * nq, k), dtype=np.uint64)
            fc = np.zeros(nb * nq, dtype=np.uint32)
            rc = hb.sdb_hostbench_batcher(ix._h, d, q_np.ctypes.data, nb * nq, k, L, 64, 48, nq, 300, workers, 1.5,
                                          first.ctypes.data, fc.ctypes.data, C.byref(qps), C.byref(batches), C.byref(served),
                                          C.byref(p50), C.byref(p99))
            out["batcher_%s_w%d" % (tag, workers)] = {"qps": round(qps.value), "p50_us": p50.value, "p99_us": p99.value, "rc": rc,
                                                      "mean_batch": round(served.value / max(1, batches.value), 1),
                                                      "same_ids": bool(np.array_equal(first[:nq], ref_ids))}
        for t_, d_ in ((1, 1), (8, 1)):
            qps, batches, served = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            p50, p99 = C.c_double(0), C.c_double(0)
            rc = hb.sdb_hostbench_batcher(ix._h, d, q_np.ctypes.data, nb * nq, k, L, t_, d_, nq, 300, 4, 0.4, None, None,
                                          C.byref(qps), C.byref(batches), C.byref(served), C.byref(p50), C.byref(p99))
            out["light_%s_%d" % (tag, t_)] = {"p50_us": p50.value, "p99_us": p99.value, "qps": round(qps.value)}
    ix.set_tuning("no_zero_copy", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
