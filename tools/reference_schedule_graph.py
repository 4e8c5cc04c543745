#!/usr/bin/env python3
"""The headline's protocol on the graph the REFERENCE's own insert schedule builds.

bench.py measures the search on a graph built by this repository's batched rounds (its own schedule: the reference's
build is a race between NumCPU-1 workers, insert.go:70-111, with no single answer).  What IS pinned to the reference is
the sequential schedule -- one insertSinglePoint after another, insert.go:16-68 -- which the device runs with
round_size = 1 and which tests/test_gpu_build.py holds to the oracle edge for edge.  This tool asks what the headline
is worth on THAT graph:

  1. a prefix of the rows is inserted sequentially on the device AND by the oracle (CPU): equal graphs, edge for edge,
     at a size the parity tests do not reach (--prefix rows of the bench's own data, R = 64, searchSize 75);
  2. the device carries on sequentially up to --rows (or until --budget-s is spent: the row count reached is printed
     and everything below uses it);
  3. the same rows are built with the batched rounds;
  4. both graphs answer the same query batches: recall@10 against the exact scan, queries/s, the kernel's time by HIP
     events, n_dist / hops per query, algorithmic bytes per launch and the fraction of the HBM peak.

One JSON object on stdout (and --out).  The oracle is used as the checker of step 1 only."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (data generator, exact top-k, the HBM peak)


def log(*a):
    print("[refsched]", *a, file=sys.stderr, flush=True)


def new_index(a, name, capacity, dev_index):
    from semadb_amd import vamana
    params = vamana.IndexVectorVamanaParameters(a.dim, "cosine", a.search_size, a.degree_bound, 1.2)
    ix = vamana.NewIndexVamana(name, params, device=dev_index, capacity=capacity + 1)
    ix.set_start(bench.start_vector(a.dim))
    return ix


def measure(a, ix, base, queries):
    nq, k, L, d = a.batch, 10, a.search_size, a.dim
    n_nodes, n_edges, _ = ix.stats()
    hits = 0
    for b in range(2):
        ids, _, _, _ = ix.search_batch(queries[b], k, L)
        truth = bench.exact_topk(queries[b], base, k)[1] + 2
        hits += int((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).sum().item())
    for b in range(2, 4):  # warm-up on batches that are not timed
        ix.search_batch(queries[b], k, L)
    torch.cuda.synchronize()
    timed = list(range(4, queries.shape[0]))
    ix.set_profiling(True)
    ix.profile_read()
    t0 = time.perf_counter()
    for b in timed:
        ix.search_batch(queries[b], k, L)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = [float(v) for v in ix.profile_read()][-len(timed):]
    ix.set_profiling(False)
    alg = nd = nh = 0.0
    for b in timed:
        _, _, _, tr = ix.search_batch(queries[b], k, L, trace=True)
        n_dist = int(tr.n_dist.to(torch.int64).sum().item())
        n_edge = int(tr.n_edges.to(torch.int64).sum().item())
        alg += n_dist * d * 4 + n_edge * 4
        nd += n_dist / nq
        nh += float(tr.n_hop.float().mean().item())
    ach = alg / (sum(kms) * 1e-3) / 1e9
    return {"nodes": int(n_nodes), "mean_degree": round(n_edges / n_nodes, 2),
            "recall_at_10": round(hits / (2 * nq * k), 4), "qps": round(len(timed) * nq / dt, 1),
            "kernel_ms_avg": round(float(np.mean(kms)), 4), "mean_n_dist": round(nd / len(timed), 1),
            "mean_n_hop": round(nh / len(timed), 1), "algorithmic_bytes_per_launch": int(alg / len(timed)),
            "hbm_GB/s": round(ach, 1), "frac_of_hbm_peak": round(ach / bench.HBM_PEAK_GBS, 4),
            "timed_batches": len(timed)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--prefix", type=int, default=20000, help="rows also inserted by the oracle and compared edge for edge")
    ap.add_argument("--budget-s", type=float, default=900.0, help="wall time for the sequential device build")
    ap.add_argument("--chunk", type=int, default=25000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--search-size", type=int, default=75)
    ap.add_argument("--degree-bound", type=int, default=64)
    ap.add_argument("--dist", default="latent:24")
    ap.add_argument("--timed-batches", type=int, default=12)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from semadb_amd import _lib
    _lib.lib()  # fails loudly without the HIP library
    dev = "cuda:0"
    base = bench.gen_rows(a.rows, a.dim, 20250620, a.dist, dev)
    queries = bench.gen_rows((4 + a.timed_batches) * a.batch, a.dim, 20250621, a.dist, dev).view(4 + a.timed_batches, a.batch, a.dim)
    out = {"workload": "%d x %d cosine, %s, searchSize %d, degreeBound %d, batch %d, k 10" %
                       (a.rows, a.dim, a.dist, a.search_size, a.degree_bound, a.batch)}

    # ---- 1 + 2: the reference's schedule on the device, its prefix against the oracle
    seq = new_index(a, "refsched", a.rows, 0)
    prefix = min(a.prefix, a.rows)
    t0 = time.time()
    if prefix:
        seq.insert_batch(np.arange(2, 2 + prefix, dtype=np.uint64), base[:prefix], round_size=1)
    torch.cuda.synchronize()
    t_prefix = time.time() - t0
    if prefix:
        from oracle import oracle
        from tests.helpers import assert_same_graph
        o = oracle.Index(a.dim, "cosine", a.degree_bound, a.search_size, 1.2,
                         impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
        o.set_start(np.asarray(bench.start_vector(a.dim), dtype=np.float32))
        rows = base[:prefix].cpu().numpy()
        t0 = time.time()
        for i in range(prefix):
            rc = o.insert(2 + i, rows[i])
            assert rc == 0, rc
        t_oracle = time.time() - t0
        assert_same_graph(seq, o)  # raises on any difference
        out["prefix"] = {"rows": prefix, "equal_to_oracle_edge_for_edge": True, "device_s": round(t_prefix, 2),
                         "oracle_s": round(t_oracle, 2),
                         "reference": "insertSinglePoint one after another, shard/index/vamana/insert.go:16-68"}
        log("prefix of %d rows: device %.1fs, oracle %.1fs, graphs equal" % (prefix, t_prefix, t_oracle))
        del o
    done, t_seq = prefix, t_prefix
    while done < a.rows and t_seq < a.budget_s:
        hi = min(a.rows, done + a.chunk)
        t0 = time.time()
        seq.insert_batch(np.arange(2 + done, 2 + hi, dtype=np.uint64), base[done:hi], round_size=1)
        torch.cuda.synchronize()
        dt = time.time() - t0
        t_seq += dt
        log("sequential build: %d rows, %.1fs (%.0f inserts/s in the last chunk)" % (hi, t_seq, (hi - done) / dt))
        done = hi
    n = done
    out["rows_reached"] = n
    out["sequential_build_s"] = round(t_seq, 1)
    if n < a.rows:
        out["note"] = "the sequential build stopped at %d rows (budget %.0f s): both graphs below hold these rows" % (n, a.budget_s)
    sub = base[:n]
    out["reference_schedule_graph"] = measure(a, seq, sub, queries)
    log("reference schedule:", json.dumps(out["reference_schedule_graph"]))
    seq.close()

    # ---- 3 + 4: the batched rounds on the same rows
    bat = new_index(a, "batched", n, 0)
    torch.cuda.synchronize()
    t0 = time.time()
    bat.insert_batch(None, sub)
    torch.cuda.synchronize()
    out["batched_build_s"] = round(time.time() - t0, 2)
    out["batched_graph"] = measure(a, bat, sub, queries)
    log("batched rounds:", json.dumps(out["batched_graph"]))
    bat.close()
    r, b = out["reference_schedule_graph"], out["batched_graph"]
    out["batched_over_reference_schedule"] = {"qps": round(b["qps"] / r["qps"], 4),
                                              "recall_at_10": round(b["recall_at_10"] - r["recall_at_10"], 4),
                                              "mean_n_dist": round(b["mean_n_dist"] / r["mean_n_dist"], 4)}
    text = json.dumps(out)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
