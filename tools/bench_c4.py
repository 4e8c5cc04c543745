#!/usr/bin/env python3
"""C4 operating points: vectorVamana + product quantizer at d = 768, K = 256, one MI355X.  For each M: whole-call and
kernel QPS at batch 1024, recall@10 (no re-ranking, like the reference), with the multi-wave walk (round 3) and, for
comparison, the one-wave kernel with the table in global memory (round 2, SDB_TUNE_PQ_NARROW).
   python tools/bench_c4.py [--rows 10000000] [--pq-m 128,192,256,384] [--narrow]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vectorstore as vs

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--pq-m", default="192")
ap.add_argument("--narrow", action="store_true", help="also measure the round-2 kernel for each M")
ap.add_argument("--batches", type=int, default=8)
a0 = ap.parse_args()


class A:
    metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2


dev, d, n, nq, k, L = "cuda:0", a0.dim, a0.rows, 1024, 10, 75
base = bench.gen_rows(n, d, 20250620, "latent:24", dev)
queries = bench.gen_rows(a0.batches * nq, d, 20250621, "latent:24", dev).view(a0.batches, nq, d)
t0 = time.time()
ix, build_s = bench.build_index(A, base, 0, name="c4")
print("[c4] built %d x %d in %.1fs" % (n, d, build_s), file=sys.stderr, flush=True)
truth = torch.cat([bench.exact_topk(queries[b], base, k)[1] + 2 for b in range(a0.batches)])
ix.set_profiling(True)


def measure():
    nb = a0.batches
    for b in range(2):
        ix.search_batch(queries[b], k, L)
    torch.cuda.synchronize()
    ix.profile_read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(2):
        for b in range(nb):
            ix.search_batch(queries[b], k, L)
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / (2 * nb)
    kms = float(np.mean(ix.profile_read()))
    hits = nd = nh = 0
    for b in range(nb):
        ids, _, _, tr = ix.search_batch(queries[b], k, L, trace=True)
        hits += int((ids.to(torch.int64).unsqueeze(2) == truth[b * nq:(b + 1) * nq].unsqueeze(1)).any(2).sum().item())
        nd += int(tr.n_dist.to(torch.int64).sum().item())
        nh += int(tr.n_hop.to(torch.int64).sum().item())
    return {"call_qps": round(nq / call_ms * 1e3, 1), "kernel_qps": round(nq / kms * 1e3, 1), "kernel_ms": round(kms, 4),
            "call_ms": round(call_ms, 4), "recall_at_10": round(hits / (nb * nq * k), 4),
            "n_dist_per_query": round(nd / nb / nq, 1), "n_hop_per_query": round(nh / nb / nq, 1)}


out = {"rows": n, "dim": d, "build_s": round(build_s, 1), "full_precision": measure()}
print("[c4] full precision:", json.dumps(out["full_precision"]), file=sys.stderr, flush=True)
keep = []
for M in [int(x) for x in a0.pq_m.split(",")]:
    train = base[:10000].cpu().numpy().copy()
    pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d, device=0)
    pq.Fit(train, np.arange(M) * 7 % 10000, alias=True)
    vs.attach(ix, pq)
    ix.set_tuning("pq_narrow", 0)
    rec = {"multi_wave": measure()}
    if M == 192:  # the variant with 32 tables per wave in LDS, 16 in registers: one query per CU (the default has two)
        ix.set_tuning("pq_narrow", 2)
        rec["multi_wave_one_per_cu"] = measure()
        ix.set_tuning("pq_narrow", 0)
    if a0.narrow:
        ix.set_tuning("pq_narrow", 1)
        rec["one_wave_global_table"] = measure()
        ix.set_tuning("pq_narrow", 0)
    out["M=%d" % M] = rec
    print("[c4] M=%d: %s" % (M, json.dumps(rec)), file=sys.stderr, flush=True)
    keep.append(pq)
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_c4_%d.json" % n, "w"), indent=1)
