"""Latency of small calls (one REST request is one query, vamana.go:278-310): whole sdb_index_search_batch calls of
1 .. 1024 queries, device-resident in and out, with one wave per query and with the workgroup-per-query walk."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=384)
ap.add_argument("--dist", default="latent:24")
ap.add_argument("--metric", default="cosine")
a = ap.parse_args()
dev = "cuda:0"
base = bench.gen_rows(a.rows, a.dim, 20250620, a.dist, dev)
queries = bench.gen_rows(8192, a.dim, 20250621, a.dist, dev)
ix = vamana.NewIndexVamana("lat", vamana.IndexVectorVamanaParameters(a.dim, a.metric, 75, 64, 1.2), capacity=a.rows + 1)
ix.set_start(bench.start_vector(a.dim))
t0 = time.time()
ix.insert_batch(None, base)
torch.cuda.synchronize()
out = {"rows": a.rows, "dim": a.dim, "build_s": round(time.time() - t0, 2), "call_ms": {}}
for mode, name in ((1, "one_wave_per_query"), (2, "workgroup_per_query")):
    ix.set_tuning("wide_walk", mode)
    res = {}
    for nq in (1, 4, 16, 64, 128, 256, 512, 1024):
        reps = 40
        qs = [queries[(i * nq) % (8192 - nq):(i * nq) % (8192 - nq) + nq].contiguous() for i in range(reps + 5)]
        for i in range(5):
            ix.search_batch(qs[i], 10, 75)
        torch.cuda.synchronize()
        ts = []
        for i in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.search_batch(qs[5 + i], 10, 75)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[str(nq)] = {"p50": round(float(np.median(ts)), 4), "min": round(float(np.min(ts)), 4)}
    out["call_ms"][name] = res
print(json.dumps(out))
