"""C4-shaped measurement: Vamana + product quantizer (d = 768, K = 256, M = 8) -- K6 encode throughput, K5
LUT-distance search QPS and recall, next to the full-precision search on the same graph."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana, vectorstore as vs

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--M", type=int, default=8)
ap.add_argument("--K", type=int, default=256)
ap.add_argument("--dist", default="latent:24")
ap.add_argument("--train", type=int, default=10000)
a = ap.parse_args()
dev = "cuda:0"
n, d, nq = a.rows, a.dim, 1024
base = bench.gen_rows(n, d, 20250620, a.dist, dev)
queries = bench.gen_rows(4 * nq, d, 20250621, a.dist, dev).view(4, nq, d)
ix = vamana.NewIndexVamana("pq", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
t0 = time.time()
ix.insert_batch(None, base)
torch.cuda.synchronize()
out = {"rows": n, "dim": d, "M": a.M, "K": a.K, "build_s": round(time.time() - t0, 2)}
truth = bench.exact_topk(queries.view(-1, d), base, 10)[1] + 2
ix.set_profiling(True)


def measure(tag):
    for b in range(2):
        ix.search_batch(queries[b], 10, 75, trace=True)
    torch.cuda.synchronize()
    ix.profile_read()
    hits = 0
    nd = ne = 0
    for b in range(4):
        ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True)
        eq = (ids.to(torch.int64).unsqueeze(2) == truth[b * nq:(b + 1) * nq].unsqueeze(1)).any(2)
        hits += int(eq.sum().item())
        nd += int(tr.n_dist.to(torch.int64).sum().item())
        ne += int(tr.n_edges.to(torch.int64).sum().item())
    ms = float(np.median(ix.profile_read()))
    # whole call (LUT build, visited-set reset, kernel, id translation), device-resident in and out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(16):
        ix.search_batch(queries[r % 4], 10, 75)
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / 16
    out[tag] = {"kernel_ms": round(ms, 4), "qps": round(nq / ms * 1e3), "call_ms": round(call_ms, 4),
                "call_qps": round(nq / call_ms * 1e3), "recall@10": round(hits / (4 * nq * 10), 4),
                "mean_n_dist": round(nd / (4 * nq), 1)}
    return nd / 4, ne / 4


nd, ne = measure("full_precision")
out["full_precision"]["GB/s"] = round((nd * d * 4 + ne * 4) / out["full_precision"]["kernel_ms"] / 1e6, 1)
# productQuantizer.Fit on the first `train` rows (TriggerThreshold max, models/quantizer.go:62), then encode all
train = base[:a.train].cpu().numpy().copy()
pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(a.K, a.M, a.train), d)
t0 = time.time()
pq.Fit(train, np.arange(a.M) * 7 % a.train, alias=True)
out["fit_s"] = round(time.time() - t0, 2)
torch.cuda.synchronize()
t0 = time.time()
vs.attach(ix, pq)  # encodes every stored vector (K6)
torch.cuda.synchronize()
enc_s = time.time() - t0
out["encode_s"] = round(enc_s, 3)
out["encode_vectors_per_s"] = round((n + 1) / enc_s)
out["encode_GFLOP/s"] = round((n + 1) * d * a.K * 2 / enc_s / 1e9, 1)
nd, ne = measure("pq_lut")
out["pq_lut"]["code_GB/s"] = round((nd * a.M + ne * 4) / out["pq_lut"]["kernel_ms"] / 1e6, 1)
print(json.dumps(out))
