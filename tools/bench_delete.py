"""Delete / update path at the headline size: time of sdb_index_delete_batch (EdgeScan + pruneDeleteNeighbour +
stragglers, prune.go:88-154) for small and large delete sets on the 1M x 384 graph, and the search rate after."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

n, d = int(os.environ.get("ROWS", 1000000)), 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
q = bench.gen_rows(1024, d, 20250621, "latent:24", "cuda:0")
ix = vamana.NewIndexVamana("del", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 5001)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
torch.cuda.synchronize()
rng = np.random.default_rng(1)
live = np.arange(2, n + 2, dtype=np.uint64)
out = {"rows": n}
for m in (500, 10000, 100000):
    dels = rng.choice(live, size=m, replace=False)
    live = np.setdiff1d(live, dels)
    t0 = time.perf_counter()
    ix.delete_batch(dels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["delete_%d" % m] = {"seconds": round(dt, 3), "deletes_per_s": round(m / dt)}
# small write transactions (the REST path's shape): new points into the 1M graph, a few at a time
newv = bench.gen_rows(2000, d, 777, "latent:24", "cuda:0")
nid, used = n + 10, 0
for m in (1, 10, 100, 1000):
    reps = 20 if m <= 10 else 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        ix.insert_batch(np.arange(nid, nid + m, dtype=np.uint64), newv[used % 1000:used % 1000 + m].contiguous())
        nid += m
        used += m
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["insert_call_%d" % m] = {"ms_per_call": round(dt * 1e3, 2), "inserts_per_s": round(m / dt)}
ids, dd, c, _ = ix.search_batch(q, 10, 75)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ix.search_batch(q, 10, 75)
e1.record()
torch.cuda.synchronize()
out["search_after_ms"] = round(e0.elapsed_time(e1) / 20, 4)
out["results_full"] = bool((c == 10).all().item())
n_nodes, n_edges, _ = ix.stats()
out["nodes_left"] = int(n_nodes)
print(json.dumps(out))
