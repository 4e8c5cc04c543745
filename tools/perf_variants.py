"""A/B harness for kernel variants: `build` saves one 1M graph to /tmp, `run` loads it with whatever
library SEMADB_AMD_LIB points at and times the search kernel (interleaved rounds inside one process are
not possible across different .so files, so every variant sees the identical graph and queries)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

mode = sys.argv[1]
n, d, nq = int(os.environ.get("PV_N", 1000000)), 384, 1024
dist = os.environ.get("PV_DIST", "latent:24")
path = "/tmp/pv_graph_%d.npz" % n
params = vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2)
if mode == "build":
    base = bench.gen_rows(n, d, 20250620, dist, "cuda:0")
    ix = vamana.NewIndexVamana("pv", params, capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    t0 = time.time()
    ix.insert_batch(None, base)
    torch.cuda.synchronize()
    print("build %.1fs" % (time.time() - t0))
    ids, vecs, off, edges = ix.export()
    np.savez(path, ids=ids, vecs=vecs, off=off, edges=edges)
    print("saved", path)
else:
    z = np.load(path)
    ix = vamana.NewIndexVamana("pv", params, capacity=n + 1)
    ix.load(z["ids"], z["vecs"], z["off"], z["edges"])
    queries = bench.gen_rows(10 * nq, d, 20250621, dist, "cuda:0").view(10, nq, d)
    ix.set_profiling(True)
    L = int(os.environ.get("PV_L", 75))
    for b in range(3):
        ix.search_batch(queries[b], 10, L, trace=True)
    torch.cuda.synchronize()
    ix.profile_read()
    algb = 0
    reps = int(os.environ.get("PV_REPS", 30))
    t0 = time.perf_counter()
    for b in range(reps):
        ids, dd, c, tr = ix.search_batch(queries[b % 10], 10, L, trace=True)
        if b < 10:
            algb += int(tr.n_dist.to(torch.int64).sum().item()) * d * 4 + int(tr.n_edges.to(torch.int64).sum().item()) * 4
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = ix.profile_read()
    chk = int(ids.to(torch.int64).sum().item())
    print(json.dumps({"lib": os.environ.get("SEMADB_AMD_LIB", "default"), "kernel_ms_med": round(float(np.median(ms)), 4),
                      "kernel_ms_min": round(float(ms.min()), 4), "GB/s": round(float(algb / 10 / np.median(ms) / 1e6), 1),
                      "qps_kernel": round(float(nq / np.median(ms) * 1e3)), "wall_qps": round(nq * reps / wall), "chk": chk}))
