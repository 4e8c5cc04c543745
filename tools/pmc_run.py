"""Workload for the rocprofv3 --pmc passes: build the bench index, run (a) a calibration launch of
k_index_distance whose HBM bytes are known exactly (random rows of the slab, 16 B/lane row reads -- the
same access shape as the search kernel) and (b) a few search batches.  Writes the expected byte counts
next to the profiler output so the FETCH_SIZE correction factor can be derived per the microarch guide."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_expected.json"
n, d, nq = int(os.environ.get("PMC_N", 1000000)), 384, 1024
dev = "cuda:0"
base = bench.gen_rows(n, d, 20250620, "latent:24", dev)
queries = bench.gen_rows(5 * nq, d, 20250621, "latent:24", dev).view(5, nq, d)
ix = vamana.NewIndexVamana("pmc", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
torch.cuda.synchronize()
rng = np.random.default_rng(0)
ncal_q, ncal_c = 32, 65536
cand = rng.integers(2, n + 2, size=(ncal_q, ncal_c)).astype(np.uint64)
ix.distance_batch(queries[0][:ncal_q], cand)
torch.cuda.synchronize()
recs = []
for b in range(5):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True)
    torch.cuda.synchronize()
    nd = int(tr.n_dist.to(torch.int64).sum().item())
    ne = int(tr.n_edges.to(torch.int64).sum().item())
    recs.append({"n_dist": nd, "n_edges": ne, "alg_bytes": nd * d * 4 + ne * 4})
json.dump({"n": n, "dim": d, "calibration": {"kernel": "k_index_distance", "rows": ncal_q * ncal_c,
                                               "bytes": ncal_q * ncal_c * d * 4},
           "search": recs}, open(out, "w"), indent=1)
print("expected written", out)
