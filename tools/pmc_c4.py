"""Workload for the rocprofv3 --pmc pass over the quantized walk (verdict r02 #2: FETCH_SIZE for the PQ search kernel):
1M x 768 (PMC_N rows), K = 256, M from PMC_M (default 192, the multi-wave walk; 8 = the one-wave kernel with the table in
LDS).  Runs the k_index_distance calibration launch (known bytes) before the quantizer is attached, then five traced
search batches, and writes the algorithmic byte counts (n_dist * M code bytes + edge ids * 4, and the
per-query tables the kernel reads: nq * M * K * 4) next to the profiler output."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana, vectorstore as vs

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_c4_expected.json"
n, d, nq = int(os.environ.get("PMC_N", 1000000)), 768, 1024
M = int(os.environ.get("PMC_M", 192))
dev = "cuda:0"
base = bench.gen_rows(n, d, 20250620, "latent:24", dev)
queries = bench.gen_rows(5 * nq, d, 20250621, "latent:24", dev).view(5, nq, d)
ix = vamana.NewIndexVamana("pmc", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
torch.cuda.synchronize()
rng = np.random.default_rng(0)
ncal_q, ncal_c = 32, 32768
cand = rng.integers(2, n + 2, size=(ncal_q, ncal_c)).astype(np.uint64)
ix.distance_batch(queries[0][:ncal_q], cand)
torch.cuda.synchronize()
train = base[:10000].cpu().numpy().copy()
pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d, device=0)
pq.Fit(train, np.arange(M) * 7 % 10000, alias=True)
vs.attach(ix, pq)
recs = []
for b in range(5):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True)
    torch.cuda.synchronize()
    nd = int(tr.n_dist.to(torch.int64).sum().item())
    ne = int(tr.n_edges.to(torch.int64).sum().item())
    recs.append({"n_dist": nd, "n_edges": ne, "code_and_edge_bytes": nd * M + ne * 4, "table_bytes": nq * M * 256 * 4,
                 "code_rows_at_64B_sectors": nd * ((M + 63) // 64) * 64 + ne * 4})
json.dump({"n": n, "dim": d, "M": M, "calibration": {"kernel": "k_index_distance", "rows": ncal_q * ncal_c,
                                                     "bytes": ncal_q * ncal_c * d * 4}, "search": recs},
          open(out, "w"), indent=1)
print("expected written", out)
