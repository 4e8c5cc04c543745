"""Per-phase cycle sums of the SDB_STAMPS diagnostic build for the quantized search (C4 shape at 1M)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana, vectorstore as vs

n, d, nq = 1000000, 384, 1024
z = np.load("/tmp/pv_graph_%d.npz" % n)
ix = vamana.NewIndexVamana("pv", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.load(z["ids"], z["vecs"], z["off"], z["edges"])
pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, 8, 10000), d)
pq.Fit(z["vecs"][1:10001].copy(), np.arange(8) * 7, alias=True)
vs.attach(ix, pq)
queries = bench.gen_rows(10 * nq, d, 20250621, "latent:24", "cuda:0").view(10, nq, d)
for b in range(4):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True, visit_cap=8)
torch.cuda.synchronize()
v = tr.visit_ids.cpu().numpy().astype(np.float64)[:, :4]
print("mean cycles/query: adj %.0f atom %.0f vec %.0f ins %.0f  total %.0f" % (*v.mean(axis=0), v.sum(axis=1).mean()))
print("per hop (cycles): ", (v.mean(axis=0) / tr.n_hop.float().mean().item()).round(0), "hops", tr.n_hop.float().mean().item())
