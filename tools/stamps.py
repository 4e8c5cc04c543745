"""Reads the per-phase cycle sums of the SDB_STAMPS diagnostic build (shares, not run time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

n, d, nq = 1000000, 384, 1024
z = np.load("/tmp/pv_graph_%d.npz" % n)
ix = vamana.NewIndexVamana("pv", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.load(z["ids"], z["vecs"], z["off"], z["edges"])
queries = bench.gen_rows(10 * nq, d, 20250621, "latent:24", "cuda:0").view(10, nq, d)
for b in range(3):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True, visit_cap=8)
torch.cuda.synchronize()
full = tr.visit_ids.cpu().numpy().astype(np.float64)
v = full[:, :4]
sub = full[:, 4:7]
print("inside vec, per hop (cycles): issue %.0f wait %.0f compute %.0f" % tuple(sub.mean(axis=0) / tr.n_hop.float().mean().item()))
tot = v.sum(axis=1)
print("mean cycles/query: adj %.0f atom %.0f vec %.0f ins %.0f  total %.0f" % (*v.mean(axis=0), tot.mean()))
print("shares: adj %.3f atom %.3f vec %.3f ins %.3f" % tuple(v.sum(axis=0) / v.sum()))
print("per hop (cycles): ", (v.mean(axis=0) / tr.n_hop.float().mean().item()).round(0))
