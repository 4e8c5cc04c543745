"""How often the walk's next expansion is the candidate guessed one hop earlier (SDB_SPEC_STATS build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from semadb_amd import vamana
n, d = 1000000, 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
ix = vamana.NewIndexVamana("pv", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
q = bench.gen_rows(256, d, 20250621, "latent:24", "cuda:0")
ids, dd, c, tr = ix.search_batch(q, 10, 75, trace=True)
torch.cuda.synchronize()
h = tr.n_edges.float().mean().item(); nh = tr.n_hop.float().mean().item()
print("hops %.1f  hits %.1f  hit rate %.3f" % (nh, h, h / nh))
