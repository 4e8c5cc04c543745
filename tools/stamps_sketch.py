"""Per-hop cycle sums of the SDB_STAMPS diagnostic build (SEMADB_AMD_LIB) for the default walk and the two-precision
hop on the headline shape: adjacency round trip, visited-set test, distances (both stages), AddWithLimit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana
n, d, nq = int(os.environ.get("ROWS", 1000000)), 384, 1024
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
q = bench.gen_rows(4 * nq, d, 20250621, "latent:24", "cuda:0").view(4, nq, d)
ix = vamana.NewIndexVamana("st", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
for mode in (0, 1):
    ix.set_tuning("sketch", mode)
    for b in range(3):
        ids, dd, c, tr = ix.search_batch(q[b], 10, 75, trace=True, visit_cap=8)
    torch.cuda.synchronize()
    full = tr.visit_ids.cpu().numpy().astype(np.float64)
    hops = tr.n_hop.float().mean().item()
    v = full[:, :4]
    print("sketch=%d per hop (cycles): adj %.0f visited %.0f distances %.0f addwithlimit %.0f | total %.0f, hops %.1f; inside distances: issue %.0f wait %.0f compute %.0f" %
          ((mode,) + tuple(v.mean(axis=0) / hops) + (v.sum(axis=1).mean() / hops, hops) + tuple(full[:, 4:7].mean(axis=0) / hops)))
