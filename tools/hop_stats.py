"""How uneven are the walks of one batch?  n_hop / n_dist distribution of 1 024-query batches at the headline shape:
what a batch that ends on its slowest walk can lose, and what helping waves could win back."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana
n, d = 1000000, 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
q = bench.gen_rows(4 * 1024, d, 20250621, "latent:24", "cuda:0").view(4, 1024, d)
ix = vamana.NewIndexVamana("hs", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
out = []
for b in range(4):
    _, _, _, tr = ix.search_batch(q[b], 10, 75, trace=True)
    h = tr.n_hop.cpu().numpy().astype(np.float64)
    nd = tr.n_dist.cpu().numpy().astype(np.float64)
    teams = nd.reshape(256, 4)
    out.append({"hops_mean": h.mean(), "hops_p50": float(np.percentile(h, 50)), "hops_p90": float(np.percentile(h, 90)),
                "hops_p99": float(np.percentile(h, 99)), "hops_max": h.max(),
                "n_dist_mean": nd.mean(), "n_dist_p99": float(np.percentile(nd, 99)), "n_dist_max": nd.max(),
                "idle_if_constant_rate": 1 - nd.mean() / nd.max(),
                "team_max_over_mean": float((teams.max(1) / teams.mean(1)).mean())})
print(json.dumps(out, indent=1))
