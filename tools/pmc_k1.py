"""Workload for the rocprofv3 --pmc FETCH_SIZE pass over K1 (sdb_distance_batch, csrc/distance_tile.hip): the
k_index_distance calibration launch (known bytes), then 64 x 1M x 384 cosine and euclidean, three launches each.
What has to come from HBM with row reuse: (nq + nc) * d * 4 bytes (plus the swizzled query image, nq * d * 4)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import distance, vamana

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_k1_expected.json"
d, nq, nc, n = 384, 64, 1_000_000, 200_000
dev = "cuda:0"
base = bench.gen_rows(n, d, 1, "latent:24", dev)
ix = vamana.NewIndexVamana("cal", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
q = bench.gen_rows(nq, d, 2, "gaussian", dev)
c = bench.gen_rows(nc, d, 3, "gaussian", dev)
rng = np.random.default_rng(0)
ncal_q, ncal_c = 32, 32768
cand = rng.integers(2, n + 2, size=(ncal_q, ncal_c)).astype(np.uint64)
ix.distance_batch(q[:ncal_q], cand)
torch.cuda.synchronize()
for metric in ("cosine", "euclidean"):
    for _ in range(3):
        distance.distance_batch(metric, q, c)
    torch.cuda.synchronize()
json.dump({"dim": d, "nq": nq, "nc": nc, "unique_bytes": (nq + nc) * d * 4, "pairs_bytes": nq * nc * d * 4,
           "out_bytes": nq * nc * 4,
           "calibration": {"kernel": "k_index_distance", "bytes": ncal_q * ncal_c * d * 4}}, open(out, "w"), indent=1)
print("expected written", out)
