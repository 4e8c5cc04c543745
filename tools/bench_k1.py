"""K1 micro-benchmark: batched distance kernels, bytes = pairs * d * 4."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import distance, vamana

dev = "cuda:0"
out = {}
for d, nq, nc in ((384, 64, 1000000), (128, 64, 2000000), (768, 64, 500000)):
    q = bench.gen_rows(nq, d, 1, "gaussian", dev)
    c = bench.gen_rows(nc, d, 2, "gaussian", dev)
    for metric in ("cosine", "euclidean"):
        distance.distance_batch(metric, q, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = distance.distance_batch(metric, q, c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        # every query re-reads the candidate matrix: algorithmic bytes = nq * nc * d * 4 (served by L2/MALL/HBM)
        out["k_distance_batch d=%d %s" % (d, metric)] = {"ms": round(dt * 1e3, 3), "pairs/s": round(nq * nc / dt / 1e9, 2),
                                                         "alg_GB/s": round(nq * nc * d * 4 / dt / 1e9, 1)}
print(json.dumps(out, indent=1))
