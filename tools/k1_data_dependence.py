"""Does K1's time depend on the data?  sdb_distance_batch of 64 x 1M x 384, cosine and euclidean, on latent:24 and
gaussian rows."""
import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, bench
from semadb_amd import distance
d, nq, nc = 384, 64, 1000000
for dist in ("latent:24", "gaussian"):
    c = bench.gen_rows(nc, d, 20250620, dist, "cuda:0"); q = bench.gen_rows(nq, d, 20250621, dist, "cuda:0")
    for metric in ("cosine", "euclidean"):
        for _ in range(2): distance.distance_batch(metric, q, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = distance.distance_batch(metric, q, c)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(dist, metric, "ms %.3f" % (dt * 1e3))
