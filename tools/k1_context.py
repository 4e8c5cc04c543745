"""Measurement: K1 euclidean (64 x 1M x 384) at several points of a bench-like process -- is its time context-dependent?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from semadb_amd import distance, flat
class A: metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2
d, nq, n = 384, 64, 1000000
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
queries = bench.gen_rows(4 * 1024, d, 20250621, "latent:24", "cuda:0").view(4, 1024, d)
q64 = queries[0][:64].contiguous()
def k1(tag):
    for metric in ("cosine", "euclidean"):
        distance.distance_batch(metric, q64, base)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = distance.distance_batch(metric, q64, base)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        del r
        print(tag, metric, "ms %.3f" % (dt * 1e3))
k1("fresh")
ix, bs = bench.build_index(A, base, 0)
k1("after build")
for b in range(20): ix.search_batch(queries[b % 4], 10, 75)
torch.cuda.synchronize()
k1("after searches")
fx = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "euclidean"), capacity=n + 1)
fx.set_vectors(None, base)
k1("with a second copy of the rows resident")
fx.close()
k1("after closing it")
