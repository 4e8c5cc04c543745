"""Wall time of three sdb_index_delete_batch calls of 1 000 points each on the 1M x 384 bench graph (under rocprofv3:
the delete path's kernel split)."""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
class A: metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2
n, d = 1000000, 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
ix, bs = bench.build_index(A, base, 0)
ids_next = 2
for r in range(3):
    ids = np.arange(ids_next, ids_next + 1000, dtype=np.uint64); ids_next += 1000
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.delete_batch(ids)
    torch.cuda.synchronize(); print("delete 1000: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
