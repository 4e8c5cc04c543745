"""per-call times of small filtered batches (the filter_10 outlier of profiles/r03_filter_hash.json)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

class A:
    metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2

base = bench.gen_rows(1000000, 384, 20250620, "latent:24", "cuda:0")
queries = bench.gen_rows(4 * 1024, 384, 20250621, "latent:24", "cuda:0").view(4, 1024, 384)
ix, _ = bench.build_index(A, base, 0)
rng = np.random.default_rng(3)
for size in (10, 1000, 10, 100):
    filt = [np.sort(rng.choice(1000000, size=size, replace=False).astype(np.uint64) + 2) for _ in range(1024)]
    off = np.zeros(1025, dtype=np.uint64)
    off[1:] = np.cumsum([len(f) for f in filt])
    flat = np.concatenate(filt)
    ts = []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.search_batch(queries[i % 4], 10, 75, filters=(off, flat))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(size, " ".join("%.2f+%.2f" % t for t in ts), flush=True)
