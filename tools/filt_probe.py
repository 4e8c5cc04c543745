"""Per-call times (enqueue / complete) of filtered searches on 200k x 384: 1 / 64 / 1 024 queries with 10 or 1 000 filter
ids each -- where a filtered call's time goes on the host."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
class A: metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2
dev="cuda:0"
base = bench.gen_rows(200000, 384, 1, "latent:24", dev)
q = bench.gen_rows(1024, 384, 2, "latent:24", dev)
ix,_ = bench.build_index(A, base, 0)
rng = np.random.default_rng(1)
for nq in (1, 64, 1024):
    for size in (10, 1000):
        filt = [np.sort(rng.choice(200000, size=size, replace=False).astype(np.uint64)+2) for _ in range(nq)]
        off = np.zeros(nq+1, dtype=np.uint64); off[1:] = np.cumsum([len(f) for f in filt]); flat = np.concatenate(filt)
        for _ in range(2): ix.search_batch(q[:nq], 10, 75, filters=(off, flat))
        torch.cuda.synchronize()
        ts=[]
        for _ in range(5):
            t0=time.perf_counter(); ix.search_batch(q[:nq], 10, 75, filters=(off, flat)); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
            ts.append(((t1-t0)*1e3,(t2-t0)*1e3))
        print(nq, size, ["%.2f/%.2f"%t for t in ts])
