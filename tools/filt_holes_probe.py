#!/usr/bin/env python3
"""Where a filtered call on a table with holes spends its time: the calls one by one, device table against host map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench


class A:
    metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2


rows, dim = int(os.environ.get("ROWS", 200000)), 128
base = bench.gen_rows(rows, dim, 1, "latent:24", "cuda:0")
q = bench.gen_rows(1024, dim, 2, "latent:24", "cuda:0")
ix, _ = bench.build_index(A, base, 0)
rng = np.random.default_rng(3)
filt = [np.sort(rng.choice(rows, size=1000, replace=False).astype(np.uint64) + 2) for _ in range(1024)]
off = np.zeros(1025, dtype=np.uint64)
off[1:] = np.cumsum([len(f) for f in filt])
flat = np.concatenate(filt)


def run(label):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.search_batch(q, 10, 75, filters=(off, flat))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(label, " ".join("%.2f" % t for t in ts), flush=True)


run("dense, device")
ix.delete_batch(np.sort(rng.choice(rows, size=500, replace=False).astype(np.uint64) + 2))
run("holes, device table")
ix.set_tuning("host_filters", 1)
run("holes, host map")
