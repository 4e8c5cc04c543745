#!/usr/bin/env python3
"""K1 (sdb_distance_batch) micro-benchmark, device memory.  SURVEY 8d bills pairs * d * 4 bytes ("algorithmic"); with
row reuse (csrc/distance_tile.hip) what has to come from HBM is (nq + nc) * d * 4 + the nq * nc * 4 output bytes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from semadb_amd import distance

dev = "cuda:0"
out = {}
shapes = [(384, 64, 1000000), (384, 1024, 1000000), (128, 64, 2000000), (768, 64, 500000), (100, 64, 1000000)]
for d, nq, nc in shapes:
    q = bench.gen_rows(nq, d, 1, "gaussian", dev)
    c = bench.gen_rows(nc, d, 2, "gaussian", dev)
    for metric in ("cosine", "euclidean"):
        distance.distance_batch(metric, q, c)
        torch.cuda.synchronize()
        reps = 5 if nq <= 64 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            r = distance.distance_batch(metric, q, c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["%dx%dx%d %s" % (nq, nc, d, metric)] = {
            "ms": round(dt * 1e3, 3), "Gpairs/s": round(nq * nc / dt / 1e9, 2),
            "algorithmic_GB/s": round(nq * nc * d * 4 / dt / 1e9, 1),
            "hbm_unique_GB/s": round(((nq + nc) * d * 4 + nq * nc * 4) / dt / 1e9, 1)}
        del r
    del q, c
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
