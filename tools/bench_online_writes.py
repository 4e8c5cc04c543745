"""Online writes against a built index (InsertUpdateDelete, vamana.go:127-201): batches of 1 / 10 / 100 / 1 000 points
inserted into 1M x 384 (each batch is its own transaction: begin, rounds, commit), then deletes of the same sizes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
class A: metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2
n, d = int(os.environ.get("ROWS", 1000000)), 384
base = bench.gen_rows(n + 40000, d, 20250620, "latent:24", "cuda:0")
ix, bs = bench.build_index(A, base[:n], 0)
out = {"rows": n, "dim": d, "build_s": round(bs, 2)}
at = n
for size in (1, 10, 100, 1000):
    reps = 20 if size < 1000 else 10
    for _ in range(2):  # warm the workspaces of this batch size
        ix.insert_batch(None, base[at:at + size]); at += size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.insert_batch(None, base[at:at + size]); at += size
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["insert_%d" % size] = {"ms_per_batch": round(dt * 1e3, 3), "inserts_per_s": round(size / dt, 1)}
ids_next = 2 + n  # ids are dense from 2 (SDB_STARTID + 1)
for size in (1, 10, 100, 1000):
    reps = 10
    t0 = time.perf_counter()
    for r in range(reps):
        ids = np.arange(ids_next, ids_next + size, dtype=np.uint64); ids_next += size
        ix.delete_batch(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["delete_%d" % size] = {"ms_per_batch": round(dt * 1e3, 3), "deletes_per_s": round(size / dt, 1)}
print(json.dumps(out, indent=1))
