"""Exact scan (IndexFlat.Search, sdb_index_flat_search) of 1 024 queries over the C2 table: milliseconds per call,
agreement with the matmul ground truth, and the same through the block path (small table) for comparison."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import flat
n, d, nq = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384)), 1024
dev = "cuda:0"
base = bench.gen_rows(n, d, 20250620, "latent:24", dev)
q = bench.gen_rows(nq, d, 20250621, "latent:24", dev)
ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "cosine"), capacity=n + 1)
ix.set_vectors(None, base) if hasattr(ix, "set_vectors") else None
out = {"rows": n, "dim": d}
for k in (10, 75):
    ids, dd, c = flat.flat_search_batch(ix._h, d, q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ids, dd, c = flat.flat_search_batch(ix._h, d, q, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    truth = bench.exact_topk(q, base, k)[1] + 2
    agree = float((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).float().mean().item())
    out["k=%d" % k] = {"ms_per_call": round(ms, 2), "G_pairs_per_s": round(nq * n / ms / 1e6, 1),
                       "useful_TFLOP/s": round(2 * nq * n * d / ms / 1e9, 1), "agreement_with_matmul_topk": round(agree, 5)}
print(json.dumps(out))
