"""Exact scan, euclidean (k_flat_scan with partial-distance pruning): 1 024 queries over 1M x 384, latent:24 and gaussian."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from semadb_amd import flat
n, d, nq = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384)), 1024
out = {"rows": n, "dim": d}
for dist in ("latent:24", "gaussian"):
    base = bench.gen_rows(n, d, 20250620, dist, "cuda:0")
    q = bench.gen_rows(nq, d, 20250621, dist, "cuda:0")
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "euclidean"), capacity=n + 1)
    ix.set_vectors(None, base)
    for k in (10, 75):
        ids, dd, c = flat.flat_search_batch(ix._h, d, q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ids, dd, c = flat.flat_search_batch(ix._h, d, q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        truth = bench.exact_topk(q, base, k)[1] + 2  # unit rows: the nearest by euclidean distance is the largest dot
        agree = float((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).float().mean().item())
        out["%s k=%d" % (dist, k)] = {"ms_per_call": round(ms, 2), "G_pairs_per_s": round(nq * n / ms / 1e6, 1),
                                      "agreement_with_matmul_topk": round(agree, 5)}
    ix.close()
    del base, q
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
