"""Exact euclidean scan, 1 024 queries over ROWS x DIM (gaussian rows), k = 10: ms per call and agreement with a matmul
top-k.  SEMADB_AMD_LIB selects a variant build of the library (A/B measurements of k_flat_scan)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from semadb_amd import flat
n, d, nq = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384)), int(os.environ.get("NQ", 1024))
metric = os.environ.get("METRIC", "euclidean")
base = bench.gen_rows(n, d, 20250620, "gaussian", "cuda:0")
q = bench.gen_rows(nq, d, 20250621, "gaussian", "cuda:0")
ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
ix.set_vectors(None, base)
for _ in range(2):
    ids, dd, c = flat.flat_search_batch(ix._h, d, q, 10)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", 5))
t0 = time.perf_counter()
for _ in range(reps):
    ids, dd, c = flat.flat_search_batch(ix._h, d, q, 10)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
truth = bench.exact_topk(q, base, 10)[1] + 2
agree = float((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).float().mean().item())
print(json.dumps({"lib": os.path.basename(os.environ.get("SEMADB_AMD_LIB", "default")), "metric": metric, "rows": n, "dim": d,
                  "nq": nq, "ms_per_call": round(ms, 3), "G_pairs_per_s": round(nq * n / ms / 1e6, 1),
                  "agreement_with_matmul_topk": round(agree, 5), "sum_dist_bits": int(dd.view(torch.int32).to(torch.int64).sum().item())}))
