"""Per-phase cycle sums of the SDB_STAMPS diagnostic build for the quantized search at a C4-like size.
usage: SEMADB_AMD_LIB=<stamps build of the library> python tools/stamps_c4.py  (ROWS / DIM env)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana, vectorstore as vs
n, d, nq = int(os.environ.get("ROWS", 4000000)), int(os.environ.get("DIM", 768)), 1024
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
ix = vamana.NewIndexVamana("st", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
M = int(os.environ.get("PQ_M", 8))
pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d)
pq.Fit(base[:10000].cpu().numpy().copy(), np.arange(M) * 7, alias=True)
vs.attach(ix, pq)
queries = bench.gen_rows(4 * nq, d, 20250621, "latent:24", "cuda:0").view(4, nq, d)
ix.set_profiling(True)
for b in range(4):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True, visit_cap=8)
torch.cuda.synchronize()
ms = ix.profile_read()
v = tr.visit_ids.cpu().numpy().astype(np.float64)[:, :4]
hops = tr.n_hop.float().mean().item()
print("kernel ms", ms, "hops", hops, "n_dist", tr.n_dist.float().mean().item())
print("mean cycles/query: adj %.0f atom %.0f vec %.0f ins %.0f  total %.0f" % (*v.mean(axis=0), v.sum(axis=1).mean()))
print("per hop (cycles): ", (v.mean(axis=0) / hops).round(0))
m4 = tr.visit_ids.cpu().numpy().astype(np.float64)[:, 4:8]
print("inside the merge, per hop (cycles): preamble %.0f, <2-candidates path %.0f, per-point pass %.0f, scatter %.0f"
      % tuple(m4.mean(axis=0) / hops))
