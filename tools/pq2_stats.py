"""How often the two-wave quantized walk's walker names the next node itself (search_kernel.h pq2_walker): a
measurement build (-DSDB_PQ2_STATS, SEMADB_AMD_LIB) reports the hops the merger had to name in place of n_edges.
usage: SEMADB_AMD_LIB=build/variants/libsemadb_amd_pq2stats.so python tools/pq2_stats.py   (ROWS / DIM / PQ_M env)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana, vectorstore as vs
n, d, nq = int(os.environ.get("ROWS", 4000000)), int(os.environ.get("DIM", 768)), 1024
M = int(os.environ.get("PQ_M", 8))
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
ix = vamana.NewIndexVamana("st", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d)
pq.Fit(base[:10000].cpu().numpy().copy(), np.arange(M) * 7, alias=True)
vs.attach(ix, pq)
queries = bench.gen_rows(2 * nq, d, 20250621, "latent:24", "cuda:0").view(2, nq, d)
ix.set_profiling(True)
for b in range(2):
    ids, dd, c, tr = ix.search_batch(queries[b], 10, 75, trace=True, visit_cap=40)
torch.cuda.synchronize()
ms = ix.profile_read()
hops = tr.n_hop.float()
slow = tr.n_edges.float()
v = tr.visit_ids.cpu().numpy().astype(np.float64)[:, :8]
per_hop = (v.mean(axis=0) / hops.mean().item()).round(0)
print("walker cycles per hop: fetch + visited set + sums %d, waiting for the merger %d, naming + post %d, told by the merger %d; total %d"
      % (per_hop[0], per_hop[1], per_hop[2], per_hop[3], per_hop[4]))
print("merger cycles per hop: waiting for points %d, AddWithLimit %d, mark + answer %d" % (per_hop[5], per_hop[6], per_hop[7]))
# hand-over timeline of query 0, sequence numbers 17 .. 24 (shader clocks): walker posts the hop's points, merger sees them,
# merger answers, walker sees the answer
tl = tr.visit_ids.cpu().numpy().astype(np.int64)[0, 8:40].reshape(8, 4)
if tl.min() > 0:
    t0 = tl[0, 0]
    print("seq  post   seen(+)  answered(+)  walker saw(+)   next post(+)")
    for k in range(7):
        print("%3d %6d %8d %11d %14d %14d" % (17 + k, tl[k, 0] - t0, tl[k, 2] - tl[k, 0], tl[k, 3] - tl[k, 0], tl[k, 1] - tl[k, 0],
                                              tl[k + 1, 0] - tl[k, 0]))
print(json.dumps({"rows": n, "M": M, "kernel_ms": [round(float(x), 4) for x in ms], "hops_mean": round(hops.mean().item(), 2),
                  "hops_max": int(hops.max().item()), "merger_named_hops_mean": round(slow.mean().item(), 2),
                  "merger_named_share": round((slow.sum() / hops.sum()).item(), 4)}))
