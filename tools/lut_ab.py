"""What a quantized search call spends outside its walk kernel -- the query-table kernel (k_pq_lut_*; product.go:255-263)
and the gap behind it -- for the library named by SEMADB_AMD_LIB (default: the built one): ROWS x 768, batch 1 024,
M = 8 and 192.  call = HIP events around 30 device-resident search_batch calls, kernel = the walk kernel's own events
(sdb_index_set_profiling); a SHA-1 of the ids so that variants can be seen to agree."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana, vectorstore as vs
out = {"lib": os.environ.get("SEMADB_AMD_LIB", "default")}
n, d = int(os.environ.get("ROWS", 300000)), 768
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
q = bench.gen_rows(10 * 1024, d, 20250621, "latent:24", "cuda:0").view(10, 1024, d)
for M in (8, 192):
    ix = vamana.NewIndexVamana("ab", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    ix.insert_batch(None, base)
    pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d)
    pq.Fit(base[:10000].cpu().numpy().copy(), np.arange(M) * 7, alias=True)
    vs.attach(ix, pq)
    h = hashlib.sha1()
    for b in range(4):
        ids, _, _, _ = ix.search_batch(q[b], 10, 75)
        h.update(ids.cpu().numpy().tobytes())
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        ix.set_profiling(True)
        ix.profile_read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(30):
            ix.search_batch(q[r % 10], 10, 75)
        e1.record()
        torch.cuda.synchronize()
        kms = float(np.mean(ix.profile_read()[-30:]))
        ix.set_profiling(False)
        call = e0.elapsed_time(e1) / 30
        if best is None or call - kms < best[0] - best[1]:
            best = (call, kms)
    out["M%d" % M] = {"call_ms": round(best[0], 4), "kernel_ms": round(best[1], 4),
                      "outside_the_walk_us": round((best[0] - best[1]) * 1000, 1), "ids_sha1": h.hexdigest()[:12]}
    ix.close()
    pq.close()
print(json.dumps(out))
