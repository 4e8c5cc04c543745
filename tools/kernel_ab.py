"""A/B harness for search-kernel variants: the library named by SEMADB_AMD_LIB (default: the built one) on
(a) the headline shape 1M x 384, batch 1024 and (b) a quantized index ROWS_PQ x 768, M = PQ_M (8); median kernel ms of 30
launches each, plus a checksum of the result ids so that variants can be seen to agree."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana, vectorstore as vs
out = {"lib": os.environ.get("SEMADB_AMD_LIB", "default"), "tune": os.environ.get("AB_TUNE", "")}


def run(n, d, with_pq):
    base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
    q = bench.gen_rows(10 * 1024, d, 20250621, "latent:24", "cuda:0").view(10, 1024, d)
    ix = vamana.NewIndexVamana("ab", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    ix.insert_batch(None, base)
    for kv in filter(None, os.environ.get("AB_TUNE", "").split(",")):  # e.g. AB_TUNE=team_walk=1
        key, value = kv.split("=")
        ix.set_tuning(key, int(value))
    if with_pq:
        M = int(os.environ.get("PQ_M", 8))
        pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(256, M, 10000), d)
        pq.Fit(base[:10000].cpu().numpy().copy(), np.arange(M) * 7, alias=True)
        vs.attach(ix, pq)
    ix.set_profiling(True)
    h = hashlib.sha1()
    for b in range(3):
        ix.search_batch(q[b], 10, 75)
    torch.cuda.synchronize()
    ix.profile_read()
    for r in range(30):
        ids, _, _, _ = ix.search_batch(q[r % 10], 10, 75)
        if r < 10:
            h.update(ids.cpu().numpy().tobytes())
    torch.cuda.synchronize()
    ms = ix.profile_read()
    ix.close()
    return {"kernel_ms_median": round(float(np.median(ms)), 4), "kernel_ms_min": round(float(ms.min()), 4), "ids_sha1": h.hexdigest()[:12]}


if not os.environ.get("SKIP_PLAIN"):
    out["plain_1Mx384"] = run(1000000, 384, False)
out["pq_%dx768_M%s" % (int(os.environ.get("ROWS_PQ", 4000000)), os.environ.get("PQ_M", "8"))] = run(int(os.environ.get("ROWS_PQ", 4000000)), 768, True)
print(json.dumps(out))
