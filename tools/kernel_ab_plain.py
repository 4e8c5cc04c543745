"""A/B of the headline kernel only (1M x 384, batch 1 024): median / min kernel ms of 60 launches and a checksum of the
result ids; SEMADB_AMD_LIB selects the library.  One graph build, so alternating runs on one box are cheap."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana
n, d = 1000000, 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
q = bench.gen_rows(20 * 1024, d, 20250621, "latent:24", "cuda:0").view(20, 1024, d)
ix = vamana.NewIndexVamana("ab", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
ix.set_profiling(True)
h = hashlib.sha1()
for b in range(3):
    ix.search_batch(q[b], 10, 75)
torch.cuda.synchronize()
ix.profile_read()
for r in range(60):
    ids, _, _, _ = ix.search_batch(q[r % 20], 10, 75)
    if r < 20:
        h.update(ids.cpu().numpy().tobytes())
torch.cuda.synchronize()
ms = ix.profile_read()
print(json.dumps({"lib": os.environ.get("SEMADB_AMD_LIB", "default"), "kernel_ms_median": round(float(np.median(ms)), 4),
                  "kernel_ms_min": round(float(ms.min()), 4), "ids_sha1": h.hexdigest()[:12]}))
