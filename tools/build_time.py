"""C3 build time and the build's counters (sdb_index_build_stats) for the library in SEMADB_AMD_LIB."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana
n, d = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384))
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
res = []
for rep in range(3):
    ix = vamana.NewIndexVamana("v", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    for kv in filter(None, os.environ.get("AB_TUNE", "").split(",")):  # e.g. AB_TUNE=wide_walk=1
        key, value = kv.split("=")
        ix.set_tuning(key, int(value))
    torch.cuda.synchronize(); t0 = time.time()
    ix.insert_batch(None, base)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = ix.build_stats()
    res.append(round(dt, 3))
    ix.close()
print(json.dumps({"lib": os.environ.get("SEMADB_AMD_LIB", "default"), "tune": os.environ.get("AB_TUNE", ""), "build_s": res, "backedge_pairs": st.get("backedge_pairs"), "backedge_cached": st.get("backedge_cached"), "reprunes": st.get("reprunes")}))
