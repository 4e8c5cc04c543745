"""Build-time comparison of the new-node prune kernel variants (tuning knob no_tile: 0 tiled 4 waves, 1 one-wave
kernel, 2 tiled 8 waves) on the C3 shape; prints seconds per build and checks that the graphs are identical."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana
n, d = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384))
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
out, ref = {}, None
for variant in [int(v) for v in os.environ.get("VARIANTS", "0,1,2,0").split(",")]:
    ix = vamana.NewIndexVamana("v", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
    ix.set_tuning("no_tile", variant)
    ix.set_start(bench.start_vector(d))
    torch.cuda.synchronize(); t0 = time.time()
    ix.insert_batch(None, base)
    torch.cuda.synchronize(); dt = time.time() - t0
    _, _, off, edges = ix.export(with_vectors=False)
    same = True if ref is None else bool(np.array_equal(off, ref[0]) and np.array_equal(edges, ref[1]))
    if ref is None: ref = (off, edges)
    out.setdefault(str(variant), []).append({"build_s": round(dt, 3), "same_graph": same})
    ix.close()
print(json.dumps(out))
