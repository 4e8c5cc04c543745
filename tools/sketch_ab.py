"""Two-precision hop (SDB_TUNE_SKETCH) against the default walk on the headline shape: same ids, distance bits, visit
counters for every timed batch; the audit's count of contradicted decisions; kernel time both ways."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semadb_amd import vamana

n, d = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("DIM", 384))
dist = os.environ.get("DIST", "latent:24")
base = bench.gen_rows(n, d, 20250620, dist, "cuda:0")
q = bench.gen_rows(12 * 1024, d, 20250621, dist, "cuda:0").view(12, 1024, d)
ix = vamana.NewIndexVamana("ab", vamana.IndexVectorVamanaParameters(d, os.environ.get("METRIC", "cosine"), 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
out = {"rows": n, "dim": d, "dist": dist}


def run(tag):
    res = []
    for b in range(2):
        ix.search_batch(q[b], 10, 75)
    torch.cuda.synchronize()
    ix.set_profiling(True)
    ix.profile_read()
    for b in range(2, 12):
        ids, dd, cnt, _ = ix.search_batch(q[b], 10, 75)
        res.append((ids.cpu().numpy(), dd.cpu().numpy().view(np.uint32), cnt.cpu().numpy()))
    torch.cuda.synchronize()
    ms = [float(v) for v in ix.profile_read()][-10:]
    ix.set_profiling(False)
    tr = []
    for b in range(2, 5):
        _, _, _, t = ix.search_batch(q[b], 10, 75, trace=True)
        tr.append((t.n_dist.cpu().numpy(), t.n_hop.cpu().numpy(), t.n_edges.cpu().numpy()))
    out[tag] = {"kernel_ms_avg": round(float(np.mean(ms)), 4), "kernel_ms_min": round(float(np.min(ms)), 4)}
    return res, tr


ref, ref_tr = run("default")
for mode, tag in ((2, "sketch_audit"), (1, "sketch")):
    ix.set_tuning("sketch", mode)
    got, got_tr = run(tag)
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(ref, got))
    same_tr = all(all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(ref_tr, got_tr))
    st = ix.sketch_stats()
    out[tag].update({"identical_results": bool(same), "identical_counters": bool(same_tr), "discarded": st[0], "contradicted": st[1], "in_use": st[2]})
ix.set_tuning("sketch", 0)
again, _ = run("default_again")
out["default_again"]["identical_results"] = bool(all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, again)))
print(json.dumps(out))
