"""K1 (sdb_distance_batch) at NQ x 1M x 384, METRIC cosine by default, one line: A/B runs of variant builds (SEMADB_AMD_LIB)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from semadb_amd import distance
d, nq, nc = 384, int(os.environ.get("NQ", 64)), 1000000
metric = os.environ.get("METRIC", "cosine")
q = bench.gen_rows(nq, d, 1, "gaussian", "cuda:0"); c = bench.gen_rows(nc, d, 2, "gaussian", "cuda:0")
for _ in range(2): distance.distance_batch(metric, q, c)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 10 if nq <= 256 else 3
for _ in range(reps): r = distance.distance_batch(metric, q, c)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(os.path.basename(os.environ.get("SEMADB_AMD_LIB", "default")), metric, nq, "ms %.3f" % (dt * 1e3), "Gpairs/s %.1f" % (nq * nc / dt / 1e9))
