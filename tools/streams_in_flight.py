"""Serving-shaped measurement: how many queries/s one GPU answers when the host keeps S batches of `--batch`
queries in flight on S streams (the metric's own protocol is S = 1: one batch at a time, bench.py).  Same index,
same queries and the same C-ABI call as bench.py; queries and results stay in HBM.  Also checks that every
stream's results equal the one-batch-at-a-time results bit for bit."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from semadb_amd import vamana

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=384)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--dist", default="latent:24")
ap.add_argument("--rounds", type=int, default=60)
ap.add_argument("--streams", default="1,2,3,4,8")
a = ap.parse_args()
dev = "cuda:0"
n, d, nq = a.rows, a.dim, a.batch
nb = 16
base = bench.gen_rows(n, d, 20250620, a.dist, dev)
queries = bench.gen_rows(nb * nq, d, 20250621, a.dist, dev).view(nb, nq, d)
ix = vamana.NewIndexVamana("s", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
t0 = time.time()
ix.insert_batch(None, base)
torch.cuda.synchronize()
out = {"rows": n, "dim": d, "batch": nq, "build_s": round(time.time() - t0, 2), "streams": {}}
want = [ix.search_batch(queries[b], 10, 75)[:2] for b in range(nb)]
torch.cuda.synchronize()

for S in [int(s) for s in a.streams.split(",")]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    got = [None] * nb

    def sweep(rounds, keep):
        for r in range(rounds):
            for i, st in enumerate(streams):
                b = (r * S + i) % nb
                with torch.cuda.stream(st):
                    res = ix.search_batch(queries[b], 10, 75)
                if keep:
                    got[b] = res[:2]

    sweep(3, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep(a.rounds, False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sweep((nb + S - 1) // S, True)
    torch.cuda.synchronize()
    same = all(g is None or (torch.equal(g[0], w[0]) and torch.equal(g[1].view(torch.int32), w[1].view(torch.int32)))
               for g, w in zip(got, want))
    out["streams"][str(S)] = {"qps": round(a.rounds * S * nq / dt), "ms_per_batch": round(dt / (a.rounds * S) * 1e3, 4),
                              "identical": bool(same)}
print(json.dumps(out))
