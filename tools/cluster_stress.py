"""Stress of the shared-device shard exchange through the ctypes Fanout: four shards on one GPU, client threads racing
requests of 1 .. 1 024 queries; every merged answer is compared with sdb_topk_merge of the shards' own answers."""
import sys, threading, time
sys.path.insert(0, "/root/repo")
import numpy as np
from semadb_amd import cluster, vamana
from tests.helpers import start_vector, unit_rows
rng = np.random.default_rng(1)
world, d, n = 4, 64, 4000
lat = rng.standard_normal((8, d)).astype(np.float32)
ixs = []
for s in range(world):
    x = rng.standard_normal((n, 8)).astype(np.float32) @ lat + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    base = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    ix = vamana.NewIndexVamana("s%d" % s, vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2))
    ix.set_start(start_vector(rng, d)); ix.insert_batch(None, base); ixs.append(ix)
ranks = cluster.Cluster.create_local([0] * world)
fan = cluster.Fanout(ranks, ixs)
qs = {nq: unit_rows(rng, nq, d) for nq in (1, 7, 64, 300, 1024)}
want = {}
for nq, q in qs.items():
    per = cluster.shard_limit(10, world, 75)
    res = [ix.search_batch(q, per, 75) for ix in ixs]
    ids = np.stack([r[0] for r in res]); dd = np.stack([r[1] for r in res]); c = np.stack([r[2] for r in res])
    want[nq] = cluster.topk_merge(ids, dd, c, 10)
bad = []; done = [0]
def client(j):
    r = np.random.default_rng(j)
    for _ in range(40):
        nq = int(r.choice(list(qs)))
        got = fan.search_points(qs[nq], 10, 75)
        w = want[nq]
        if not (np.array_equal(got[0], w[0]) and np.array_equal(got[2], w[2]) and np.array_equal(got[3], w[3])): bad.append((j, nq))
        done[0] += 1
t0 = time.time()
ts = [threading.Thread(target=client, args=(j,)) for j in range(12)]
[t.start() for t in ts]; [t.join() for t in ts]
print("requests", done[0], "bad", len(bad), "seconds %.2f" % (time.time() - t0), "next ticket", ranks[0].next_ticket())
