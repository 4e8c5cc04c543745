"""Exploration harness (not the bench): build + search timing and recall on synthetic data."""
import argparse
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from semadb_amd import vamana


def gen(n, d, seed, dist, dev, latent=None):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if dist == "gaussian":
        x = torch.randn(n, d, generator=g, device=dev)
    else:
        k = int(dist.split(":")[1]) if ":" in dist else 32
        z = torch.randn(n, k, generator=g, device=dev)
        x = z @ latent + 0.1 * torch.randn(n, d, generator=g, device=dev)
    return torch.nn.functional.normalize(x, dim=1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--L", type=int, default=75)
    ap.add_argument("--Ls", type=str, default="")
    ap.add_argument("--R", type=int, default=64)
    ap.add_argument("--dist", default="gaussian")
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--round", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda:0"
    latent = None
    if a.dist.startswith("latent"):
        k = int(a.dist.split(":")[1]) if ":" in a.dist else 32
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        latent = torch.randn(k, a.d, generator=g, device=dev)
    base = gen(a.n, a.d, 20250620, a.dist, dev, latent)
    queries = gen(a.nq, a.d, 20250621, a.dist, dev, latent)
    sv = torch.rand(a.d, generator=torch.Generator().manual_seed(20250622)) * 2 - 1
    sv = (sv / sv.norm()).numpy().astype(np.float32)
    ix = vamana.NewIndexVamana("x", vamana.IndexVectorVamanaParameters(a.d, a.metric, 75, a.R, 1.2), capacity=a.n + 1, strict=False)
    ix.set_start(sv)
    torch.cuda.synchronize()
    t0 = time.time()
    ix.insert_batch(None, base, round_size=a.round)
    torch.cuda.synchronize()
    tb = time.time() - t0
    n_nodes, n_edges, _ = ix.stats()
    print(json.dumps({"n": a.n, "d": a.d, "build_s": round(tb, 2), "inserts_per_s": round(a.n / tb),
                      "avg_deg": round(n_edges / n_nodes, 2)}), flush=True)
    # ground truth, chunked (a single [nq, n] topk mis-indexes beyond 2^32 elements on this torch build)
    import bench
    if a.metric == "euclidean":
        truth = torch.cdist(queries, base).topk(10, dim=1, largest=False).indices + 2
    else:
        truth = bench.exact_topk(queries, base, 10)[1] + 2
    ix.set_profiling(True)
    for L in ([int(v) for v in a.Ls.split(",")] if a.Ls else [a.L]):
        ids, d, c, tr = ix.search_batch(queries, 10, L, trace=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            ids, d, c, tr = ix.search_batch(queries, 10, L, trace=True)
            ts.append(ix.last_search_ms())
        ids_np = ids.cpu().numpy()
        tnp = truth.cpu().numpy()
        hits = sum(len(set(ids_np[i].tolist()) & set(tnp[i].tolist())) for i in range(a.nq))
        nd = tr.n_dist.cpu().numpy().astype(np.int64)
        nh = tr.n_hop.cpu().numpy().astype(np.int64)
        ne = tr.n_edges.cpu().numpy().astype(np.int64)
        bytes_ = float((nd * a.d * 4 + ne * 4).sum())
        ms = float(np.median(ts))
        print(json.dumps({"L": L, "recall@10": round(hits / (a.nq * 10), 4), "kernel_ms": round(ms, 3),
                          "qps": round(a.nq / ms * 1e3), "n_dist": float(nd.mean()), "n_hop": float(nh.mean()),
                          "max_hop": int(nh.max()), "GB/s": round(bytes_ / ms / 1e6, 1),
                          "MB/query": round(bytes_ / a.nq / 1e6, 3)}), flush=True)


if __name__ == "__main__":
    main()
