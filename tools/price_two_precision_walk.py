"""Pricing of an idea that is NOT built: a half-precision copy of the rows read first, the float32 row only
for candidates the walk might keep.  AddWithLimit discards a neighbour whose distance exceeds the candidate array's
last one (distset.go:184) and never looks at that distance again, so a neighbour whose distance is PROVABLY above the
threshold needs no exact evaluation.  This script replays greedySearch in numpy on an oracle-built graph (CPU only; the
oracle supplies graph and visit order, the replay re-derives every AddWithLimit decision) and counts, per query:

  evaluated   neighbours that pass CheckAndVisit (= n_dist - 1)
  kept        inserted into the candidate array (need the exact distance)
  decisive    discarded with a margin above eps (a bound on |float16-row distance - float32-row distance|: the rows are
              unit vectors, ||q|| ||y - y16|| <= 2^-11 -> 4.9e-4, plus the float32 summation's own rounding)
  ambiguous   discarded within eps of the threshold (would be fetched in float32 after all)

and the bytes a two-stage hop would read against the n_dist * d * 4 it reads now."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--dist", default="latent:24")
    ap.add_argument("--eps", type=float, default=6e-4)
    a = ap.parse_args()
    n, d, L = a.rows, a.dim, 75
    base = bench.gen_rows(n, d, 20250620, a.dist, "cpu").numpy()
    q = bench.gen_rows(a.queries, d, 20250621, a.dist, "cpu").numpy()
    o = oracle.Index(d, "cosine", 64, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    start = np.asarray(bench.start_vector(d), dtype=np.float32)
    o.set_start(start)
    assert o.insert_rounds(np.arange(2, n + 2, dtype=np.uint64), base) == 0
    ids, _, off, edges = o.export(with_vectors=False)
    slot = {int(v): i for i, v in enumerate(ids)}
    rows = np.vstack([start[None, :], base])  # id 1 = start node, id i + 2 = row i
    row_of = lambda node: 0 if node == 1 else node - 1
    tot = dict(evaluated=0, kept=0, decisive=0, ambiguous=0, hops=0)
    for qi in range(a.queries):
        _, _, visit, tr = o.search(q[qi], 10, L)
        cand = [(np.float32(1) - np.float32(rows[0] @ q[qi]), 1)]  # (distance, id), ascending
        seen = {1}
        for node in map(int, visit):
            s = slot[node]
            nb = [int(e) for e in edges[int(off[s]):int(off[s + 1])] if int(e) not in seen and int(e) in slot]
            seen.update(nb)
            if not nb:
                continue
            dist = (np.float32(1) - (rows[[row_of(e) for e in nb]] @ q[qi]).astype(np.float32))
            tot["hops"] += 1
            for dd, e in zip(dist, nb):
                tot["evaluated"] += 1
                if len(cand) == L and dd > cand[-1][0]:
                    tot["decisive" if dd > cand[-1][0] + a.eps else "ambiguous"] += 1
                    continue
                tot["kept"] += 1
                if len(cand) == L:
                    cand.pop()
                k = len(cand)
                while k > 0 and dd < cand[k - 1][0]:
                    k -= 1
                cand.insert(k, (dd, e))
        assert tr.n_dist - 1 == 0 or abs(tot["evaluated"]) > 0
    ev = tot["evaluated"]
    now = ev * d * 4
    two = ev * d * 2 + (tot["kept"] + tot["ambiguous"]) * d * 4
    print({"rows": n, "queries": a.queries, "eps": a.eps, **{k: v for k, v in tot.items()},
           "kept_frac": round(tot["kept"] / ev, 4), "decisive_frac": round(tot["decisive"] / ev, 4),
           "ambiguous_frac": round(tot["ambiguous"] / ev, 4), "bytes_two_stage_over_now": round(two / now, 4)})


if __name__ == "__main__":
    main()
