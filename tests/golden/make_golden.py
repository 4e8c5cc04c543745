"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path (all of its large tests use an unseeded RNG,
SURVEY.md section 8c) and cannot be run here (Go, no toolchain), so these fixtures freeze the oracle's
restatement: seeded inputs plus expected outputs as bit patterns.  tests/test_golden.py checks that the
oracle still reproduces them (CPU) and that the HIP path reproduces them (GPU).

    python tests/golden/make_golden.py          # rewrites the .npz files
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402
from tests.helpers import build_oracle_index, unit_rows  # noqa: E402


def distances():
    rng = np.random.default_rng(20250620)
    out = {}
    for d in (2, 31, 32, 33, 128, 384, 385, 768):
        q = (rng.standard_normal((3, d)) * 2).astype(np.float32)
        c = (rng.standard_normal((17, d)) * 2).astype(np.float32)
        out["q_%d" % d], out["c_%d" % d] = q, c
        for m in ("euclidean", "cosine", "dot"):
            out["%s_%d" % (m, d)] = orc.distance_matrix(q, c, m, orc.IMPL_ASM).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "distance_bits.npz"), **out)


def vamana(name, n, d, metric, R, L, nq, k, seed):
    rng = np.random.default_rng(seed)
    base = unit_rows(rng, n, d)
    o = build_oracle_index(orc, base, metric, R=R, L=L, seed=seed)
    ids, vecs, off, edges = o.export()
    q = unit_rows(rng, nq, d)
    res_ids = np.zeros((nq, k), np.uint64)
    res_d = np.zeros((nq, k), np.uint32)
    n_dist = np.zeros(nq, np.uint32)
    n_hop = np.zeros(nq, np.uint32)
    visits = np.zeros((nq, 256), np.uint64)
    for i in range(nq):
        r_ids, r_d, vis, tr = o.search(q[i], k, L)
        res_ids[i, :len(r_ids)] = r_ids
        res_d[i, :len(r_d)] = r_d.view(np.uint32)
        n_dist[i], n_hop[i] = tr.n_dist, tr.n_hop
        visits[i, :len(vis)] = vis
    np.savez_compressed(os.path.join(HERE, name), ids=ids, vecs=vecs, off=off.astype(np.uint32),
                        edges=edges.astype(np.uint32), queries=q, res_ids=res_ids, res_dist_bits=res_d,
                        n_dist=n_dist, n_hop=n_hop, visits=visits,
                        params=np.array([d, R, L, k], np.int64), metric=np.array(metric))


def pq():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((400, 32)).astype(np.float32)
    first = np.array([3, 50, 7, 200], np.int32)
    p = orc.PQ(32, "euclidean", 4, 16)
    xa = X.copy()
    codes = p.fit(xa, first, alias=True)
    qv = rng.standard_normal((5, 32)).astype(np.float32)
    lut = np.stack([p.lut(v) for v in qv])
    np.savez_compressed(os.path.join(HERE, "pq_400x32.npz"), X=X, first=first, codes=codes,
                        centroids_bits=p.flat_centroids.view(np.uint32), cdists_bits=p.centroid_dists.view(np.uint32),
                        X_after_bits=xa.view(np.uint32), queries=qv, lut_bits=lut.view(np.uint32))


def batched_build():
    """the device's round schedule (build.hip), restated by orc_index_insert_round: graphs for the default hub
    threshold and for one that sends nearly every multi-request target through the hub rule"""
    from tests.helpers import start_vector
    rng = np.random.default_rng(77)
    n, d, R, L = 3000, 32, 16, 30
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base = rng.standard_normal((n, 8)).astype(np.float32) @ lat + 0.15 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    sv = start_vector(np.random.default_rng(3), d)
    out = {"base": base, "start": sv, "params": np.array([d, R, L], np.int64), "metric": np.array("cosine")}
    for big_min in (512, 3):
        o = orc.Index(d, "cosine", R, L, 1.2)
        o.set_start(sv)
        assert o.insert_rounds(np.arange(2, n + 2, dtype=np.uint64), base, round_size=0, big_min=big_min) == 0
        ids, _, off, edges = o.export(with_vectors=False)
        out["off_%d" % big_min], out["edges_%d" % big_min] = off.astype(np.uint32), edges.astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "batched_build_3000x32_cosine.npz"), **out)


if __name__ == "__main__":
    distances()
    vamana("vamana_2000x32_cosine.npz", 2000, 32, "cosine", 32, 50, 24, 10, 11)
    vamana("vamana_1500x128_euclidean.npz", 1500, 128, "euclidean", 64, 75, 16, 10, 12)
    pq()
    batched_build()
    print("golden fixtures written to", HERE)
