"""Randomised differential run: device path (through the C ABI) against the oracle over randomly drawn shapes,
metrics, parameters and operation sequences -- a soak beyond the fixed cases of tests/.  Every trial builds the
same index on both sides (sequentially or in the batched round schedule), applies deletes / updates, optionally
switches the store to a product quantizer, and after every stage compares the exported graphs edge for edge and
a batch of searches (plain and filtered) id for id, distance bit for distance bit, visit for visit.

  python tools/fuzz_parity.py --trials 200 --seed 1      (prints one JSON line; exit code 1 on the first mismatch)
"""
import argparse
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as orc
from semadb_amd import vamana, vectorstore as vs
from tests.helpers import assert_same_graph, bits, start_vector

DIMS = [1, 2, 3, 7, 16, 24, 31, 32, 33, 48, 64, 65, 96, 100, 127, 128, 130, 160, 200, 256, 300, 384, 385, 512, 640, 768,
        784, 1024, 1280, 1536, 2048, 2112, 3072, 4096]
METRICS = ["euclidean", "cosine", "dot"]


def draw_rows(rng, n, d, kind):
    if kind == "grid":  # many equal distances
        return rng.integers(0, 3, size=(n, d)).astype(np.float32)
    if kind == "dups":  # repeated points
        pool = rng.standard_normal((max(2, n // 3), d)).astype(np.float32)
        return pool[rng.integers(0, pool.shape[0], n)].copy()
    if kind == "latent":
        k = max(1, min(8, d))
        x = rng.standard_normal((n, k)).astype(np.float32) @ rng.standard_normal((k, d)).astype(np.float32)
        x += 0.15 * rng.standard_normal((n, d)).astype(np.float32)
    else:
        x = rng.standard_normal((n, d)).astype(np.float32)
    nrm = np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return (x / np.maximum(nrm, np.float32(1e-20))).astype(np.float32)


def compare_exact(g, o, q, metric, tag):
    """full-precision store only: the exact scan (flat.go:76-132, first seen stays among equals, the start node is
    not a point) and K1 (plain.go:76-85; an unknown id gives MaxFloat32) against the oracle's distances"""
    from semadb_amd import flat
    ids, vecs, _, _ = o.export()
    keep = ids != 1
    ids, vecs = ids[keep], vecs[keep]
    dm = orc.distance_matrix(q, vecs, metric, orc.IMPL_ASM)
    k = min(10, len(ids))
    f_ids, f_d, f_c = flat.flat_search_batch(g._h, q.shape[1], q, k)
    for i in range(q.shape[0]):
        order = np.argsort(dm[i], kind="stable")[:k]
        assert int(f_c[i]) == k, (tag, "flat count", i)
        assert np.array_equal(f_ids[i], ids[order]) and np.array_equal(bits(f_d[i]), bits(dm[i, order])), (tag, "flat", i)
    pick = np.arange(0, len(ids), max(1, len(ids) // 16))[:16]
    cand = np.tile(np.concatenate([ids[pick], [np.uint64(10 ** 12)]]), (q.shape[0], 1)).astype(np.uint64)
    k1 = g.distance_batch(q, cand)
    assert np.array_equal(bits(k1[:, :-1]), bits(dm[:, pick])), (tag, "K1")
    assert (k1[:, -1] == np.finfo(np.float32).max).all(), (tag, "K1 unknown id")


def compare_searches(rng, g, o, d, kind, L, live, tag, metric=None):
    nq = 12
    q = draw_rows(rng, nq, d, kind)
    if metric is not None and len(live) >= 1:
        compare_exact(g, o, q, metric, tag)
    k = int(rng.integers(1, L + 1))
    sl = int(rng.integers(max(k, 1), 2 * L + 1))
    ids_g, d_g, c_g, tr = g.search_batch(q, k, sl, trace=True, visit_cap=2048)
    for i in range(nq):
        o_ids, o_d, o_vis, o_tr = o.search(q[i], k, sl)
        assert int(c_g[i]) == len(o_ids), (tag, "count", i)
        assert np.array_equal(ids_g[i, :len(o_ids)], o_ids), (tag, "ids", i)
        assert np.array_equal(bits(d_g[i, :len(o_ids)]), bits(o_d)), (tag, "dist bits", i)
        assert int(tr.n_dist[i]) == o_tr.n_dist and int(tr.n_hop[i]) == o_tr.n_hop, (tag, "counters", i)
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), (tag, "visit order", i)
    if len(live) >= 4:
        filters = []
        for i in range(nq):
            m = int(rng.integers(0, min(len(live), 3 * L) + 1))
            f = set(int(v) for v in rng.choice(live, size=m, replace=False)) if m else set()
            if i % 4 == 1:
                f |= {1, 10 ** 9 + i}  # the start id and an unknown id
            filters.append(f)
        ids_g, d_g, c_g, tr = g.search_batch(q, k, sl, filters=filters, trace=True, visit_cap=2048)
        for i in range(nq):
            o_ids, o_d, o_vis, o_tr = o.search(q[i], k, sl, filter_ids=sorted(filters[i]))
            assert int(c_g[i]) == len(o_ids), (tag, "filtered count", i)
            assert np.array_equal(ids_g[i, :len(o_ids)], o_ids), (tag, "filtered ids", i)
            assert np.array_equal(bits(d_g[i, :len(o_ids)]), bits(o_d)), (tag, "filtered dist bits", i)
            assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), (tag, "filtered visit order", i)
        # the same filters as bitmaps (sdb_index_search_batch_bitmap), and both walks forms for the id lists
        bm = vamana.FilterBitmaps.from_sets([set(v for v in f if v < 2 ** 30) for f in filters], align=int(rng.choice([1, 64])))
        trimmed = [set(v for v in f if v < 2 ** 30) for f in filters]
        b_ids, b_d, b_c, btr = g.search_batch(q, k, sl, filters=bm, trace=True, visit_cap=2048)
        l_ids, l_d, l_c, ltr = g.search_batch(q, k, sl, filters=trimmed, trace=True, visit_cap=2048)
        assert np.array_equal(b_ids, l_ids) and np.array_equal(bits(b_d), bits(l_d)) and np.array_equal(b_c, l_c), (tag, "bitmap filters")
        assert np.array_equal(btr.visit_ids, ltr.visit_ids) and np.array_equal(btr.n_dist, ltr.n_dist), (tag, "bitmap filters, visits")
        g.set_tuning("wide_walk", 1)
        w_ids, w_d, w_c, wtr = g.search_batch(q, k, sl, filters=filters, trace=True, visit_cap=2048)
        g.set_tuning("wide_walk", 0)
        assert np.array_equal(w_ids, ids_g) and np.array_equal(bits(w_d), bits(d_g)) and np.array_equal(w_c, c_g), (tag, "one wave per query, filtered")
        assert np.array_equal(wtr.visit_ids, tr.visit_ids), (tag, "one wave per query, filtered visits")


CURRENT = {}
VERBOSE = False
BUDGET = 300000


def explain(g, o, tag):
    o_ids, _, o_off, o_e = o.export(with_vectors=False)
    g_ids, _, g_off, g_e = g.export(with_vectors=False)
    if not np.array_equal(g_ids, o_ids):
        print(tag, ": id lists differ", len(g_ids), len(o_ids))
        return
    bad = [(int(o_ids[i]), g_e[g_off[i]:g_off[i + 1]].tolist(), o_e[o_off[i]:o_off[i + 1]].tolist())
           for i in range(len(o_ids)) if not np.array_equal(g_e[g_off[i]:g_off[i + 1]], o_e[o_off[i]:o_off[i + 1]])]
    print(tag, ": nodes differing", len(bad), "of", len(o_ids))
    for b in bad[:4]:
        print("   node %d\n     device %s\n     oracle %s" % b)


def check_graph(g, o):
    assert_same_graph(g, o)
    # outside a transaction the committed copy and the writer's copy of the graph are the same rows: a write path
    # that forgot to flag a row it changed shows up here
    assert g.version_diff() == 0, "graph versions differ after a committed write"


def trial(rng, t):
    d = int(rng.choice(DIMS))
    metric = str(rng.choice(METRICS))
    kind = str(rng.choice(["unit", "latent", "grid", "dups"], p=[0.35, 0.35, 0.15, 0.15]))
    R = int(rng.integers(4, 65))
    L = int(rng.integers(max(R // 2, 5), 101))
    alpha = float(rng.choice([1.0, 1.1, 1.2, 1.5]))
    budget = BUDGET  # rows * dim, keeps the oracle's sequential build in seconds
    n = int(rng.integers(50, max(60, min(3000 * max(1, BUDGET // 300000), budget // d))))
    batched = bool(rng.integers(0, 2))
    big_min = int(rng.choice([2, 3, 8, 512]))
    round_size = int(rng.choice([0, 0, 17, 64, 300]))
    desc = dict(trial=t, d=d, metric=metric, kind=kind, R=R, L=L, alpha=alpha, n=n, batched=batched,
                big_min=big_min, round_size=round_size)
    CURRENT.clear()
    CURRENT.update(desc)
    impl = orc.IMPL_AVX2 if orc.has_avx2() else orc.IMPL_ASM
    sv = start_vector(np.random.default_rng(int(rng.integers(1 << 30))), d)
    o = orc.Index(d, metric, R, L, alpha, impl=impl)
    o.set_start(sv)
    g = vamana.NewIndexVamana("fz", vamana.IndexVectorVamanaParameters(d, metric, L, R, alpha), strict=False)
    g.set_tuning("hub_min", big_min)
    g.set_start(sv)
    try:
        base = draw_rows(rng, n, d, kind)
        ids = np.arange(2, n + 2, dtype=np.uint64)
        if batched:
            assert o.insert_rounds(ids, base, round_size=round_size, big_min=big_min) == 0
            g.insert_batch(ids, base, round_size=round_size)
        else:
            for i in range(n):
                assert o.insert(int(ids[i]), base[i]) == 0
            g.insert_batch(ids, base, round_size=1)
        check_graph(g, o)
        live = [int(v) for v in ids]
        compare_searches(rng, g, o, d, kind, L, live, "after build", metric)
        quantized = False
        for step in range(int(rng.integers(1, 4))):
            if not quantized and d >= 4 and rng.integers(0, 3) == 0:
                M = int(rng.choice([m for m in (2, 4, 8, 16) if d % m == 0] or [0]))
                if M:
                    Kc = int(rng.choice([4, 16, 64, 256]))
                    o_ids, vecs, _, _ = o.export()
                    if len(o_ids) > Kc:
                        first = rng.integers(0, len(o_ids), M)
                        opq = orc.PQ(d, metric, M, Kc)
                        codes = opq.fit(vecs.copy(), first, alias=True)
                        assert o.attach_pq(opq, codes) == 0
                        gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(Kc, M), d)
                        gcodes = gpq.Fit(vecs.copy(), first, alias=True)
                        assert np.array_equal(gcodes, codes), "fit codes"
                        vs.attach(g, gpq, o_ids, gcodes)
                        quantized = True
                        desc["pq"] = CURRENT["pq"] = [M, Kc]
                        compare_searches(rng, g, o, d, kind, L, live, "after attach")
            n_del = int(rng.integers(0, max(1, len(live) // 4)))
            dels = [int(v) for v in rng.choice(live, size=n_del, replace=False)] if n_del else []
            rest = [v for v in live if v not in set(dels)]
            n_upd = int(rng.integers(0, min(10, len(rest)) + 1)) if rest else 0
            upds = [int(v) for v in rng.choice(rest, size=n_upd, replace=False)] if n_upd else []
            upd_vecs = draw_rows(rng, max(n_upd, 1), d, kind)
            n_ins = int(rng.integers(0, 40)) if rng.integers(0, 3) else int(rng.integers(40, 200))
            wr = int(rng.choice([1, 1, 0, 7, 64]))  # how the batch's new points are inserted: one by one or in rounds
            new_vecs = draw_rows(rng, max(n_ins, 1), d, kind)
            reuse = bool(dels) and bool(rng.integers(0, 2))
            if reuse:
                # freed ids come back, as the shard's id counter hands them out (idcounter.go:75-84): the deletes
                # are their own batch, the inserts that re-use the ids follow
                g.InsertUpdateDelete([vamana.IndexVectorChange(i, None) for i in dels], round_size=1)
                assert o.delete(np.array(dels, dtype=np.uint64)) == 0
                CURRENT["stage"] = "step %d: deletes ahead of id re-use (%d)" % (step, len(dels))
                check_graph(g, o)
                live = [v for v in live if v not in set(dels)]
                new_ids = dels[:n_ins]
                dels = []
            else:
                new_ids = []
            first_new = max(live + new_ids + dels + [1]) + 1 + int(rng.integers(0, 3))
            new_ids = new_ids + list(range(first_new, first_new + n_ins - len(new_ids)))
            ch = [vamana.IndexVectorChange(i, new_vecs[k]) for k, i in enumerate(new_ids)]
            ch += [vamana.IndexVectorChange(i, None) for i in dels]
            ch += [vamana.IndexVectorChange(i, upd_vecs[k]) for k, i in enumerate(upds)]
            CURRENT["stage"] = "step %d: %d inserts (%s, round_size %d), %d deletes, %d updates" % (
                step, len(new_ids), "re-used ids" if reuse else "fresh ids", wr, len(dels), len(upds))
            if os.environ.get("FUZZ_SPLIT"):  # debugging aid: the same three phases as separate calls, checked one by one
                for k, i in enumerate(new_ids):
                    # the search an insert starts with (insert.go:22), on both sides, before the point goes in
                    gi, gd, gc, gtr = g.search_batch(new_vecs[k:k + 1], 1, L, trace=True, visit_cap=2048)
                    oi, od, ovis, otr = o.search(new_vecs[k], 1, L)
                    same = (np.array_equal(gtr.visit_ids[0, :otr.n_hop], ovis) and int(gtr.n_hop[0]) == otr.n_hop
                            and int(gtr.n_dist[0]) == otr.n_dist)
                    print("insert %d: pre-search %s (hops %d/%d, dists %d/%d)" % (
                        i, "same" if same else "DIFFERS", int(gtr.n_hop[0]), otr.n_hop, int(gtr.n_dist[0]), otr.n_dist))
                    g.insert_batch(np.array([i], dtype=np.uint64), new_vecs[k:k + 1], round_size=1)
                    assert o.insert(i, new_vecs[k]) == 0
                    explain(g, o, "after insert of %d" % i)
                if dels or upds:
                    g.delete_batch(np.array(dels + upds, dtype=np.uint64))
                    assert o.delete(np.array(dels + upds, dtype=np.uint64)) == 0
                explain(g, o, "deletes")
                for k, i in enumerate(upds):
                    g.insert_batch(np.array([i], dtype=np.uint64), upd_vecs[k:k + 1], round_size=1)
                    assert o.insert(i, upd_vecs[k]) == 0
                    explain(g, o, "re-insert of %d" % i)
            else:
                between = None
                if rng.integers(0, 3) == 0:
                    # snapshot isolation: answers computed before the write must come back unchanged after every step
                    # inside the open transaction -- plain and filtered walks, the exact scan, K1; the filters and the
                    # K1 candidates name ids the transaction deletes, updates and (not yet visibly) inserts
                    sq = draw_rows(rng, 6, d, kind)
                    named = [int(v) for v in rng.choice(live, size=min(len(live), 20), replace=False)] + dels[:5] + upds + new_ids[:3]
                    sf = [set(named) for _ in range(6)]
                    cand = np.tile(np.array(named, dtype=np.uint64), (6, 1))
                    from semadb_amd import flat as _flat

                    def snapshot():
                        a1 = g.search_batch(sq, 5, L)
                        a2 = g.search_batch(sq, 5, L, filters=sf)
                        out = [a1[0].copy(), bits(a1[1]).copy(), a2[0].copy(), bits(a2[1]).copy()]
                        if not quantized:
                            a3 = _flat.flat_search_batch(g._h, d, sq, min(5, len(live)))
                            out += [a3[0].copy(), bits(a3[1]).copy(), bits(g.distance_batch(sq, cand)).copy()]
                        return out

                    pre = snapshot()

                    def between(tag):
                        now = snapshot()
                        for x, y in zip(pre, now):
                            assert np.array_equal(x, y), ("a search inside the transaction saw it", tag)
                if rng.integers(0, 3) == 0:
                    # a rehearsal that is called off: the same inserts, deletes and updates inside a transaction that is
                    # ABORTED (sdb_index_abort_write) -- the graph must be the oracle's untouched one again, and the real
                    # write below must still build what the oracle builds
                    CURRENT["stage"] += " [after an aborted rehearsal]"
                    g.begin_write()
                    if new_ids:
                        g.insert_batch(np.array(new_ids, dtype=np.uint64), new_vecs[:len(new_ids)], round_size=wr)
                    if dels or upds:
                        g.delete_batch(np.array(dels + upds, dtype=np.uint64))
                    for k, i in enumerate(upds):
                        g.insert_batch(np.array([i], dtype=np.uint64), upd_vecs[k:k + 1], round_size=1)
                    assert g.abort_write() is True
                    assert g.version_diff() == 0
                    check_graph(g, o)
                g.InsertUpdateDelete(ch, round_size=wr, _between=between)
                if wr == 1:
                    for k, i in enumerate(new_ids):
                        assert o.insert(i, new_vecs[k]) == 0
                elif new_ids:
                    assert o.insert_rounds(np.array(new_ids, dtype=np.uint64), new_vecs[:len(new_ids)], round_size=wr,
                                           big_min=big_min) == 0
                if dels or upds:
                    assert o.delete(np.array(dels + upds, dtype=np.uint64)) == 0
                for k, i in enumerate(upds):
                    assert o.insert(i, upd_vecs[k]) == 0
            live = sorted((set(live) - set(dels)) | set(new_ids))
            if rng.integers(0, 3) == 0:  # tombstones squeezed out: nothing observable may change
                g.compact()
                assert g.row_usage()[1] == 0
            check_graph(g, o)
            compare_searches(rng, g, o, d, kind, L, live, "after write batch %d" % step, None if quantized else metric)
    finally:
        g.close()
    return desc


def merge_trial(rng):
    """the shard fan-out's merge (cluster/actions.go:357-376) on random ragged per-shard results with ties"""
    from semadb_amd import cluster
    n_shards = int(rng.integers(1, 17))
    per = int(rng.integers(1, 76))
    limit = int(rng.integers(1, 101))
    nq = int(rng.integers(1, 70))
    d = np.sort(rng.integers(0, int(rng.choice([4, 50, 100000])), size=(n_shards, nq, per)).astype(np.float32) / 8, axis=2)
    if rng.integers(0, 2):
        d = -d[:, :, ::-1].copy()  # dot-product distances are negative
    ids = rng.integers(2, 10 ** 6, size=(n_shards, nq, per)).astype(np.uint64)
    counts = rng.integers(0, per + 1, size=(n_shards, nq)).astype(np.uint32)
    o_ids, o_d, o_s, o_c = cluster.topk_merge(ids, d, counts, limit)
    for q in range(nq):
        w_ids, w_d, w_s = orc.cluster_merge(ids[:, q, :], d[:, q, :], counts[:, q].astype(np.int32), limit)
        n = len(w_ids)
        assert int(o_c[q]) == n, ("merge count", n_shards, per, limit, q)
        assert np.array_equal(o_ids[q, :n], w_ids) and np.array_equal(bits(o_d[q, :n]), bits(w_d)), ("merge", q)
        assert np.array_equal(o_s[q, :n].astype(np.int32), w_s), ("merge shards", q)


def pq_trial(rng):
    """productQuantizer.Fit / encode / LUT / symmetric distance (product.go, utils/kmeans.go) on random shapes and
    degenerate data (ties, repeated points, fewer distinct points than centroids)"""
    M = int(rng.choice([2, 3, 4, 8, 16]))
    sub = int(rng.choice([1, 2, 3, 4, 8, 24, 32, 33, 64, 100]))
    d = M * sub
    K = int(rng.choice([2, 3, 16, 64, 255, 256]))
    metric = str(rng.choice(METRICS))
    kind = str(rng.choice(["unit", "latent", "grid", "dups"]))
    n = int(rng.integers(1, 4)) * K + int(rng.integers(0, 50)) if rng.integers(0, 4) else int(rng.integers(2, K + 2))
    n = max(2, min(n, max(2, 400000 // (d * max(1, K // 8)))))
    X = draw_rows(rng, n, d, kind)
    first = rng.integers(0, n, M)
    alias = bool(rng.integers(0, 2))
    CURRENT.clear()
    CURRENT.update(dict(pq_trial=True, d=d, M=M, K=K, metric=metric, kind=kind, n=n, alias=alias))
    xo, xg = X.copy(), X.copy()
    opq = orc.PQ(d, metric, M, K)
    o_codes = opq.fit(xo, first, alias=alias)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    try:
        g_codes = gpq.Fit(xg, first, alias=alias)
        assert np.array_equal(g_codes, o_codes), "fit codes"
        fc, cd = gpq.codebook()
        assert np.array_equal(bits(fc), bits(opq.flat_centroids)), "codebook"
        assert np.array_equal(bits(cd), bits(opq.centroid_dists)), "centroid-pair table"
        assert np.array_equal(bits(xg), bits(xo)), "aliasing write-through"
        V = draw_rows(rng, 20, d, kind)
        ge = gpq.encode(V)
        assert np.array_equal(ge, np.stack([opq.encode(v) for v in V])), "encode"
        Q = draw_rows(rng, 4, d, kind)
        got = gpq.lut_distance(Q, ge)
        want = np.array([[opq.dist_lut(opq.lut(q), c) for c in ge] for q in Q], dtype=np.float32)
        assert np.array_equal(bits(got), bits(want)), "LUT distance"
        perm = rng.permutation(20)
        gs = gpq.sym_distance(ge, ge[perm])
        ws = np.array([opq.dist_sym(ge[i], ge[perm[i]]) for i in range(20)], dtype=np.float32)
        assert np.array_equal(bits(gs), bits(ws)), "symmetric distance"
    finally:
        gpq.close()


def flat_trial(rng):
    """flat.IndexFlat (shard/index/flat/flat.go): random Set / replace / Delete sequences, explicit transactions with
    a search in between (which must not see them), compaction, and exact searches with random limits and filters
    against the oracle's distances over a storage-order model of the store (Set of a stored id drops its row and
    appends the new one).  One trial in four is a table of 33 000+ rows: the streaming scans (matrix cores for dot /
    cosine, with and without a tail; packed FMAs for euclidean rows of whole blocks), tombstones included."""
    from semadb_amd import flat
    metric = str(rng.choice(METRICS))
    kind = str(rng.choice(["unit", "latent", "grid", "dups"]))
    big = rng.integers(0, 4) == 0
    d = int(rng.choice([32, 33, 64, 96, 100, 128, 300, 384])) if big else int(rng.choice([1, 2, 3, 31, 32, 33, 64, 100, 128, 200, 384, 768]))
    n0 = int(rng.integers(33000, 42000)) if big else int(rng.integers(1, max(2, min(3000, 200000 // d))))
    CURRENT.clear()
    CURRENT.update(dict(flat_trial=True, d=d, metric=metric, kind=kind, n0=n0))
    if VERBOSE:
        print("  flat:", CURRENT, file=sys.stderr, flush=True)
    ids, rows = [], []

    def m_set(i, v):
        m_del(i)
        ids.append(int(i)), rows.append(np.asarray(v, dtype=np.float32))

    def m_del(i):
        if int(i) in pos_of():
            k = ids.index(int(i))
            del ids[k], rows[k]

    def pos_of():
        return set(ids)

    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric))
    try:
        base = draw_rows(rng, n0, d, kind)
        first_ids = rng.permutation(np.arange(1, 3 * n0 + 1))[:n0].astype(np.uint64) if rng.integers(0, 2) else \
            np.arange(5, n0 + 5, dtype=np.uint64)
        ix.set_vectors(first_ids, base)
        ids.extend(int(v) for v in first_ids), rows.extend(base)
        next_id = int(first_ids.max()) + 1

        def check(tag):
            if VERBOSE:
                print("  flat check", tag, len(ids), file=sys.stderr, flush=True)
            if not ids:
                return
            nq = int(rng.integers(1, 20))
            q = draw_rows(rng, nq, d, kind)
            limit = int(rng.choice([1, 3, 10, 75, 128]))
            if VERBOSE:
                print("    nq %d limit %d" % (nq, limit), file=sys.stderr, flush=True)
            I, B = np.array(ids, dtype=np.uint64), np.stack(rows)
            dm = orc.distance_matrix(q, B, metric, orc.IMPL_ASM)
            allowed = None
            if rng.integers(0, 3) == 0:
                pool = np.concatenate([I, np.array([10 ** 9 + 1], dtype=np.uint64)])
                allowed = [set(int(v) for v in rng.choice(pool, size=min(len(pool), int(rng.integers(1, 60))), replace=False))
                           for _ in range(nq)]
            if VERBOSE:
                print("    filtered %s" % (allowed is not None), file=sys.stderr, flush=True)
            g_ids, g_d, g_c = ix.search_batch(q, limit, filters=allowed)
            for i in range(nq):
                idx = np.arange(len(I)) if allowed is None else np.array([j for j in range(len(I)) if int(I[j]) in allowed[i]], dtype=np.int64)
                order = idx[np.argsort(dm[i, idx], kind="stable")][:limit]
                assert int(g_c[i]) == len(order), (tag, "flat count", i, int(g_c[i]), len(order))
                assert np.array_equal(g_ids[i, :len(order)], I[order]), (tag, "flat ids", i)
                assert np.array_equal(bits(g_d[i, :len(order)]), bits(dm[i, order])), (tag, "flat distances", i)

        check("initial")
        for step in range(int(rng.integers(2, 6))):
            ch = []
            live = list(ids)
            for _ in range(int(rng.integers(1, 40))):
                op = int(rng.integers(0, 5))
                if op <= 1:  # new point
                    ch.append(flat.IndexVectorChange(next_id, draw_rows(rng, 1, d, kind)[0]))
                    next_id += 1
                elif op == 2 and live:  # replace
                    ch.append(flat.IndexVectorChange(int(rng.choice(live)), draw_rows(rng, 1, d, kind)[0]))
                elif op == 3 and live:  # delete
                    ch.append(flat.IndexVectorChange(int(rng.choice(live)), None))
                else:  # delete of a missing id
                    ch.append(flat.IndexVectorChange(next_id + 10 ** 6, None))
            explicit = rng.integers(0, 3) == 0
            if VERBOSE:
                print("  flat step %d: %d changes, explicit %s: %s" % (step, len(ch), explicit, [(c.Id, c.Vector is None) for c in ch]), file=sys.stderr, flush=True)
            if explicit:  # the same changes by hand inside one transaction, with a search that must not see them
                ix.begin_write()
                for c in ch:
                    if c.Vector is None:
                        ix.remove_vectors([c.Id])
                    else:
                        ix.set_vectors(np.array([c.Id], dtype=np.uint64), c.Vector.reshape(1, -1))
                check("inside transaction %d" % step)
                ix.commit()
            else:
                ix.InsertUpdateDelete(ch)
            for c in ch:
                m_del(c.Id) if c.Vector is None else m_set(c.Id, c.Vector)
            assert ix.version_diff() == 0, ("flat version diff", step)
            r_rows, r_dead = ix.row_usage()
            assert r_rows - r_dead == len(ids), ("flat row usage", step, r_rows, r_dead, len(ids))
            if rng.integers(0, 3) == 0:
                if VERBOSE:
                    print("  flat compact", ix.row_usage(), file=sys.stderr, flush=True)
                ix.compact()
                assert ix.row_usage() == (len(ids), 0), ("flat compaction", step)
            check("after step %d" % step)
    finally:
        ix.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=0, help="stop after this much wall time (0 = run all trials)")
    ap.add_argument("--only", type=int, default=-1, help="run just this trial number")
    ap.add_argument("--verbose", action="store_true", help="print every trial and stage as it starts (to find a crash)")
    ap.add_argument("--budget", type=int, default=300000, help="rows * dim of a trial's index at most (and 3000 rows per 300000)")
    a = ap.parse_args()
    global BUDGET, VERBOSE
    BUDGET = a.budget
    VERBOSE = a.verbose
    t0 = time.time()
    done = 0
    dims = set()
    for t in ([a.only] if a.only >= 0 else range(a.trials)):
        rng = np.random.default_rng([a.seed, t])
        try:
            t1 = time.time()
            if a.verbose:
                print("trial %d: merge" % t, file=sys.stderr, flush=True)
            merge_trial(rng)
            t2 = time.time()
            if a.verbose:
                print("trial %d: pq" % t, file=sys.stderr, flush=True)
            pq_trial(rng)
            pq_desc = dict(CURRENT)
            if a.verbose:
                print("trial %d: flat %s" % (t, pq_desc), file=sys.stderr, flush=True)
            flat_trial(rng)
            t3 = time.time()
            if a.verbose:
                print("trial %d: index %s" % (t, CURRENT), file=sys.stderr, flush=True)
            desc = trial(rng, t)
            if time.time() - t1 > 20:
                print("slow trial %d: merge %.1fs, pq %.1fs %s, index %.1fs %s" % (
                    t, t2 - t1, t3 - t2, pq_desc, time.time() - t3, CURRENT), file=sys.stderr)
        except Exception:
            traceback.print_exc()
            print(json.dumps({"failed_trial": t, "seed": a.seed, "trials_passed": done, "config": CURRENT}))
            sys.exit(1)
        dims.add(desc["d"])
        done += 1
        if a.seconds and time.time() - t0 > a.seconds:
            break
    print(json.dumps({"trials_passed": done, "seed": a.seed, "seconds": round(time.time() - t0, 1),
                      "distinct_dims": len(dims), "mismatches": 0}))


if __name__ == "__main__":
    main()
