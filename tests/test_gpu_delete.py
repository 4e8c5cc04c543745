"""Delete / update path on device (vamana.go:175-251, prune.go, node.go:142-199): identical graphs to the
oracle's restatement, plus the reference's own invariants (shard/shard_vector_test.go:129-245)."""
import numpy as np
import pytest

from tests.helpers import bits, start_vector, unit_rows

pytestmark = pytest.mark.gpu


def _pair(oracle, d, metric, R, L, base, seed=3):
    """the same sequentially built graph on both sides"""
    from semadb_amd import vamana
    sv = start_vector(np.random.default_rng(seed), d)
    o = oracle.Index(d, metric, R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    for i in range(base.shape[0]):
        assert o.insert(i + 2, base[i]) == 0
    g = vamana.NewIndexVamana("d", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    g.set_start(sv)
    g.insert_batch(None, base, round_size=1)
    return o, g


def _same_graph(o, g):
    o_ids, o_v, o_off, o_e = o.export()
    g_ids, g_v, g_off, g_e = g.export()
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(g_off, o_off), "degree sequence differs"
    assert np.array_equal(g_e, o_e), "edge lists differ"
    assert np.array_equal(bits(g_v), bits(o_v))


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("d,n,R,L,ndel", [(2, 200, 8, 25, 20), (16, 400, 4, 20, 80), (64, 500, 16, 30, 100),
                                          (128, 400, 32, 50, 40), (384, 300, 32, 50, 30)])
def test_delete_matches_oracle(oracle, metric, d, n, R, L, ndel):
    rng = np.random.default_rng(d + n + ndel)
    base = unit_rows(rng, n, d) if d > 2 else rng.random((n, d), dtype=np.float32)
    # (un-normalised 2-d data under "cosine" with R = 8 strands > 100 nodes per delete: the start node grows far
    # past the 64 entries of its row, on both sides)
    o, g = _pair(oracle, d, metric, R, L, base)
    _same_graph(o, g)
    dels = rng.choice(np.arange(2, n + 2), size=ndel, replace=False).astype(np.uint64)
    dels_with_unknown = np.concatenate([dels, [10 ** 9]]).astype(np.uint64)  # unknown ids are skipped
    assert o.delete(dels_with_unknown) == 0
    g.delete_batch(dels_with_unknown)
    _same_graph(o, g)
    n_nodes, _, _ = g.stats()
    assert n_nodes == n + 1 - ndel
    # searches walk the same path and never return a deleted id
    q = unit_rows(rng, 16, d) if d > 2 else rng.random((16, d), dtype=np.float32)
    ids, dist, cnt, tr = g.search_batch(q, 10, L, trace=True, visit_cap=512)
    for i in range(16):
        o_ids, o_d, o_vis, o_tr = o.search(q[i], 10, L)
        assert np.array_equal(ids[i, :len(o_ids)], o_ids) and np.array_equal(bits(dist[i, :len(o_ids)]), bits(o_d))
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis)
        assert not (set(int(v) for v in o_ids) & set(int(v) for v in dels))
    # a second round of deletes on the already-holed index, then re-use of freed ids (idcounter.go:75-88)
    left = np.setdiff1d(np.arange(2, n + 2), dels)
    dels2 = rng.choice(left, size=max(1, ndel // 2), replace=False).astype(np.uint64)
    assert o.delete(dels2) == 0
    g.delete_batch(dels2)
    _same_graph(o, g)
    newv = unit_rows(rng, 5, d) if d > 2 else rng.random((5, d), dtype=np.float32)
    for k in range(5):
        assert o.insert(int(dels[k]), newv[k]) == 0
    g.insert_batch(dels[:5], newv, round_size=1)
    _same_graph(o, g)
    g.close()


def test_update_is_delete_plus_reinsert(oracle):
    # vamana.go:170-174 (classification), :223-227 (inbound edges removed), :247-251 (re-inserted one by one)
    from semadb_amd import vamana
    rng = np.random.default_rng(9)
    base = unit_rows(rng, 300, 32)
    o, g = _pair(oracle, 32, "euclidean", 16, 30, base)
    upd = [7, 100, 250]
    newv = unit_rows(rng, 3, 32)
    changes = [vamana.IndexVectorChange(i, newv[k]) for k, i in enumerate(upd)]
    changes.append(vamana.IndexVectorChange(40, None))        # delete
    changes.append(vamana.IndexVectorChange(999999, None))    # delete of a missing id: skipped
    changes.append(vamana.IndexVectorChange(5000, unit_rows(rng, 1, 32)[0]))  # insert
    g.InsertUpdateDelete(changes)
    assert o.insert(5000, changes[-1].Vector) == 0            # inserts first (vamana.go:190-201)
    assert o.delete(np.array([40] + upd, dtype=np.uint64)) == 0
    for k, i in enumerate(upd):
        assert o.insert(i, newv[k]) == 0
    _same_graph(o, g)
    rset, res = g.Search(vamana.SearchVectorVamanaOptions(newv[1], 30, 5))
    assert res[0].NodeId == 100 and res[0].Distance == 0       # the updated vector is what is found
    g.close()


def test_reference_shard_invariants_after_delete_reinsert(oracle):
    """shard/shard_vector_test.go:668-700 pattern at reduced size: insert, delete 500, re-insert, search;
    checkNoReferences (:187-214), checkConnectivity (:150-185), checkPointCount (:129-148), self retrieval"""
    from semadb_amd import vamana
    rng = np.random.default_rng(17)
    n, d = 4000, 32
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base = rng.standard_normal((n, 8)).astype(np.float32) @ lat
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    g = vamana.NewIndexVamana("s", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2))
    g.set_start(start_vector(rng, d))
    g.insert_batch(None, base)
    dels = rng.choice(np.arange(2, n + 2), size=500, replace=False).astype(np.uint64)
    g.InsertUpdateDelete([vamana.IndexVectorChange(int(i), None) for i in dels])

    def check(expected):
        ids, vecs, off, edges = g.export()
        assert len(ids) == expected + 1                                  # checkPointCount (+ start node)
        idset = set(int(v) for v in ids)
        assert all(int(e) in idset for e in edges)                       # checkNoReferences
        pos = {int(v): i for i, v in enumerate(ids)}
        seen, stack = set(), [1]
        while stack:
            v = stack.pop()
            if v in seen:
                continue
            seen.add(v)
            stack.extend(int(e) for e in edges[int(off[pos[v]]):int(off[pos[v] + 1])])
        assert len(seen) - 1 == expected                                 # checkConnectivity
        assert int(np.diff(off.astype(np.int64)).max()) <= 64

    check(n - 500)
    assert not (set(int(v) for v in g.search_batch(base[:200], 10, 75)[0].ravel()) & set(int(v) for v in dels))
    g.insert_batch(dels, base[dels.astype(np.int64) - 2])               # re-insert the same ids
    check(n)
    ids, dist, cnt, _ = g.search_batch(base[:300], 10, 75)
    assert np.array_equal(ids[:, 0], np.arange(2, 302, dtype=np.uint64)) and np.all(dist[:, 0] <= 1e-6)
    g.close()


def test_flat_scan_skips_deleted(oracle):
    from semadb_amd import flat, vamana
    rng = np.random.default_rng(2)
    base = unit_rows(rng, 200, 16)
    o, g = _pair(oracle, 16, "euclidean", 8, 25, base)
    g.delete_batch(np.array([2, 3, 50], dtype=np.uint64))
    ids, d, c = flat.flat_search_batch(g._h, 16, base[:3], 5)
    assert not (set(int(v) for v in ids.ravel()) & {1, 2, 3, 50})
    assert int(ids[2, 0]) == 4 and d[2, 0] == 0
    g.close()


def test_delete_id_rules():
    # vamana.go:150-157
    from semadb_amd import vamana, SemaDBError
    g = vamana.NewIndexVamana("r", vamana.IndexVectorVamanaParameters(2, "euclidean"))
    g.set_start([0.6, 0.8])
    g.InsertUpdateDelete([vamana.IndexVectorChange(5, [0.1, 0.2])])
    for bad in (0, 1):
        with pytest.raises(SemaDBError):
            g.delete_batch(np.array([bad], dtype=np.uint64))
    g.delete_batch(np.array([77], dtype=np.uint64))  # unknown: no-op
    g.delete_batch(np.array([5], dtype=np.uint64))
    rset, res = g.Search(vamana.SearchVectorVamanaOptions([0.1, 0.2], 75, 10))
    assert res == []
    g.close()


def test_start_row_overflow_keeps_graph_valid(oracle):
    """More stragglers than the 64-edge start row can take (the reference's start node has no bound, node.go:73-80):
    the graph equals the oracle's, stays well-formed, and every live point is still found."""
    from semadb_amd import flat, vamana
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(1)
    base = rng.random((200, 2), dtype=np.float32)
    o, g = _pair(oracle, 2, "cosine", 8, 25, base)
    dels = rng.choice(np.arange(2, 202), size=20, replace=False).astype(np.uint64)
    g.delete_batch(dels)
    assert o.delete(dels) == 0
    assert_same_graph(g, o)
    ids, vecs, off, edges = g.export()
    assert len(ids) == 181 and int(off[1] - off[0]) > 64
    idset = set(int(v) for v in ids)
    assert all(int(e) in idset for e in edges)
    for i in range(len(ids)):
        row = [int(e) for e in edges[int(off[i]):int(off[i + 1])]]
        assert len(row) == len(set(row)) and int(ids[i]) not in row and (i == 0 or len(row) <= 64)
    f_ids, f_d, f_c = flat.flat_search_batch(g._h, 2, base[:50], 5)
    assert np.all(f_c == 5) and not (set(int(v) for v in f_ids.ravel()) & (set(int(v) for v in dels) | {1}))
    # the walk still works from the start node
    s_ids, _, s_c, _ = g.search_batch(base[:50], 5, 25)
    assert np.all(s_c == 5) and not (set(int(v) for v in s_ids.ravel()) & set(int(v) for v in dels))
    g.close()


@pytest.mark.parametrize("metric,quantized", [("euclidean", False), ("cosine", False), ("euclidean", True)])
def test_mixed_operations_differential(oracle, metric, quantized):
    """A random sequence of write batches -- inserts, updates (delete + re-insert, vamana.go:170-174,247-251) and
    deletes -- applied to both sides; after every batch the graphs are identical and searches walk the same path.
    With `quantized` the store is switched to a fitted product quantizer half way (all later distances are table
    distances)."""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import assert_same_graph
    d, R, L = 32, 12, 30
    rng = np.random.default_rng(2024 + quantized)
    sv = start_vector(np.random.default_rng(9), d)
    o = oracle.Index(d, metric, R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    g = vamana.NewIndexVamana("mix", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    g.set_start(sv)
    live, next_id = [], 2
    for step in range(14):
        changes = []
        n_ins = int(rng.integers(20, 60))
        for _ in range(n_ins):
            changes.append((next_id, unit_rows(rng, 1, d)[0]))
            next_id += 1
        n_del = int(rng.integers(0, min(15, len(live)) + 1)) if live else 0
        dels = [int(v) for v in rng.choice(live, size=n_del, replace=False)] if n_del else []
        rest = [v for v in live if v not in dels]
        n_upd = int(rng.integers(0, min(8, len(rest)) + 1)) if rest else 0
        upds = [int(v) for v in rng.choice(rest, size=n_upd, replace=False)] if n_upd else []
        upd_vecs = unit_rows(rng, max(n_upd, 1), d)
        # device side: one InsertUpdateDelete per batch
        ch = [vamana.IndexVectorChange(i, v) for i, v in changes]
        ch += [vamana.IndexVectorChange(i, None) for i in dels]
        ch += [vamana.IndexVectorChange(i, upd_vecs[k]) for k, i in enumerate(upds)]
        g.InsertUpdateDelete(ch, round_size=1)
        # oracle side: the same order vamana.go applies -- inserts, one removeInboundEdges over deleted + updated,
        # then the updated points re-inserted one by one
        for i, v in changes:
            assert o.insert(i, v) == 0
        if dels or upds:
            assert o.delete(np.array(dels + upds, dtype=np.uint64)) == 0
        for k, i in enumerate(upds):
            assert o.insert(i, upd_vecs[k]) == 0
        live = [v for v in live if v not in dels] + [i for i, _ in changes]
        assert_same_graph(g, o)
        if quantized and step == 6:
            ids, vecs, _, _ = o.export()
            first = rng.integers(0, len(ids), 4)
            opq = oracle.PQ(d, metric, 4, 16)
            codes = opq.fit(vecs.copy(), first, alias=True)
            assert o.attach_pq(opq, codes) == 0
            gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(16, 4), d)
            gcodes = gpq.Fit(vecs.copy(), first, alias=True)
            assert np.array_equal(gcodes, codes)
            vs.attach(g, gpq, ids, gcodes)
        q = unit_rows(rng, 6, d)
        ids_g, d_g, c_g, tr = g.search_batch(q, 5, L, trace=True, visit_cap=256)
        for i in range(6):
            o_ids, o_d, o_vis, o_tr = o.search(q[i], 5, L)
            assert np.array_equal(ids_g[i, :len(o_ids)], o_ids) and np.array_equal(bits(d_g[i, :len(o_ids)]), bits(o_d))
            assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis)
    g.close()


@pytest.mark.parametrize("metric,d,n,R,dup,quantized", [("cosine", 48, 1500, 4, False, False),
                                                        ("euclidean", 16, 900, 5, True, False),
                                                        ("euclidean", 32, 1200, 4, False, True)])
def test_start_node_overflow_list(oracle, metric, d, n, R, dup, quantized):
    """Stragglers of a delete are appended to the start node with no bound (prune.go:131-151, node.go:73-80).  A
    small degree bound and a large delete leave hundreds of them: the start node's list outgrows its 64-entry row
    and stays that long until it is next pruned.  Through every stage -- the overflowing delete, searches that
    expand the long list, export / load of it, a second delete that hits edges of the list (pruneDeleteNeighbour
    over row + overflow), an insert whose back-edge re-prunes the start node -- graphs and walks equal the
    oracle's."""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import assert_same_graph
    L = 20
    rng = np.random.default_rng(n + R)
    sv = start_vector(np.random.default_rng(5), d)
    base = unit_rows(rng, n, d)
    if dup:  # repeated points: zero distances and ties on top
        base = base[rng.integers(0, n // 3, n)].copy()
    o = oracle.Index(d, metric, R, L, 1.5, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    g = vamana.NewIndexVamana("ovf", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.5), strict=False)
    g.set_start(sv)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    for i in range(n):
        assert o.insert(int(ids[i]), base[i]) == 0
    g.insert_batch(ids, base, round_size=1)
    if quantized:
        o_ids, vecs, _, _ = o.export()
        first = rng.integers(0, len(o_ids), 4)
        opq = oracle.PQ(d, metric, 4, 16)
        codes = opq.fit(vecs.copy(), first, alias=True)
        assert o.attach_pq(opq, codes) == 0
        gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(16, 4), d)
        assert np.array_equal(gpq.Fit(vecs.copy(), first, alias=True), codes)
        vs.attach(g, gpq, o_ids, codes)

    def start_degree():
        e_ids, _, off, _ = o.export(with_vectors=False)
        assert e_ids[0] == 1
        return int(off[1] - off[0])

    def same_walks():
        q = unit_rows(rng, 16, d)
        g_ids, g_d, g_c, tr = g.search_batch(q, 5, L, trace=True, visit_cap=512)
        for i in range(16):
            o_ids, o_d, o_vis, o_tr = o.search(q[i], 5, L)
            assert np.array_equal(g_ids[i, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d))
            assert int(tr.n_dist[i]) == o_tr.n_dist and int(tr.n_edges[i]) == o_tr.n_edges
            assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis)

    live = list(int(v) for v in ids)
    dels = rng.choice(live, size=n // 8, replace=False).astype(np.uint64)
    g.delete_batch(dels)
    assert o.delete(dels) == 0
    live = sorted(set(live) - set(int(v) for v in dels))
    assert start_degree() > 64 + 64, "the case must overflow the row by more than one chunk (%d)" % start_degree()
    assert_same_graph(g, o)
    assert g.stats()[1] == o.export(with_vectors=False)[3].size
    same_walks()
    # the long list survives export -> load
    e_ids, e_v, e_off, e_e = g.export()
    g2 = vamana.NewIndexVamana("ovf2", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.5), strict=False)
    g2.load(e_ids, e_v, e_off, e_e)
    r_ids, _, r_off, r_e = g2.export(with_vectors=False)
    assert np.array_equal(r_ids, e_ids) and np.array_equal(r_off, e_off) and np.array_equal(r_e, e_e)
    g2.close()
    # a second delete: more stragglers on top of the list, and deleted ids among its edges
    _, _, off, edges = o.export(with_vectors=False)
    in_list = [int(v) for v in edges[off[0] + 64:off[1]]]
    dels2 = np.array(in_list[:7] + [int(v) for v in rng.choice([v for v in live if v not in set(in_list)], 40, replace=False)],
                     dtype=np.uint64)
    g.delete_batch(dels2)
    assert o.delete(dels2) == 0
    live = sorted(set(live) - set(int(v) for v in dels2))
    assert_same_graph(g, o)
    same_walks()
    # build the list up again, then an insert: its back-edge to the start node re-prunes row + list + new point
    dels3 = rng.choice(live, size=len(live) // 6, replace=False).astype(np.uint64)
    g.delete_batch(dels3)
    assert o.delete(dels3) == 0
    live = sorted(set(live) - set(int(v) for v in dels3))
    assert_same_graph(g, o)
    long_before = start_degree()
    new = unit_rows(rng, 30, d)
    new_ids = np.arange(n + 10, n + 40, dtype=np.uint64)
    for i in range(30):
        assert o.insert(int(new_ids[i]), new[i]) == 0
    g.insert_batch(new_ids, new, round_size=1)
    assert_same_graph(g, o)
    if long_before > 64 and not dup and not quantized:  # (in the other cases no new node picks the start node)
        assert start_degree() <= R, "an insert was expected to re-prune the start node"
    same_walks()
    g.close()


def test_start_node_overflow_batched_round(oracle, monkeypatch):
    """the same state met by a batched insert: the round's requests to the long start node are taken all at once
    (the hub rule), as the oracle's restatement of the round schedule does"""
    from semadb_amd import vamana
    from tests.helpers import assert_same_graph
    d, n, R, L = 24, 1200, 5, 20
    rng = np.random.default_rng(77)
    sv = start_vector(np.random.default_rng(5), d)
    base = unit_rows(rng, n, d)
    o = oracle.Index(d, "euclidean", R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    g = vamana.NewIndexVamana("ovb", vamana.IndexVectorVamanaParameters(d, "euclidean", L, R, 1.2), strict=False)
    g.set_start(sv)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    assert o.insert_rounds(ids, base) == 0
    g.insert_batch(ids, base)
    assert_same_graph(g, o)
    dels = rng.choice(ids, size=n // 6, replace=False)
    g.delete_batch(dels)
    assert o.delete(dels) == 0
    _, _, off, _ = o.export(with_vectors=False)
    assert int(off[1] - off[0]) > 64
    assert_same_graph(g, o)
    new = unit_rows(rng, 300, d)
    new_ids = np.arange(n + 10, n + 310, dtype=np.uint64)
    assert o.insert_rounds(new_ids, new) == 0
    g.insert_batch(new_ids, new)
    assert_same_graph(g, o)
    g.close()


def test_edge_scan_matches_the_reference_rule(oracle):
    """sdb_index_edge_scan (IndexVamana.EdgeScan, node.go:142-199) against the oracle's restatement on the exported
    graph: nodes with an edge into the delete set, and valid nodes nobody points at."""
    from semadb_amd import vamana
    rng = np.random.default_rng(404)
    d, n = 24, 1200
    base = unit_rows(rng, n, d)
    ix = vamana.NewIndexVamana("es", vamana.IndexVectorVamanaParameters(d, "euclidean", 30, 8, 1.2), strict=False)
    ix.set_start(start_vector(rng, d))
    ix.insert_batch(None, base)
    ids, _, off, edges = ix.export(with_vectors=False)
    for trial in range(4):
        dele = set(int(v) for v in rng.choice(np.arange(2, n + 2), size=[1, 17, 200, 900][trial], replace=False))
        dele.add(10 ** 9)  # an id that is not stored changes nothing
        tp, ts = ix.EdgeScan(dele)
        w_tp, w_ts = oracle.edge_scan(ids, off, edges, dele)
        assert sorted(int(v) for v in tp) == w_tp
        assert sorted(int(v) for v in ts) == w_ts
    e_tp, e_ts = ix.EdgeScan(set())
    assert len(e_tp) == 0 and sorted(int(v) for v in e_ts) == oracle.edge_scan(ids, off, edges, set())[1]
    # after a real delete the graph has tombstones: the scan skips them
    ix.delete_batch(np.array(sorted(dele - {10 ** 9})[:50], dtype=np.uint64))
    ids, _, off, edges = ix.export(with_vectors=False)
    dele2 = set(int(v) for v in rng.choice(ids[1:], size=30, replace=False))
    tp, ts = ix.EdgeScan(dele2)
    w_tp, w_ts = oracle.edge_scan(ids, off, edges, dele2)
    assert sorted(int(v) for v in tp) == w_tp and sorted(int(v) for v in ts) == w_ts
    ix.close()


@pytest.mark.parametrize("metric,d", [("cosine", 48), ("euclidean", 33)])
def test_compact_changes_nothing_but_the_rows(oracle, metric, d):
    """sdb_index_compact squeezes the tombstones out (the reference frees deleted nodes at flush, node.go:129-134):
    the exported graph, every search (ids, distance bits, visit order, counters) and everything the write path
    remembers per row are as before -- later inserts and deletes still build the oracle's graph edge for edge."""
    from semadb_amd import vamana
    from tests.helpers import assert_same_graph, build_oracle_index
    rng = np.random.default_rng(d * 3)
    n, R, L = 1800, 16, 40
    base = unit_rows(rng, n + 600, d)
    o = build_oracle_index(oracle, base[:n], metric, R=R, L=L, seed=9)
    ix = vamana.NewIndexVamana("cp", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    ix.set_start(start_vector(np.random.default_rng(9), d))
    ix.insert_batch(np.arange(2, n + 2, dtype=np.uint64), base[:n], round_size=1)
    gone = rng.choice(np.arange(2, n + 2), 500, replace=False).astype(np.uint64)
    ix.delete_batch(gone)
    assert o.delete(gone) == 0
    assert ix.row_usage() == (n + 1, 500)
    q = unit_rows(rng, 24, d)
    before = ix.search_batch(q, 10, L, trace=True, visit_cap=512)
    g0 = ix.export()
    ix.compact()
    assert ix.row_usage() == (n + 1 - 500, 0) and ix.version_diff() == 0
    g1 = ix.export()
    for a, b in zip(g0, g1):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    after = ix.search_batch(q, 10, L, trace=True, visit_cap=512)
    assert np.array_equal(before[0], after[0]) and np.array_equal(bits(before[1]), bits(after[1]))
    assert np.array_equal(before[3].visit_ids, after[3].visit_ids) and np.array_equal(before[3].n_dist, after[3].n_dist)
    assert_same_graph(ix, o)
    assert not ix.exists_batch(gone[:5]).any() and ix.exists(int(sorted(set(range(2, n + 2)) - set(int(g) for g in gone))[0]))
    # the write path carries on from the compacted state exactly as the oracle does from its own
    new_ids = np.concatenate([gone[:100], np.arange(n + 2, n + 502, dtype=np.uint64)])  # freed ids come back (idcounter.go)
    for k, i in enumerate(new_ids):
        assert o.insert(int(i), base[n + k]) == 0
    ix.insert_batch(new_ids, base[n:n + 600], round_size=1)
    assert_same_graph(ix, o)
    gone2 = rng.choice(new_ids, 150, replace=False).astype(np.uint64)
    ix.delete_batch(gone2)
    assert o.delete(gone2) == 0
    assert_same_graph(ix, o)
    ix.compact()
    assert_same_graph(ix, o)
    ix.compact()  # nothing to do
    assert ix.row_usage()[1] == 0
    ix.close()
