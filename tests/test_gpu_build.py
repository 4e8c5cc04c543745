"""K4 parity (GPU): batched insert.  With round_size = 1 the device build is a sequential
insertSinglePoint loop (insert.go:16-68) and must produce the oracle's graph edge for edge; with
larger rounds it is validated the way the reference validates its own (non-deterministic) build:
connectivity from the start node, degree bound, self-retrieval (vamana_test.go:29-46,63-75,230-252)."""
import numpy as np
import pytest

from tests.helpers import bits, build_oracle_index, start_vector, unit_rows

pytestmark = pytest.mark.gpu


def _new_gpu(d, metric, R, L, alpha=1.2, strict=False):
    from semadb_amd import vamana
    return vamana.NewIndexVamana("b", vamana.IndexVectorVamanaParameters(d, metric, L, R, alpha), strict=strict)


def _reachable(ids, offsets, edges):
    pos = {int(v): i for i, v in enumerate(ids)}
    seen, stack = set(), [1]
    while stack:
        v = stack.pop()
        if v in seen:
            continue
        seen.add(v)
        i = pos[v]
        stack.extend(int(e) for e in edges[int(offsets[i]):int(offsets[i + 1])])
    return len(seen) - 1


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,n,R,L", [(2, 250, 8, 25), (33, 300, 16, 30), (96, 400, 32, 50), (128, 500, 64, 75),
                                     (384, 300, 32, 50), (160, 200, 16, 30), (1536, 150, 16, 30), (3072, 100, 8, 25),
                                     (1280, 120, 8, 25)])
def test_sequential_build_identical_to_oracle(oracle, metric, d, n, R, L):
    rng = np.random.default_rng(d + n)
    base = unit_rows(rng, n, d) if d > 2 else rng.random((n, d), dtype=np.float32)
    o = build_oracle_index(oracle, base, metric, R=R, L=L, seed=77)
    sv = start_vector(np.random.default_rng(77), d)
    ix = _new_gpu(d, metric, R, L)
    ix.set_start(sv)
    ix.insert_batch(np.arange(2, n + 2, dtype=np.uint64), base, round_size=1)
    g_ids, g_vecs, g_off, g_edges = ix.export()
    o_ids, o_vecs, o_off, o_edges = o.export()
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(bits(g_vecs), bits(o_vecs))
    assert np.array_equal(g_off, o_off), "degree sequence differs"
    assert np.array_equal(g_edges, o_edges), "edge lists differ"
    ix.close()


def test_small_degree_bound_forces_reprunes(oracle):
    """R = 4: nearly every back-edge overflows and re-prunes (insert.go:47-58)"""
    rng = np.random.default_rng(21)
    base = unit_rows(rng, 400, 16)
    o = build_oracle_index(oracle, base, "euclidean", R=4, L=20, alpha=1.2, seed=5)
    ix = _new_gpu(16, "euclidean", 4, 20)
    ix.set_start(start_vector(np.random.default_rng(5), 16))
    ix.insert_batch(None, base, round_size=1)  # ids assigned 2..n+1
    g = ix.export(with_vectors=False)
    oe = o.export(with_vectors=False)
    assert np.array_equal(g[0], oe[0]) and np.array_equal(g[2], oe[2]) and np.array_equal(g[3], oe[3])
    ix.close()


@pytest.mark.parametrize("round_size", [0, 7, 64])
def test_batched_build_invariants(oracle, round_size):
    rng = np.random.default_rng(31 + round_size)
    n, d = 3000, 64
    base = unit_rows(rng, n, d)
    ix = _new_gpu(d, "cosine", 32, 50)
    ix.set_start(start_vector(rng, d))
    ix.insert_batch(None, base, round_size=round_size)
    ids, vecs, off, edges = ix.export()
    assert len(ids) == n + 1 and _reachable(ids, off, edges) == n          # checkConnectivity
    deg = np.diff(off.astype(np.int64))
    assert deg.max() <= 32 and deg.min() >= 1
    # no self loops, no duplicate edges, no dangling ids
    idset = set(int(v) for v in ids)
    for i in range(len(ids)):
        row = [int(e) for e in edges[int(off[i]):int(off[i + 1])]]
        assert len(set(row)) == len(row) and int(ids[i]) not in row and all(e in idset for e in row)
    # Test_Search vamana_test.go:230-252: every point finds itself first
    g_ids, g_d, g_c, _ = ix.search_batch(base[:256], 10, 50)
    assert np.all(g_c == 10) and np.array_equal(g_ids[:, 0], np.arange(2, 258, dtype=np.uint64))
    # and the search on this graph is still oracle-identical (graph built on device, walked on both)
    o = oracle.Index(d, "cosine", 32, 50, 1.2)
    o.load(ids, vecs, off, edges)
    q = unit_rows(rng, 16, d)
    g_ids, g_d, g_c, tr = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
    for k in range(16):
        o_ids, o_d, o_vis, o_tr = o.search(q[k], 10, 50)
        assert np.array_equal(g_ids[k], o_ids) and np.array_equal(bits(g_d[k]), bits(o_d))
        assert np.array_equal(tr.visit_ids[k, :o_tr.n_hop], o_vis)
    ix.close()


def test_recall_of_batched_build(oracle):
    """recall@10 of the device-built graph against exact kNN (the flat index's test pattern,
    shard/index/flat/flat_test.go:134-191)"""
    rng = np.random.default_rng(11)
    n, d, nq = 5000, 32, 200
    centers = rng.standard_normal((20, d)).astype(np.float32)
    base = (centers[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    q = (centers[rng.integers(0, 20, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    ix = _new_gpu(d, "euclidean", 64, 75, strict=True)
    ix.set_start(start_vector(rng, d))
    ix.insert_batch(None, base)
    g_ids, _, _, _ = ix.search_batch(q, 10, 75)
    dmat = ((q[:, None, :] - base[None, :, :]) ** 2).sum(-1)
    truth = np.argsort(dmat, axis=1)[:, :10] + 2
    hits = sum(len(set(map(int, g_ids[i])) & set(map(int, truth[i]))) for i in range(nq))
    assert hits / (nq * 10) >= 0.95
    ix.close()


def test_insert_id_rules(oracle):
    # Test_InvalidIdInsert vamana_test.go:77-90 ; vamana.go:150-157
    from semadb_amd import vamana, SemaDBError
    ix = _new_gpu(2, "euclidean", 64, 75)
    with pytest.raises(SemaDBError):
        ix.insert_batch(np.array([5], dtype=np.uint64), np.zeros((1, 2), np.float32))  # no start node
    ix.set_start([0.6, 0.8])
    for bad in (0, 1):
        with pytest.raises(SemaDBError):
            ix.InsertUpdateDelete([vamana.IndexVectorChange(bad, [0.5, 0.5])])
    ix.InsertUpdateDelete([vamana.IndexVectorChange(7, [0.5, 0.5]), vamana.IndexVectorChange(9, [0.1, 0.2])])
    with pytest.raises(SemaDBError):
        ix.insert_batch(np.array([7], dtype=np.uint64), np.array([[0.5, 0.5]], np.float32))  # exists: not an insert
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([0.5, 0.5], 75, 10))
    assert [r.NodeId for r in res] == [7, 9] and res[0].Distance == 0
    n_nodes, n_edges, max_id = ix.stats()
    assert n_nodes == 3 and max_id == 9
    ix.close()


@pytest.mark.parametrize("chip_wide", [False, True])
@pytest.mark.parametrize("metric,d,n,m,R", [("euclidean", 32, 700, 40, 16), ("cosine", 96, 900, 300, 32),
                                           ("dot", 100, 2600, 2400, 32), ("cosine", 384, 1500, 1300, 64),
                                           ("euclidean", 8, 1800, 1700, 8)])
def test_union_prune_paths_match_oracle(oracle, chip_wide, metric, d, n, m, R):
    """insert.go:47-58 over a node's neighbours + MANY new candidates at once (the rule a build round applies to
    a target with several requests; the entry node of a big round gets thousands): the one-wavefront kernel and
    the chip-wide kernel sequence (bigprune.inc) both give the oracle's row -- sort order with ties, the
    degree-bound stop, self / duplicate / unknown candidates dropped like candidateSet.Add does."""
    from semadb_amd import vamana
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(n + m)
    base = unit_rows(rng, n, d) if d > 8 else rng.integers(0, 5, size=(n, d)).astype(np.float32)  # d = 8: ties
    o = build_oracle_index(oracle, base, metric, R=R, L=max(25, R))
    ids, vecs, off, edges = o.export()
    g = vamana.NewIndexVamana("u", vamana.IndexVectorVamanaParameters(d, metric, max(25, R), R, 1.2), strict=False)
    g.load(ids, vecs, off, edges)
    for node in (1, int(ids[len(ids) // 2])):  # the start node and an ordinary one
        extra = rng.choice(ids[1:], size=m, replace=False).astype(np.uint64)
        extra = np.concatenate([extra, [node, 10 ** 9], extra[:5]]).astype(np.uint64)  # self, unknown, repeats
        assert o.union_prune(node, extra) == 0
        g.union_prune(node, extra, chip_wide=chip_wide)
        assert_same_graph(g, o)
    g.close()


def test_hub_targets_through_the_build(oracle, monkeypatch):
    """hub_min = 3 sends every target with three or more requests in a round through the chip-wide prune:
    the graph keeps the reference's invariants and the recall of the ordinary build."""
    from semadb_amd import vamana
    rng = np.random.default_rng(77)
    d, n = 48, 6000
    base = unit_rows(rng, n, d)
    g = vamana.NewIndexVamana("hub", vamana.IndexVectorVamanaParameters(d, "cosine", 50, 24, 1.2), strict=False)
    g.set_tuning("hub_min", 3)
    g.set_start(unit_rows(np.random.default_rng(1), 1, d)[0])
    g.insert_batch(None, base, round_size=512)
    ids, vecs, off, edges = g.export()
    deg = np.diff(off.astype(np.int64))
    assert deg.max() <= 24 and len(ids) == n + 1
    idset = set(int(v) for v in ids)
    assert all(int(e) in idset for e in edges)
    for i in range(len(ids)):  # no self loops, no duplicate edges
        row = [int(e) for e in edges[int(off[i]):int(off[i + 1])]]
        assert int(ids[i]) not in row and len(set(row)) == len(row)
    q = base[:200]
    got = g.search_batch(q, 1, 50)[0]
    assert (got[:, 0] == np.arange(2, 202, dtype=np.uint64)).mean() > 0.97  # every point finds itself
    g.close()


@pytest.mark.parametrize("chip_wide", [False, True])
def test_union_prune_on_a_quantized_store(oracle, chip_wide):
    """the grouped rule with DistanceFromPoint = centroid-pair sums (product.go:279-305), both device forms"""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(404)
    d, n, M, K, R = 32, 1500, 8, 16, 24
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, "euclidean", R=R, L=30)
    ids, vecs, off, edges = o.export()
    first = rng.integers(0, 500, M)
    opq = oracle.PQ(d, "euclidean", M, K)
    opq.fit(vecs[1:501].copy(), first, alias=True)
    assert o.attach_pq(opq, np.stack([opq.encode(v) for v in vecs])) == 0
    g = vamana.NewIndexVamana("uq", vamana.IndexVectorVamanaParameters(d, "euclidean", 30, R, 1.2), strict=False)
    g.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(vecs[1:501].copy(), first, alias=True)
    vs.attach(g, gpq)
    for node in (1, int(ids[700])):
        extra = rng.choice(ids[1:], size=1200, replace=False).astype(np.uint64)
        assert o.union_prune(node, extra) == 0
        g.union_prune(node, extra, chip_wide=chip_wide)
        assert_same_graph(g, o)
    g.close()
    gpq.close()


@pytest.mark.parametrize("metric,d,n,R,L", [("cosine", 48, 6000, 24, 40), ("euclidean", 128, 3000, 64, 75),
                                           ("dot", 33, 4000, 12, 30)])
def test_sequential_build_identical_to_oracle_larger(oracle, metric, d, n, R, L):
    """thousands of sequential inserts: rows fill up, so almost every back-edge goes through the re-prune with
    its distance caches, the search-time distance table and the few-new-candidates decision -- and the graph is
    still the oracle's, edge for edge"""
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(d * n)
    lat = rng.standard_normal((10, d)).astype(np.float32)
    base = rng.standard_normal((n, 10)).astype(np.float32) @ lat + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    o = build_oracle_index(oracle, base, metric, R=R, L=L, seed=31)
    ix = _new_gpu(d, metric, R, L)
    ix.set_start(start_vector(np.random.default_rng(31), d))
    ix.insert_batch(np.arange(2, n + 2, dtype=np.uint64), base, round_size=1)
    assert_same_graph(ix, o)
    ix.close()


@pytest.mark.parametrize("metric,d,n,R,L,round_size,big_min", [("cosine", 48, 5000, 24, 40, 0, 512),
                                                              ("euclidean", 32, 4000, 16, 30, 64, 512),
                                                              ("cosine", 64, 6000, 32, 50, 0, 3),
                                                              ("dot", 24, 3000, 8, 25, 200, 2)])
def test_batched_build_identical_to_oracle_schedule(oracle, monkeypatch, metric, d, n, R, L, round_size, big_min):
    """The batched build (rounds of up to 2 % of the graph: the form the bench and bulk loads use) against the
    oracle's restatement of the SAME schedule -- snapshot searches, own prunes, back-edge requests per target in
    insert order, grouped, hubs all at once (oracle/sdb_oracle.c orc_index_insert_round): the graphs are equal
    edge for edge.  big_min 2 / 3 sends nearly every multi-request target through the chip-wide hub prune."""
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(d + n + big_min)
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base = rng.standard_normal((n, 8)).astype(np.float32) @ lat + 0.15 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    sv = start_vector(np.random.default_rng(3), d)
    o = oracle.Index(d, metric, R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    assert o.insert_rounds(ids, base, round_size=round_size, big_min=big_min) == 0
    # both forms of the back-edge re-prunes that cannot be settled from a few rows of pair distances (a node's first
    # prune, many new candidates): handed to the LDS-tiled prune behind k_backedges (default), or pruned in place
    for no_defer in (0, 1):
        ix = _new_gpu(d, metric, R, L)
        ix.set_tuning("hub_min", big_min)
        ix.set_tuning("no_defer", no_defer)
        ix.set_start(sv)
        ix.insert_batch(ids, base, round_size=round_size)
        assert_same_graph(ix, o)
        ix.close()


@pytest.mark.parametrize("metric,d,n,R,L", [("cosine", 16, 60000, 32, 50), ("euclidean", 128, 40000, 64, 75)])
def test_batched_build_at_natural_scale(oracle, monkeypatch, metric, d, n, R, L):
    """The batched build at a size where the schedule's own machinery is in play with its default settings: rounds
    of 800-1200 points, the start node collecting more than 512 back-edge requests per round (the chip-wide hub
    prune, bigprune.inc), groups of requests on ordinary targets -- equal to the oracle's restatement of the
    schedule edge for edge, and the searches that follow walk the same path."""
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(d + n)
    lat = rng.standard_normal((6, d)).astype(np.float32)
    base = rng.standard_normal((n, 6)).astype(np.float32) @ lat + 0.2 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    sv = start_vector(np.random.default_rng(3), d)
    o = oracle.Index(d, metric, R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    assert o.insert_rounds(ids, base) == 0
    ix = _new_gpu(d, metric, R, L)
    ix.set_start(sv)
    ix.insert_batch(ids, base)
    assert_same_graph(ix, o)
    q = base[rng.integers(0, n, 32)] + 0.05 * rng.standard_normal((32, d)).astype(np.float32)
    g_ids, g_d, g_c, tr = ix.search_batch(q.astype(np.float32), 10, L, trace=True, visit_cap=512)
    for i in range(32):
        o_ids, o_d, o_vis, o_tr = o.search(q[i].astype(np.float32), 10, L)
        assert np.array_equal(g_ids[i, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d))
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis)
    ix.close()


@pytest.mark.parametrize("metric,d,n,R,L,searchL", [("cosine", 96, 2500, 32, 50, 50), ("euclidean", 100, 1500, 16, 30, 30),
                                                    ("dot", 768, 600, 16, 120, 120), ("cosine", 1536, 300, 8, 40, 40)])
def test_new_node_prune_tiled_and_untiled_agree(oracle, metric, d, n, R, L, searchL):
    """robustPrune of the new nodes has two device forms: candidate rows staged in an LDS tile by a 256-thread
    workgroup (k_prune_new_tiled, the default) and rows read from global memory by one wavefront (k_prune_new,
    forced with the no_tile knob).  Both give the oracle's graph edge for edge -- sequential inserts and batched
    rounds; d = 100 has a tail chain, d = 768 / 1536 and searchSize 120 make visit lists longer than the tile."""
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(d * 7 + n)
    lat = rng.standard_normal((6, d)).astype(np.float32)
    base = rng.standard_normal((n, 6)).astype(np.float32) @ lat + 0.2 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    sv = start_vector(np.random.default_rng(3), d)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    for round_size in (1, 0):
        o = oracle.Index(d, metric, R, L, 1.2, impl=impl)
        o.set_start(sv)
        if round_size == 1:
            for i in range(n):
                assert o.insert(int(ids[i]), base[i]) == 0
        else:
            assert o.insert_rounds(ids, base, round_size=0) == 0
        for no_tile in (0, 1, 3):  # 3: tiled with the selection loop inside the tiled kernel (no k_prune_select)
            ix = _new_gpu(d, metric, R, L)
            ix.set_tuning("no_tile", no_tile)
            ix.set_start(sv)
            ix.insert_batch(ids, base, round_size=round_size)
            assert_same_graph(ix, o)
            st = ix.build_stats()
            # rows wider than 4 KB leave no room for a useful tile: those dimensions keep the one-wave kernel
            assert (st["staged_rows"] > 0) == (no_tile != 1 and d <= 1024)
            ix.close()
