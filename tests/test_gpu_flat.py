"""IndexFlat.Search on device (shard/index/flat/flat.go:76-132): exact scan, bit-identical distances, the
reference's `dist >= tail -> skip` rule in storage order."""
import numpy as np
import pytest

from tests.helpers import bits, unit_rows

pytestmark = pytest.mark.gpu


def _expected(oracle, q, base, ids, metric, k, allowed=None):
    d = oracle.distance_matrix(q, base, metric, oracle.IMPL_ASM)
    out = []
    for i in range(q.shape[0]):
        idx = np.arange(base.shape[0])
        if allowed is not None:
            idx = np.array([j for j in idx if int(ids[j]) in allowed[i]], dtype=np.int64)
        order = idx[np.argsort(d[i, idx], kind="stable")][:k]  # first seen stays among equals
        out.append((ids[order], d[i, order]))
    return out


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,n", [(2, 300), (33, 500), (128, 1000), (384, 700)])
def test_flat_exact_scan(oracle, metric, d, n):
    from semadb_amd import flat
    rng = np.random.default_rng(d + n)
    base = unit_rows(rng, n, d) if d > 2 else rng.integers(0, 6, size=(n, d)).astype(np.float32)  # d=2: ties
    ids = np.arange(5, n + 5, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric))
    ix.InsertUpdateDelete([flat.IndexVectorChange(int(ids[i]), base[i]) for i in range(n)])
    q = unit_rows(rng, 9, d) if d > 2 else rng.integers(0, 6, size=(9, d)).astype(np.float32)
    for k in (1, 10, 75):
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, k)):
            assert int(g_c[i]) == len(e_ids)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))
    # filter: only ids in the bitmap are scanned (flat.go:100)
    allowed = [set(int(v) for v in rng.choice(ids, size=40, replace=False)) | {10 ** 9} for _ in range(9)]
    g_ids, g_d, g_c = ix.search_batch(q, 10, filters=allowed)
    for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, 10, allowed)):
        assert np.array_equal(g_ids[i, :len(e_ids)], e_ids) and np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))
    ix.close()


def test_flat_reference_kats():
    # shard/index/search_test.go:89-144 on the flat property: data (ii, ii+1), query (42,43) -> first is 42;
    # flat.go:114 HybridScore = -1 * weight * dist
    from semadb_amd import flat
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(2, "euclidean"))
    ix.InsertUpdateDelete([flat.IndexVectorChange(ii, [ii, ii + 1]) for ii in range(2, 102)])
    rset, res = ix.Search(flat.SearchVectorFlatOptions([42, 43], 10))
    assert len(res) == 10 and res[0].NodeId == 42 and res[0].Distance == 0
    rset, res = ix.Search(flat.SearchVectorFlatOptions([42, 43], 5, Weight=0.5))
    assert rset == {40, 41, 42, 43, 44}
    assert all(r.HybridScore + r.HybridScore == -r.Distance for r in res)
    rset, res = ix.Search(flat.SearchVectorFlatOptions([42, 43], 10), filter={47})
    assert len(res) == 1 and res[0].Distance == np.float32(50)
    ix.close()


def test_flat_is_the_ground_truth_of_the_graph_index(oracle):
    """Test_Recall pattern of shard/index/flat/flat_test.go:134-191: the graph index against the exact scan of
    the same store (the start node never shows up in a flat result)."""
    from semadb_amd import flat, vamana
    from tests.helpers import start_vector
    rng = np.random.default_rng(3)
    lat = rng.standard_normal((8, 64)).astype(np.float32)
    rows = lambda m: (lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))(
        rng.standard_normal((m, 8)).astype(np.float32) @ lat)
    base, q = rows(4000), rows(100)
    ix = vamana.NewIndexVamana("g", vamana.IndexVectorVamanaParameters(64, "cosine", 75, 64, 1.2))
    ix.set_start(start_vector(rng, 64))
    ix.insert_batch(None, base)
    t_ids, t_d, t_c = flat.flat_search_batch(ix._h, 64, q, 10)
    assert np.all(t_c == 10) and not np.any(t_ids == 1)
    g_ids, _, _, _ = ix.search_batch(q, 10, 75)
    hits = sum(len(set(map(int, g_ids[i])) & set(map(int, t_ids[i]))) for i in range(100))
    assert hits / 1000 >= 0.95
    ix.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
def test_flat_over_quantized_store(oracle, metric):
    """flat.go:86 binds vecStore.DistanceFromFloat: over a fitted product quantizer that is the LUT distance
    (product.go:250-277), same scan and same `dist >= tail -> skip` rule"""
    from semadb_amd import flat, vectorstore as vs
    d, M, K, n = 32, 8, 16, 900
    rng = np.random.default_rng(17)
    base = unit_rows(rng, n, d)
    ids = np.arange(5, n + 5, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric))
    ix.InsertUpdateDelete([flat.IndexVectorChange(int(ids[i]), base[i]) for i in range(n)])
    first = rng.integers(0, 400, M)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(base[:400].copy(), first, alias=True)
    vs.attach(ix, gpq)
    opq = oracle.PQ(d, metric, M, K)
    opq.fit(base[:400].copy(), first, alias=True)
    codes = np.stack([opq.encode(v) for v in base])
    q = unit_rows(rng, 7, d)
    allowed = [set(int(v) for v in rng.choice(ids, size=60, replace=False)) for _ in range(7)]
    for filt in (None, allowed):
        g_ids, g_d, g_c = ix.search_batch(q, 10, filters=filt)
        for i in range(7):
            lut = opq.lut(q[i])
            dist = np.array([opq.dist_lut(lut, codes[j]) for j in range(n)], dtype=np.float32)
            idx = np.arange(n) if filt is None else np.array([j for j in range(n) if int(ids[j]) in filt[i]])
            order = idx[np.argsort(dist[idx], kind="stable")][:10]
            assert int(g_c[i]) == len(order)
            assert np.array_equal(g_ids[i, :len(order)], ids[order])
            assert np.array_equal(bits(g_d[i, :len(order)]), bits(dist[order]))
    ix.close()
    gpq.close()


@pytest.mark.parametrize("metric,d,n,nq", [("cosine", 64, 40000, 37), ("euclidean", 384, 33000, 19), ("dot", 224, 50000, 130)])
def test_streaming_scan_equals_the_oracle(oracle, metric, d, n, nq):
    """Tables of 32 768 rows and more take the streaming scan (k_flat_scan: every row read once, lane-per-row packed
    FMA chains, thresholded candidates, k_flat_merge): same ids, same distance bits as the oracle's exact scan in
    storage order, for limits 1 / 10 / 128 and a query count that fills no block of 16 evenly."""
    from semadb_amd import flat
    rng = np.random.default_rng(d + n)
    lat = rng.standard_normal((12, d)).astype(np.float32)
    base = rng.standard_normal((n, 12)).astype(np.float32) @ lat + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    if metric != "dot":
        base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    q = base[rng.choice(n, nq, replace=False)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    ids = np.arange(7, n + 7, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
    ix.set_vectors(ids, base)
    for k in (1, 10, 128):
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, k)):
            assert int(g_c[i]) == len(e_ids)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))
    ix.close()


def test_streaming_scan_with_a_sea_of_ties(oracle):
    """small-integer data: thousands of rows at exactly the same distance as the k-th best.  The candidate lists
    overflow (every tie passes `not above the threshold`) and the call starts over on the block path: the answer is
    still the storage-order one, first seen stays (flat.go:104)."""
    from semadb_amd import flat
    rng = np.random.default_rng(5)
    n, d = 70000, 32
    base = rng.integers(0, 2, size=(n, d)).astype(np.float32)
    q = rng.integers(0, 2, size=(9, d)).astype(np.float32)
    ids = np.arange(2, n + 2, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "euclidean"), capacity=n + 1)
    ix.set_vectors(ids, base)
    for k in (10, 100):
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, "euclidean", k)):
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))
    ix.close()


def test_streaming_scan_skips_the_start_node_and_tombstones(oracle):
    """on a graph index the exact scan leaves out the start node (not a point) and deleted rows"""
    from semadb_amd import flat, vamana
    from tests.helpers import start_vector
    rng = np.random.default_rng(8)
    n, d = 36000, 32
    base = unit_rows(rng, n, d)
    ix = vamana.NewIndexVamana("fs", vamana.IndexVectorVamanaParameters(d, "cosine", 30, 8, 1.2), strict=False)
    ix.set_start(start_vector(rng, d))
    ix.insert_batch(None, base)
    q = base[:24] + 0.01
    gone = np.arange(2, 26, dtype=np.uint64)  # the queries' own rows: their exact nearest neighbours
    ix.delete_batch(gone)
    keep = np.ones(n, dtype=bool)
    keep[:24] = False
    ids = np.arange(2, n + 2, dtype=np.uint64)
    g_ids, g_d, g_c = flat.flat_search_batch(ix._h, d, q, 10)
    for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base[keep], ids[keep], "cosine", 10)):
        assert np.array_equal(g_ids[i], e_ids) and np.array_equal(bits(g_d[i]), bits(e_d))
    ix.close()
