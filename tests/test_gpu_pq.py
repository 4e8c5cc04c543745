"""K5-K8 parity (GPU): k-means, product quantizer fit/encode/LUT/symmetric distance and the PQ-backed
greedy search are bit-identical to the oracle's restatement of utils/kmeans.go and
shard/vectorstore/product.go."""
import numpy as np
import pytest

from tests.helpers import assert_same_graph, bits, build_oracle_index, unit_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,stride,offset,length,K", [(6, 4, 0, 2, 3), (6, 4, 2, 2, 3), (500, 16, 4, 8, 16),
                                                      (1000, 96, 32, 48, 32), (300, 8, 0, 4, 256), (5, 4, 2, 2, 256)])
@pytest.mark.parametrize("alias", [True, False])
def test_kmeans_matches_oracle(oracle, n, stride, offset, length, K, alias):
    from semadb_amd.kmeans import KMeans
    rng = np.random.default_rng(n + K)
    X = rng.standard_normal((n, stride)).astype(np.float32)
    first = int(rng.integers(0, n))
    xo = X.copy()
    o_cent, o_lab, o_it = oracle.kmeans_fit(xo, offset, length, K, max_iter=25, first_idx=first, alias=alias)
    xg = X.copy()
    km = KMeans(K, 25, offset, length, first_idx=first, alias=alias).Fit(xg)
    assert np.array_equal(km.Labels, o_lab)
    assert np.array_equal(bits(km.Centroids), bits(o_cent))
    assert km.iters == o_it
    assert np.array_equal(bits(xg), bits(xo))  # aliasing writes through to the caller's rows (or not at all)


def test_kmeans_reference_kat():
    # TestKMeans_Fit utils/kmeans_test.go:15-68
    from semadb_amd.kmeans import KMeans
    rng = np.random.default_rng(3)
    offs = np.array([[-1, -1, 1, 1], [-1, -1, 1, 1], [0, 0, -1, 1], [0, 0, -1, 1], [1, 1, 1, -1], [1, 1, 1, -1]],
                    dtype=np.float32)
    for off in (0, 2):
        data = (offs * 10 + rng.random(offs.shape, dtype=np.float32)).astype(np.float32)
        km = KMeans(3, 10, off, 2, first_idx=1).Fit(data)
        lb = km.Labels
        assert km.Centroids.shape == (3, 2)
        assert lb[0] == lb[1] and lb[0] != lb[2] and lb[2] == lb[3] and lb[2] != lb[4] and lb[4] == lb[5]


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,M,K,n", [(4, 2, 256, 5), (32, 8, 16, 600), (96, 2, 32, 400), (768, 8, 64, 700),
                                     (64, 16, 256, 1200),
                                     # the short sub-vector lengths with kernels of their own (4 above; 8, 16, 24) and one without (12)
                                     (64, 8, 32, 500), (128, 8, 64, 500), (96, 4, 16, 300), (96, 8, 16, 300),
                                     # sub-vector lengths 100 (3 blocks + tail), 256 (8 blocks), 320 (generic kernel)
                                     (200, 2, 16, 300), (512, 2, 8, 200), (640, 2, 8, 200),
                                     # 4 .. 7 blocks, whole and with a tail: the scalar operands of these go block by block
                                     # through hand-placed s_load_dwordx16 (pq.hip dist_regs, round 6)
                                     (256, 2, 8, 200), (320, 2, 12, 200), (384, 2, 8, 150), (448, 2, 8, 150), (460, 2, 8, 150),
                                     (270, 2, 8, 150),
                                     # whole 32-float blocks and K % 16 == 0: the LUT of dot / cosine is built on the matrix cores
                                     (256, 8, 256, 1300), (512, 2, 16, 300), (768, 8, 256, 1300)])
def test_pq_matches_oracle(oracle, metric, d, M, K, n):
    from semadb_amd import vectorstore as vs
    rng = np.random.default_rng(d + M + K)
    X = rng.standard_normal((n, d)).astype(np.float32)
    first = rng.integers(0, n, M)
    xo, xg = X.copy(), X.copy()
    opq = oracle.PQ(d, metric, M, K)
    o_codes = opq.fit(xo, first, alias=True)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    g_codes = gpq.Fit(xg, first, alias=True)
    assert np.array_equal(g_codes, o_codes)
    fc, cd = gpq.codebook()
    assert np.array_equal(bits(fc), bits(opq.flat_centroids))
    assert np.array_equal(bits(cd), bits(opq.centroid_dists))
    assert np.array_equal(bits(xg), bits(xo))
    # encode (product.go:136-159)
    V = rng.standard_normal((50, d)).astype(np.float32)
    ge = gpq.encode(V)
    assert np.array_equal(ge, np.stack([opq.encode(v) for v in V]))
    # asymmetric LUT distance (product.go:250-277)
    Q = rng.standard_normal((7, d)).astype(np.float32)
    got = gpq.lut_distance(Q, ge)
    want = np.array([[opq.dist_lut(opq.lut(q), c) for c in ge] for q in Q], dtype=np.float32)
    assert np.array_equal(bits(got), bits(want))
    # symmetric distance (product.go:293-304)
    perm = rng.permutation(50)
    gs = gpq.sym_distance(ge, ge[perm])
    ws = np.array([opq.dist_sym(ge[i], ge[perm[i]]) for i in range(50)], dtype=np.float32)
    assert np.array_equal(bits(gs), bits(ws))
    gpq.close()


def test_pq_reference_contract():
    # Test_DistanceFromFloat / Test_DistanceFromPoint, product store, fit=true
    # (shard/vectorstore/vectorestore_test.go:17,37-50,112-154)
    from semadb_amd import vectorstore as vs
    X = np.array([[1, 2, 3, 4], [4, 5, 6, 7], [7, 8, 9, 10], [-10, -11, -12, -13], [-13, 14, -15, 16]], dtype=np.float32)
    pq = vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(256, 2, 5), 4)
    pq.Fit(X.copy(), [0, 0])
    c = pq.encode(np.array([[1, 2, 3, 4], [4, 5, 6, 7]], dtype=np.float32))
    d = pq.lut_distance(np.array([[1, 2, 3, 4]], dtype=np.float32), c)
    assert d[0, 0] == 0 and d[0, 0] < d[0, 1]
    s = pq.sym_distance(c[[0, 0]], c[[0, 1]])
    assert s[0] == 0 and s[0] < s[1]
    pq.close()


def test_pq_parameter_errors():
    # product.go:44-46,63-65
    from semadb_amd import vectorstore as vs, SemaDBError
    with pytest.raises(SemaDBError):
        vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(16, 3), 10)
    with pytest.raises(SemaDBError):
        vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(257, 2), 8)
    with pytest.raises(SemaDBError):
        vs.ProductQuantizer("hamming", vs.ProductQuantizerParameters(16, 2), 8)
    pq = vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(16, 2), 8)
    with pytest.raises(SemaDBError):  # unfitted: encode returns nil in the reference (product.go:137-139)
        pq.encode(np.zeros((1, 8), np.float32))
    pq.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,M,K", [(32, 8, 16), (96, 8, 256), (64, 32, 64), (128, 64, 32)])  # (the last: code rows gathered by slot, M > 32)
def test_pq_search_parity(oracle, metric, d, M, K):
    """greedy search over a quantized store: LUT distances (product.go:250-277) drive the same walk"""
    from semadb_amd import vamana, vectorstore as vs
    rng = np.random.default_rng(d * M + K)
    n = 1500
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, metric, R=32, L=50)
    ids, vecs, off, edges = o.export()
    train = vecs[1:701].copy()
    first = rng.integers(0, 700, M)
    opq = oracle.PQ(d, metric, M, K)
    opq.fit(train.copy(), first, alias=True)
    codes = np.stack([opq.encode(v) for v in vecs])
    assert o.attach_pq(opq, codes) == 0
    ix = vamana.NewIndexVamana("pq", vamana.IndexVectorVamanaParameters(d, metric, 50, 32, 1.2), strict=False)
    ix.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(train.copy(), first, alias=True)
    vs.attach(ix, gpq)
    q = unit_rows(rng, 32, d)
    g_ids, g_d, g_c, tr = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
    for k in range(32):
        o_ids, o_d, o_vis, o_tr = o.search(q[k], 10, 50)
        assert int(g_c[k]) == len(o_ids)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids)
        assert np.array_equal(bits(g_d[k, :len(o_ids)]), bits(o_d))
        assert int(tr.n_hop[k]) == o_tr.n_hop and int(tr.n_dist[k]) == o_tr.n_dist
        assert np.array_equal(tr.visit_ids[k, :o_tr.n_hop], o_vis)
    # repeated batches walk the same path (workspaces, LUT buffers and the hash table are reused)
    for rep in range(2):
        g2 = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
        assert np.array_equal(g2[0], g_ids) and np.array_equal(bits(g2[1]), bits(g_d))
        assert np.array_equal(g2[3].visit_ids, tr.visit_ids)
    # the default is the two-wave kernel (walker + merger, search_kernel.h k_greedy_search_pq2); the one-wave kernel
    # (SDB_TUNE_PQ_NARROW = 1), also with the other visited sets, walks the same path
    for key in ("pq_narrow", "wide_hash"):
        ix.set_tuning(key, 1)
        g2 = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
        ix.set_tuning(key, 0)
        assert np.array_equal(g2[0], g_ids) and np.array_equal(bits(g2[1]), bits(g_d)), key
        assert np.array_equal(g2[3].visit_ids, tr.visit_ids) and np.array_equal(g2[3].n_dist, tr.n_dist), key
        assert np.array_equal(g2[3].n_edges, tr.n_edges) and np.array_equal(g2[3].n_hop, tr.n_hop), key
    # queries that make the table non-finite (a NaN component: every sum NaN; components that overflow: +-inf and NaN
    # sums side by side) -- the walker names nothing then and the merger's insertions decide every hop (distset.go:184-198
    # with NaN: `d > tail` is false, `d < items[i-1]` is false)
    from tests.test_gpu_nonfinite import same_bits_or_both_nan  # (a NaN's payload is the one thing the machines do not share)
    qn = q[:8].copy()
    qn[0, 3] = np.nan
    qn[1, :] = np.float32(3e38)
    qn[2, ::2] = np.float32(-3e38)
    qn[3, 0] = np.inf
    qn[4, : d // 2] = np.float32(2e38)
    with np.errstate(all="ignore"):
        n_ids, n_d, n_c, n_tr = ix.search_batch(qn, 10, 50, trace=True, visit_cap=512)
        for k in range(8):
            o_ids, o_d, o_vis, o_tr = o.search(qn[k], 10, 50)
            assert int(n_c[k]) == len(o_ids), k
            assert np.array_equal(n_ids[k, :len(o_ids)], o_ids) and same_bits_or_both_nan(n_d[k, :len(o_ids)], o_d), k
            assert int(n_tr.n_hop[k]) == o_tr.n_hop and int(n_tr.n_dist[k]) == o_tr.n_dist, k
            assert np.array_equal(n_tr.visit_ids[k, :o_tr.n_hop], o_vis), k
        ix.set_tuning("pq_narrow", 1)
        n2 = ix.search_batch(qn, 10, 50, trace=True, visit_cap=512)
        ix.set_tuning("pq_narrow", 0)
        assert np.array_equal(n2[0], n_ids) and same_bits_or_both_nan(n2[1], n_d)
        assert np.array_equal(n2[3].visit_ids, n_tr.visit_ids)
    # filtered search over the quantized store: seeds and result set use the LUT distance too
    filters = [set(int(v) for v in rng.choice(ids[1:], size=40, replace=False)) for _ in range(32)]
    f_ids, f_d, f_c, f_tr = ix.search_batch(q, 5, 50, filters=filters, trace=True, visit_cap=512)
    for k in range(32):
        o_ids, o_d, o_vis, o_tr = o.search(q[k], 5, 50, filter_ids=sorted(filters[k]))
        assert int(f_c[k]) == len(o_ids)
        assert np.array_equal(f_ids[k, :len(o_ids)], o_ids) and np.array_equal(bits(f_d[k, :len(o_ids)]), bits(o_d))
        assert np.array_equal(f_tr.visit_ids[k, :o_tr.n_hop], o_vis)
    ix.close()
    gpq.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,M,K", [(32, 8, 16), (96, 8, 256)])
def test_pq_insert_delete_parity(oracle, metric, d, M, K):
    """a fitted quantizer encodes on Set (product.go:161-169); the insert's search then runs on LUT distances
    and its prunes on the centroid-pair table (product.go:279-305), and so does the delete path"""
    from semadb_amd import vamana, vectorstore as vs
    rng = np.random.default_rng(7 * d + M + K)
    n0, n1 = 600, 250
    base = unit_rows(rng, n0 + n1, d)
    o = build_oracle_index(oracle, base[:n0], metric, R=16, L=30)
    ids, vecs, off, edges = o.export()
    train = vecs[1:401].copy()
    first = rng.integers(0, 400, M)
    opq = oracle.PQ(d, metric, M, K)
    opq.fit(train.copy(), first, alias=True)
    assert o.attach_pq(opq, np.stack([opq.encode(v) for v in vecs])) == 0
    ix = vamana.NewIndexVamana("pq", vamana.IndexVectorVamanaParameters(d, metric, 30, 16, 1.2), strict=False)
    ix.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(train.copy(), first, alias=True)
    vs.attach(ix, gpq)
    new_ids = np.arange(n0 + 2, n0 + 2 + n1, dtype=np.uint64)
    for i in range(n1):
        assert o.insert(int(new_ids[i]), base[n0 + i]) == 0
    ix.insert_batch(new_ids, base[n0:], round_size=1)  # one point per round = sequential insertSinglePoint
    assert_same_graph(ix, o)
    # searches over the grown quantized store agree too (the new rows' codes are the reference's encode())
    q = unit_rows(rng, 16, d)
    g_ids, g_d, g_c = ix.search_batch(q, 10, 30)[:3]
    for k in range(16):
        o_ids, o_d, _, _ = o.search(q[k], 10, 30)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[k, :len(o_ids)]), bits(o_d))
    # delete a tenth of the points: pruneDeleteNeighbour binds DistanceFromPoint (prune.go:58)
    dead = rng.choice(np.arange(2, n0 + 2 + n1, dtype=np.uint64), 80, replace=False)
    assert o.delete(dead) == 0
    ix.delete_batch(dead)
    assert_same_graph(ix, o)
    ix.close()
    gpq.close()


def test_fit_trigger_matches_oracle(oracle):
    """vecStore.Fit after a write (vamana.go:257-260): the index crosses TriggerThreshold, k-means runs over every
    stored vector in storage order, the labels become the centroid ids (product.go:216-218) and later inserts and
    searches run on table distances -- the oracle given the same first-centroid draws ends in the same graph."""
    from semadb_amd import vamana, vectorstore as vs
    d, M, K, thr = 16, 4, 16, 1000
    rng = np.random.default_rng(99)
    base = unit_rows(rng, 1400, d)
    sv = unit_rows(np.random.default_rng(5), 1, d)[0]
    q = vs.Quantizer(vs.QuantizerProduct, vs.ProductQuantizerParameters(K, M, thr))
    ix = vamana.NewIndexVamana("fit", vamana.IndexVectorVamanaParameters(d, "euclidean", 30, 16, 1.2, q), strict=False,
                               fit_seed=7)
    ix.set_start(sv)
    o = oracle.Index(d, "euclidean", 16, 30, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    # 600 points: below the threshold, nothing happens
    ch = [vamana.IndexVectorChange(i + 2, base[i]) for i in range(600)]
    ix.InsertUpdateDelete(ch, round_size=1)
    assert ix._pq is None
    for i in range(600):
        assert o.insert(i + 2, base[i]) == 0
    # 500 more: 1 101 points with the start node -> Fit
    ch = [vamana.IndexVectorChange(i + 2, base[i]) for i in range(600, 1100)]
    ix.InsertUpdateDelete(ch, round_size=1)
    assert ix._pq is not None
    for i in range(600, 1100):
        assert o.insert(i + 2, base[i]) == 0
    ids, vecs, _, _ = o.export()
    opq = oracle.PQ(d, "euclidean", M, K)
    o_codes = opq.fit(vecs.copy(), ix.last_fit_first_idx, alias=True)
    assert o.attach_pq(opq, o_codes) == 0
    assert np.array_equal(vs.get_codes(ix, ids), o_codes)
    fc, cd = ix._pq.codebook()
    assert np.array_equal(bits(fc.ravel()), bits(np.asarray(opq.flat_centroids).ravel()))
    # the quantized index keeps taking writes
    ch = [vamana.IndexVectorChange(i + 2, base[i]) for i in range(1100, 1400)] + [vamana.IndexVectorChange(10, None)]
    ix.InsertUpdateDelete(ch, round_size=1)
    for i in range(1100, 1400):
        assert o.insert(i + 2, base[i]) == 0
    assert o.delete(np.array([10], dtype=np.uint64)) == 0
    assert_same_graph(ix, o)
    qs = unit_rows(rng, 8, d)
    g_ids, g_d, g_c = ix.search_batch(qs, 5, 30)[:3]
    for k in range(8):
        o_ids, o_d, _, _ = o.search(qs[k], 5, 30)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[k, :len(o_ids)]), bits(o_d))
    ix.close()


def test_batched_insert_into_quantized_store_matches_oracle_schedule(oracle, monkeypatch):
    """rounds of inserts into a store with a fitted quantizer (LUT searches, centroid-pair prunes, hub path on)
    against the oracle's restatement of the same round schedule: equal graphs, equal codes"""
    from semadb_amd import vamana, vectorstore as vs
    rng = np.random.default_rng(808)
    d, M, K, R, L = 32, 8, 16, 16, 30
    base = unit_rows(rng, 4000, d)
    o = build_oracle_index(oracle, base[:1200], "euclidean", R=R, L=L)
    ids, vecs, off, edges = o.export()
    first = rng.integers(0, 800, M)
    opq = oracle.PQ(d, "euclidean", M, K)
    opq.fit(vecs[1:801].copy(), first, alias=True)
    assert o.attach_pq(opq, np.stack([opq.encode(v) for v in vecs])) == 0
    g = vamana.NewIndexVamana("bq", vamana.IndexVectorVamanaParameters(d, "euclidean", L, R, 1.2), strict=False)
    g.set_tuning("hub_min", 4)
    g.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer("euclidean", vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(vecs[1:801].copy(), first, alias=True)
    vs.attach(g, gpq)
    new_ids = np.arange(1202, 1202 + 2800, dtype=np.uint64)
    assert o.insert_rounds(new_ids, base[1200:], round_size=0, big_min=4) == 0
    g.insert_batch(new_ids, base[1200:], round_size=0)
    assert_same_graph(g, o)
    g.close()
    gpq.close()


@pytest.mark.parametrize("metric,M,K", [("euclidean", 2, 8), ("cosine", 4, 4)])
def test_quantizer_attached_to_a_device_built_graph(oracle, monkeypatch, metric, M, K):
    """The graph is built on the device with full-precision distances, THEN the store switches to a (coarse)
    quantizer: whatever the write path remembers per row from its earlier prunes -- which leading edges came out
    of a robustPrune, their cached distances -- was made with the old distance function and must not be used by
    the inserts that follow (found by tools/fuzz_parity.py, seed 1 trial 88: the chip-wide prune of a target with
    several requests read the row's cached full-precision distances).  Sequential inserts first, then batched
    rounds with the hub threshold at 2."""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import start_vector
    d, n0, n1, R, L = 16, 700, 200, 8, 24
    rng = np.random.default_rng(M * 100 + K)
    base = unit_rows(rng, n0 + n1, d)
    sv = start_vector(np.random.default_rng(11), d)
    o = oracle.Index(d, metric, R, L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
    o.set_start(sv)
    ix = vamana.NewIndexVamana("dq", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    ix.set_tuning("hub_min", 2)
    ix.set_start(sv)
    ids = np.arange(2, n0 + 2, dtype=np.uint64)
    for i in range(n0):
        assert o.insert(int(ids[i]), base[i]) == 0
    ix.insert_batch(ids, base[:n0], round_size=1)
    assert_same_graph(ix, o)
    o_ids, vecs, _, _ = o.export()
    first = rng.integers(0, len(o_ids), M)
    opq = oracle.PQ(d, metric, M, K)
    codes = opq.fit(vecs.copy(), first, alias=True)
    assert o.attach_pq(opq, codes) == 0
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    assert np.array_equal(gpq.Fit(vecs.copy(), first, alias=True), codes)
    vs.attach(ix, gpq, o_ids, codes)
    new_ids = np.arange(n0 + 2, n0 + 2 + n1, dtype=np.uint64)
    h = n1 // 4
    for i in range(h):
        assert o.insert(int(new_ids[i]), base[n0 + i]) == 0
    ix.insert_batch(new_ids[:h], base[n0:n0 + h], round_size=1)
    assert_same_graph(ix, o)
    assert o.insert_rounds(new_ids[h:], base[n0 + h:], big_min=2) == 0
    ix.insert_batch(new_ids[h:], base[n0 + h:])
    assert_same_graph(ix, o)
    ix.close()
    gpq.close()


def test_quantized_walk_is_the_same_under_every_visited_set(oracle):
    """The walk over a quantized store keeps its visited ids in 16-bit LDS cells (HashVisited16: bucket + remainder
    of a 24-bit bijection, six walks per CU), or -- by tuning -- in 32-bit cells, or in the HBM bitset, or spills
    from the cells to the bitset after a handful of ids.  All of them are exact sets: a graph of 120 000 nodes
    walked with searchSize 96 (spilling early and mid-walk through the limit knob) gives the same ids,
    distance bits, hop and distance counts under each, and the first queries agree with the oracle."""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import start_vector
    rng = np.random.default_rng(31)
    n, d, M, K, R, L = 120000, 32, 8, 256, 48, 96
    lat = rng.standard_normal((10, d)).astype(np.float32)
    base = rng.standard_normal((n, 10)).astype(np.float32) @ lat + 0.2 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    ix = vamana.NewIndexVamana("vs", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(start_vector(np.random.default_rng(3), d))
    ix.insert_batch(None, base)
    ids, vecs, off, edges = ix.export()
    first = rng.integers(0, 5000, M)
    gpq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(K, M), d)
    codes = gpq.Fit(vecs[1:5001].copy(), first, alias=True)
    vs.attach(ix, gpq)
    q = base[rng.choice(n, 96, replace=False)] + 0.05 * rng.standard_normal((96, d)).astype(np.float32)
    ref = ix.search_batch(q, 10, L, trace=True, visit_cap=1024)
    assert int(ref[3].n_dist.max()) > 1200, int(ref[3].n_dist.max())
    # hash16_probes 1 / 2: a key whose first (two) bucket(s) are full sends the walk to the bitset -- the `stuck` spill,
    # a one-in-ten-million event with the full budget of 15 buckets
    for key, value in [("wide_hash", 1), ("no_hash", 1), ("hash_limit", 40), ("hash_limit", 600), ("hash16_probes", 1),
                       ("hash16_probes", 2), ("pq_narrow", 1)]:
        ix.set_tuning("wide_hash", 0), ix.set_tuning("no_hash", 0), ix.set_tuning("hash_limit", 0)
        ix.set_tuning("hash16_probes", 0), ix.set_tuning("pq_narrow", 0)
        ix.set_tuning(key, value)
        got = ix.search_batch(q, 10, L, trace=True, visit_cap=1024)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])), key
        assert np.array_equal(got[3].n_dist, ref[3].n_dist) and np.array_equal(got[3].n_hop, ref[3].n_hop), key
        assert np.array_equal(got[3].visit_ids, ref[3].visit_ids), key
    # and the walk is the reference's: the oracle over the exported graph with the same codes
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=oracle.IMPL_ASM)
    o.load(ids, vecs, off, edges)
    opq = oracle.PQ(d, "cosine", M, K)
    ocodes = opq.fit(vecs[1:5001].copy(), first, alias=True)
    assert np.array_equal(ocodes, codes)
    all_codes = np.stack([opq.encode(v) for v in vecs])
    assert o.attach_pq(opq, all_codes) == 0
    for k in range(6):
        o_ids, o_d, o_vis, o_tr = o.search(q[k], 10, L)
        assert np.array_equal(ref[0][k, :len(o_ids)], o_ids) and np.array_equal(bits(ref[1][k, :len(o_ids)]), bits(o_d))
        assert int(ref[3].n_dist[k]) == o_tr.n_dist
    ix.close()
    gpq.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,M,K", [(128, 128, 64), (192, 192, 256), (256, 256, 32), (384, 384, 256), (768, 192, 256)])
def test_multi_wave_quantized_walk_parity(oracle, metric, d, M, K):
    """Round 3: a quantizer whose per-query table does not fit beside a one-wave walk (M = 128 .. 384) is searched by
    four waves per query -- tables in LDS and in registers, the waves adding their index ranges in turn
    (search_kernel.h PQWideDist).  Ids, LUT-distance bits, visit order and counters equal the oracle's, equal the one-wave
    kernel's (table in global memory, SDB_TUNE_PQ_NARROW), and stay equal when the visited set spills or is a bitset."""
    from semadb_amd import vamana, vectorstore as vs
    rng = np.random.default_rng(d + M + K)
    # the oracle's k-means and encoder are one distance call per (point, centroid, sub-vector): the large quantizers train
    # on 400 rows (more centroids than points per cluster: empty clusters on the way) and encode 1 500
    big = M * K >= 192 * 256
    n, nt = (1500, 400) if big else (2500, 900)
    lat = rng.standard_normal((12, d)).astype(np.float32)
    base = rng.standard_normal((n, 12)).astype(np.float32) @ lat + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    o = build_oracle_index(oracle, base, metric, R=32, L=50)
    ids, vecs, off, edges = o.export()
    train = vecs[1:nt + 1].copy()
    first = rng.integers(0, nt, M)
    opq = oracle.PQ(d, metric, M, K)
    opq.fit(train.copy(), first, alias=True)
    codes = np.stack([opq.encode(v) for v in vecs])
    assert o.attach_pq(opq, codes) == 0
    ix = vamana.NewIndexVamana("pqw", vamana.IndexVectorVamanaParameters(d, metric, 50, 32, 1.2), strict=False)
    ix.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(train.copy(), first, alias=True)
    vs.attach(ix, gpq)
    q = unit_rows(rng, 48, d)
    g_ids, g_d, g_c, tr = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
    for k in range(48):
        o_ids, o_d, o_vis, o_tr = o.search(q[k], 10, 50)
        assert int(g_c[k]) == len(o_ids)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids), k
        assert np.array_equal(bits(g_d[k, :len(o_ids)]), bits(o_d)), k
        assert int(tr.n_hop[k]) == o_tr.n_hop and int(tr.n_dist[k]) == o_tr.n_dist, k
        assert np.array_equal(tr.visit_ids[k, :o_tr.n_hop], o_vis), k
    # (pq_narrow 3: the walker keeps the candidate array itself; the default for M = 192 is the split form, the array in a helper wave)
    for key, value in [("pq_narrow", 1), ("pq_narrow", 2), ("pq_narrow", 3), ("hash_limit", 30), ("wide_hash", 1), ("no_hash", 1)]:
        for kk in ("pq_narrow", "hash_limit", "wide_hash", "no_hash"):
            ix.set_tuning(kk, 0)
        ix.set_tuning(key, value)
        got = ix.search_batch(q, 10, 50, trace=True, visit_cap=512)
        assert np.array_equal(got[0], g_ids) and np.array_equal(bits(got[1]), bits(g_d)), key
        assert np.array_equal(got[3].n_dist, tr.n_dist) and np.array_equal(got[3].visit_ids, tr.visit_ids), key
    ix.close()
    gpq.close()


@pytest.mark.parametrize("d,M,K", [(96, 8, 64), (96, 16, 32), (64, 32, 16), (96, 12, 32), (90, 6, 16), (96, 3, 32)])
def test_neighbour_code_rows_follow_the_graph(oracle, d, M, K):
    """Quantizers of up to 32 sub-vectors: a node's neighbours' code rows are stored once more behind its adjacency row
    (node.go:37-54: the reference keeps a node's neighbours as cached point objects), so that a hop is one fetch.  That
    copy has to follow everything that changes adjacency rows or codes -- attach, insert, delete, an aborted and a
    committed transaction, set_codes, compact, growth of the tables -- and the walk over it is the oracle's walk: ids,
    distance bits, visit order, counters; with and without filters, over the start node's overflow list too."""
    from semadb_amd import vamana, vectorstore as vs
    metric = "euclidean"
    rng = np.random.default_rng(1000 * d + M)
    n0, n1 = 500, 300
    base = unit_rows(rng, n0 + n1, d)
    o = build_oracle_index(oracle, base[:n0], metric, R=16, L=30)
    ids, vecs, off, edges = o.export()
    train = vecs[1:401].copy()
    first = rng.integers(0, 400, M)
    opq = oracle.PQ(d, metric, M, K)
    opq.fit(train.copy(), first, alias=True)
    assert o.attach_pq(opq, np.stack([opq.encode(v) for v in vecs])) == 0
    # capacity just above the first load: the inserts below make the tables grow (reserve copies the code rows too)
    ix = vamana.NewIndexVamana("pq", vamana.IndexVectorVamanaParameters(d, metric, 30, 16, 1.2), strict=False, capacity=n0 + 8)
    ix.load(ids, vecs, off, edges)
    gpq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    gpq.Fit(train.copy(), first, alias=True)
    vs.attach(ix, gpq)
    q = unit_rows(rng, 24, d)

    def same_walks(tag, filters=None):
        g_ids, g_d, g_c, tr = ix.search_batch(q, 10, 30, filters=filters, trace=True, visit_cap=512)
        for k in range(len(q)):
            o_ids, o_d, o_vis, o_tr = o.search(q[k], 10, 30, filter_ids=sorted(filters[k]) if filters else None)
            assert int(g_c[k]) == len(o_ids), tag
            assert np.array_equal(g_ids[k, :len(o_ids)], o_ids), tag
            assert np.array_equal(bits(g_d[k, :len(o_ids)]), bits(o_d)), tag
            assert int(tr.n_hop[k]) == o_tr.n_hop and int(tr.n_dist[k]) == o_tr.n_dist, tag
            assert np.array_equal(tr.visit_ids[k, :o_tr.n_hop], o_vis), tag

    same_walks("attached")
    # inserts (sequential rounds = the oracle's loop), the tables grow
    new_ids = np.arange(n0 + 2, n0 + 2 + n1, dtype=np.uint64)
    for i in range(n1):
        assert o.insert(int(new_ids[i]), base[n0 + i]) == 0
    ix.insert_batch(new_ids, base[n0:], round_size=1)
    assert_same_graph(ix, o)
    same_walks("after inserts")
    live = [int(v) for v in o.export()[0][1:]]
    same_walks("filtered", [set(int(v) for v in rng.choice(live, size=60, replace=False)) for _ in range(len(q))])
    # a transaction that is aborted leaves the committed rows and their code rows alone
    ix.begin_write()
    ix.delete_batch(new_ids[:40])
    same_walks("inside an open transaction")  # searches walk the committed copy
    ix.abort_write()
    same_walks("after abort")
    # centroid ids exist once, not per graph version: they are not set inside a transaction (it could not take them back)
    ix.begin_write()
    from semadb_amd._lib import SemaDBError
    with pytest.raises(SemaDBError):
        vs.set_codes(ix, np.array(live[:3], dtype=np.uint64), np.zeros((3, M), dtype=np.uint8))
    ix.abort_write()
    same_walks("after a refused set_codes and an abort")
    # deletes (stragglers go onto the start node: rows change far from the deleted ones)
    dead = rng.choice(np.arange(2, n0 + 2 + n1, dtype=np.uint64), 120, replace=False)
    assert o.delete(dead) == 0
    ix.delete_batch(dead)
    assert_same_graph(ix, o)
    same_walks("after deletes")
    # set_codes: the code rows of some points change (k-means labels / codes from the bucket); every node that has one of
    # them as a neighbour carries a copy
    some = np.array(sorted(rng.choice([v for v in live if v not in set(int(x) for x in dead)], size=50, replace=False)), dtype=np.uint64)
    newc = rng.integers(0, K, size=(len(some), M)).astype(np.uint8)
    vs.set_codes(ix, some, newc)
    live_ids = o.export()[0]
    assert o.attach_pq(opq, vs.get_codes(ix, live_ids)) == 0  # the oracle takes the whole table (storage order of the live rows)
    assert np.array_equal(vs.get_codes(ix, some), newc)
    same_walks("after set_codes")
    ix.compact()
    same_walks("after compact")
    # and a second quantizer with another layout replaces the first
    ix.close()
    gpq.close()


@pytest.mark.parametrize("metric,d,M,K", [("euclidean", 32, 8, 2), ("cosine", 32, 4, 4), ("dot", 48, 16, 16), ("euclidean", 24, 2, 3),
                                          ("cosine", 64, 32, 8), ("euclidean", 40, 8, 256)])
def test_two_wave_walk_names_what_the_array_says(metric, d, M, K):
    """k_greedy_search_pq2's walker names the next node from the array's first unvisited entry and the hop's distances
    instead of waiting for AddWithLimit.  The rule's edge is equal distances and the array's tail: a point equal to the
    entry it is compared with, two points sharing the smallest distance, an entry that a full array drops.  Coarse
    quantizers (K = 2 .. 16: a handful of distinct sums, ties in every hop) over every searchSize from a nearly empty
    array to the reference's maximum, against the one-wave kernel (SDB_TUNE_PQ_NARROW = 1: AddWithLimit as written,
    itself held to the oracle above): ids, distance bits, visit order, the three counters, query by query."""
    from semadb_amd import vamana, vectorstore as vs
    from tests.helpers import start_vector
    rng = np.random.default_rng(7000 + 31 * M + K)
    n = 6000
    lat = rng.standard_normal((6, d)).astype(np.float32)
    base = rng.standard_normal((n, 6)).astype(np.float32) @ lat + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    ix = vamana.NewIndexVamana("tw", vamana.IndexVectorVamanaParameters(d, metric, 64, 32, 1.2), strict=False)
    ix.set_start(start_vector(np.random.default_rng(5), d))
    ix.insert_batch(None, base)
    pq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M), d)
    pq.Fit(base[:1500].copy(), rng.integers(0, 1500, M), alias=True)
    vs.attach(ix, pq)
    q = base[rng.choice(n, 512, replace=False)] + 0.05 * rng.standard_normal((512, d)).astype(np.float32)
    # (the two-wave kernel serves tables of up to 2 048 entries on graphs whose start node has no overflow list --
    # index.hip pq_two_waves; rocprofv3 --kernel-trace of this test shows k_greedy_search_pq2 beside k_greedy_search<PQDist>)
    assert M * K <= 2048
    _, _, g_off, _ = ix.export(with_vectors=False)
    assert int(g_off[1] - g_off[0]) <= 64
    ties = 0
    for L in (1, 2, 3, 5, 10, 17, 33, 64, 75, 96):
        ix.set_tuning("pq_narrow", 0)
        two = ix.search_batch(q, min(10, L), L, trace=True, visit_cap=512)
        ix.set_tuning("pq_narrow", 1)
        one = ix.search_batch(q, min(10, L), L, trace=True, visit_cap=512)
        assert np.array_equal(two[2], one[2]), L
        assert np.array_equal(two[0], one[0]) and np.array_equal(bits(two[1]), bits(one[1])), L
        assert np.array_equal(two[3].visit_ids, one[3].visit_ids), L
        for f in ("n_dist", "n_hop", "n_edges"):
            assert np.array_equal(getattr(two[3], f), getattr(one[3], f)), (L, f)
        dd = two[1][:, : min(10, L)]
        ties += int((dd[:, 1:] == dd[:, :-1]).sum())
    ix.set_tuning("pq_narrow", 0)
    if K ** M <= 4096:  # a few hundred distinct sums at most
        assert ties > 100, ties  # the case the test is about did occur: equal distances among the results themselves
    ix.close()
    pq.close()
