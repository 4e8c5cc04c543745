"""K2/K3 parity (GPU): greedy search over an HBM-resident graph returns the oracle's result ids,
distance bits, visit order, n_dist and n_hop on the same graph and queries."""
import numpy as np
import pytest

from tests.helpers import bits, build_oracle_index, unit_rows

pytestmark = pytest.mark.gpu


def _gpu_index(o, d, metric, R, L, strict=False):
    from semadb_amd import vamana
    ids, vecs, offsets, edges = o.export()
    ix = vamana.NewIndexVamana("t", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=strict)
    ix.load(ids, vecs, offsets, edges)
    return ix


def _check_batch(o, ix, queries, limit, L, visit_cap=1024):
    """both forms of the walk: one wave per query (wide_walk = 1) and the workgroup-per-query walk of small calls
    (wide_walk = 2: a hop's rows split over four waves; shapes it does not cover fall back to the one-wave kernel)"""
    for mode in (1, 2):
        ix.set_tuning("wide_walk", mode)
        _check_batch_mode(o, ix, queries, limit, L, visit_cap)
    ix.set_tuning("wide_walk", 0)


def _check_batch_mode(o, ix, queries, limit, L, visit_cap):
    g_ids, g_d, g_c, tr = ix.search_batch(queries, limit, L, trace=True, visit_cap=visit_cap)
    for q in range(queries.shape[0]):
        o_ids, o_d, o_vis, o_tr = o.search(queries[q], limit, L)
        assert int(g_c[q]) == len(o_ids)
        assert np.array_equal(g_ids[q, :len(o_ids)], o_ids), "query %d ids" % q
        assert np.array_equal(bits(g_d[q, :len(o_ids)]), bits(o_d)), "query %d dist bits" % q
        assert int(tr.n_hop[q]) == o_tr.n_hop and int(tr.n_dist[q]) == o_tr.n_dist
        assert int(tr.n_edges[q]) == o_tr.n_edges
        assert np.array_equal(tr.visit_ids[q, :o_tr.n_hop], o_vis), "query %d visit order" % q


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,n", [(2, 300), (33, 400), (96, 600), (128, 800), (384, 700), (160, 300), (768, 300),
                                 # 1024 / 1536 / 2048 / 3072: register-query kernels; 1280: rows zero-padded to the
                                 # 1536 kernel; 2112 and 4096: the generic (LDS tile) kernel
                                 (1024, 200), (1536, 200), (2048, 150), (3072, 120), (1280, 150), (2112, 100), (4096, 80),
                                 # the reference's ann-benchmarks shapes with a tail (d % 32 != 0): glove-25/-100, mnist
                                 (25, 500), (100, 500), (784, 300)])
def test_search_parity(oracle, metric, d, n):
    rng = np.random.default_rng(d * 13 + n)
    base = unit_rows(rng, n, d) if d > 2 else rng.random((n, d), dtype=np.float32)
    o = build_oracle_index(oracle, base, metric, R=32, L=50)
    ix = _gpu_index(o, d, metric, 32, 50)
    queries = unit_rows(rng, 24, d) if d > 2 else rng.random((24, d), dtype=np.float32)
    _check_batch(o, ix, queries, 10, 50)
    _check_batch(o, ix, base[:8], 5, 25)  # queries that are in the set: exact zero / tie handling
    ix.close()


def test_search_reference_defaults(oracle):
    # the reference's default parameters: searchSize 75, degreeBound 64, alpha 1.2 (README.md:184-200)
    rng = np.random.default_rng(42)
    base = unit_rows(rng, 2500, 128)
    o = build_oracle_index(oracle, base, "cosine", R=64, L=75)
    ix = _gpu_index(o, 128, "cosine", 64, 75, strict=True)
    _check_batch(o, ix, unit_rows(rng, 64, 128), 10, 75)
    _check_batch(o, ix, unit_rows(rng, 8, 128), 75, 75)
    ix.close()


def test_search_many_ties(oracle):
    """integer grid data: lots of equal distances, exercises the strict </> rules of distset.go:184,196"""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 4, size=(500, 8)).astype(np.float32)
    o = build_oracle_index(oracle, base, "euclidean", R=32, L=40)
    ix = _gpu_index(o, 8, "euclidean", 32, 40)
    q = rng.integers(0, 4, size=(32, 8)).astype(np.float32)
    _check_batch(o, ix, q, 10, 40)
    ix.close()


def test_large_search_size(oracle):
    """search sizes beyond the reference's API maximum use the 512-entry candidate array"""
    rng = np.random.default_rng(4)
    base = unit_rows(rng, 1500, 64)
    o = build_oracle_index(oracle, base, "euclidean", R=32, L=50)
    ix = _gpu_index(o, 64, "euclidean", 32, 50)
    q = unit_rows(rng, 8, 64)
    _check_batch(o, ix, q, 10, 200, visit_cap=2048)
    _check_batch(o, ix, q, 10, 129, visit_cap=2048)
    ix.close()


def test_deterministic_reference_data(oracle):
    # shard/index/dispatch_test.go:66-89 data, shard/index/search_test.go:89-144,414-457 assertions
    from semadb_amd import vamana
    base = np.array([[ii, ii + 1] for ii in range(2, 102)], dtype=np.float32)
    o = build_oracle_index(oracle, base, "euclidean", R=64, L=75)
    ix = _gpu_index(o, 2, "euclidean", 64, 75, strict=True)
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([42, 43], 75, 10))
    assert len(res) == 10 and res[0].NodeId == 42 and 42 in rset
    w = 0.5
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([42, 43], 75, 5, Weight=w))
    assert rset == {40, 41, 42, 43, 44}
    for r in res:
        assert r.HybridScore + r.HybridScore == -r.Distance
    ix.close()


def test_empty_index_and_errors(oracle):
    # Test_EmptySearch vamana_test.go:213-228 ; search.go:23-25 ; models/search.go:287-297
    from semadb_amd import vamana, SemaDBError
    p = vamana.IndexVectorVamanaParameters(2, "euclidean", 75, 64, 1.2)
    ix = vamana.NewIndexVamana("e", p)
    with pytest.raises(SemaDBError):  # no start node yet
        ix.Search(vamana.SearchVectorVamanaOptions([0.5, 0.5], 75, 10))
    ix.set_start([0.6, 0.8])
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([0.5, 0.5], 75, 10))
    assert len(rset) == 0 and res == []
    with pytest.raises(SemaDBError):
        ix.Search(vamana.SearchVectorVamanaOptions([0.5, 0.5], 25, 30))  # searchSize < k
    with pytest.raises(SemaDBError):
        ix.Search(vamana.SearchVectorVamanaOptions([0.5, 0.5], 200, 10))  # strict: 25..75
    with pytest.raises(SemaDBError):
        vamana.NewIndexVamana("bad", vamana.IndexVectorVamanaParameters(2, "euclidean", 75, 64, 3.0))
    with pytest.raises(SemaDBError):
        vamana.NewIndexVamana("bad", vamana.IndexVectorVamanaParameters(5000, "euclidean"))
    ix.close()


def test_load_drops_unknown_and_duplicate_edges(oracle):
    """ItemCache.GetMany skips unknown ids (itemcache.go:109-128); a repeated id can never pass
    CheckAndVisit twice (distset.go:174)"""
    rng = np.random.default_rng(8)
    base = unit_rows(rng, 200, 32)
    o = build_oracle_index(oracle, base, "euclidean", R=16, L=30)
    ids, vecs, offsets, edges = o.export()
    # splice junk into every row: an unknown id and a duplicate of the first edge
    new_off, new_edges = [0], []
    for i in range(len(ids)):
        row = list(edges[int(offsets[i]):int(offsets[i + 1])])
        if row:
            row = row[:1] + [10 ** 9 + i] + row[1:] + [row[0]]
        new_edges.extend(row)
        new_off.append(len(new_edges))
    from semadb_amd import vamana
    ix = vamana.NewIndexVamana("j", vamana.IndexVectorVamanaParameters(32, "euclidean", 30, 16, 1.2), strict=False)
    ix.load(ids, vecs, np.array(new_off, dtype=np.uint64), np.array(new_edges, dtype=np.uint64))
    o2 = oracle.Index(32, "euclidean", 16, 30, 1.2)
    o2.load(ids, vecs, np.array(new_off, dtype=np.uint64), np.array(new_edges, dtype=np.uint64))
    q = unit_rows(rng, 16, 32)
    g_ids, g_d, g_c, tr = ix.search_batch(q, 10, 30, trace=True, visit_cap=256)
    for k in range(16):
        o_ids, o_d, o_vis, o_tr = o2.search(q[k], 10, 30)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids)
        assert np.array_equal(tr.visit_ids[k, :o_tr.n_hop], o_vis)
        assert int(tr.n_dist[k]) == o_tr.n_dist
    e_ids, e_vecs, e_off, e_edges = ix.export()
    assert np.array_equal(e_ids, ids) and np.array_equal(bits(e_vecs), bits(vecs))
    assert np.array_equal(e_off, offsets) and np.array_equal(e_edges, edges)  # junk removed, order kept
    ix.close()


def test_sparse_ids_and_device_queries(oracle):
    import torch
    rng = np.random.default_rng(10)
    base = unit_rows(rng, 300, 64)
    d = 64
    o = oracle.Index(d, "cosine", 32, 50, 1.2)
    from tests.helpers import start_vector
    o.set_start(start_vector(rng, d))
    sparse_ids = np.sort(rng.choice(np.arange(2, 100000), size=300, replace=False)).astype(np.uint64)
    for i in range(300):
        assert o.insert(int(sparse_ids[i]), base[i]) == 0
    ix = _gpu_index(o, d, "cosine", 32, 50)
    q = unit_rows(rng, 16, d)
    tq = torch.from_numpy(q).cuda()
    g_ids, g_d, g_c, _ = ix.search_batch(tq, 10, 50)
    torch.cuda.synchronize()
    g_ids = g_ids.cpu().numpy().view(np.uint64)
    for k in range(16):
        o_ids, o_d, _, _ = o.search(q[k], 10, 50)
        assert np.array_equal(g_ids[k, :len(o_ids)], o_ids)
        assert np.array_equal(bits(g_d[k, :len(o_ids)].cpu().numpy()), bits(o_d))
    # plainStore.DistanceFromFloat batched, unknown id -> MaxFloat32 (plain.go:78-82)
    cand = np.tile(np.array([sparse_ids[0], sparse_ids[5], 1, 4242424242], dtype=np.uint64), (16, 1))
    got = ix.distance_batch(q, cand)
    want = oracle.distance_matrix(q, np.stack([base[0], base[5]]), "cosine")
    assert np.array_equal(bits(got[:, :2]), bits(want))
    assert np.all(got[:, 3] == np.finfo(np.float32).max)
    ix.close()


def test_hash_set_overflow_falls_back_to_bitset(oracle, monkeypatch):
    """The LDS hash visited set is exact; when it fills, the query reruns on the HBM bitset.  A tiny
    limit forces that path for every query: results must not change."""
    rng = np.random.default_rng(77)
    base = unit_rows(rng, 3000, 96)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu_index(o, 96, "cosine", 32, 50)
    q = unit_rows(rng, 32, 96)
    _check_batch(o, ix, q, 10, 50)            # hash path
    ix.set_tuning("hash_limit", 40)
    _check_batch(o, ix, q, 10, 50)            # every query overflows -> bitset rerun
    ix.set_tuning("hash_limit", 700)
    _check_batch(o, ix, q, 10, 50)            # a mix of both
    ix.close()


def test_async_batches_on_two_streams(oracle):
    """Device-memory calls return before the kernel runs; batches in flight on different streams must not
    share a visited-set workspace."""
    import torch
    rng = np.random.default_rng(5)
    base = unit_rows(rng, 4000, 64)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu_index(o, 64, "cosine", 32, 50)
    qs = [torch.from_numpy(unit_rows(rng, 256, 64)).cuda() for _ in range(6)]
    ref = []
    for q in qs:
        ids, d, c, _ = ix.search_batch(q, 10, 120)   # bitset path (searchSize > 96)
        torch.cuda.synchronize()
        ref.append(ids.clone())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None] * len(qs)
    for rep in range(3):
        for i, q in enumerate(qs):
            with torch.cuda.stream(streams[i % 2]):
                outs[i] = ix.search_batch(q, 10, 120)[0]
        torch.cuda.synchronize()
        for i in range(len(qs)):
            assert torch.equal(outs[i], ref[i])
    ix.close()


@pytest.mark.parametrize("d,nq", [(96, 40), (384, 300), (2112, 24)])
def test_host_memory_search_in_place(oracle, d, nq):
    """Queries and outputs in page-locked host memory (sdb_host_alloc): the walk reads and writes them in place, one
    launch per call.  Same answers as the staged path (pageable buffers, or SDB_TUNE_NO_ZERO_COPY), bit for bit, plain
    and filtered, also when only some of the buffers are page-locked, and the oracle's."""
    from semadb_amd import _buf
    rng = np.random.default_rng(d + nq)
    base = unit_rows(rng, 1500, d)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu_index(o, d, "cosine", 32, 50)
    queries = unit_rows(rng, nq, d)
    k, L = 10, 50
    ref_ids, ref_d, ref_c, _ = ix.search_batch(queries, k, L)  # pageable numpy: staged
    for q in range(0, nq, 7):
        o_ids, o_d, _, _ = o.search(queries[q], k, L)
        assert np.array_equal(ref_ids[q, :len(o_ids)], o_ids) and np.array_equal(bits(ref_d[q, :len(o_ids)]), bits(o_d))
    pq_ = _buf.pinned_empty((nq, d), "float32")
    pq_[:] = queries
    p_ids, p_d, p_c = _buf.pinned_empty((nq, k), "uint64"), _buf.pinned_empty((nq, k), "float32"), _buf.pinned_empty((nq,), "uint32")
    filt = [set(int(v) for v in rng.choice(np.arange(2, 1502), size=int(rng.integers(0, 60)), replace=False)) for _ in range(nq)]
    f_ids, f_d, f_c, _ = ix.search_batch(queries, k, L, filters=filt)
    for no_zc in (0, 1, 0):
        ix.set_tuning("no_zero_copy", no_zc)
        for rep in range(3):  # the same slabs call after call, like a batcher's
            p_ids[:] = 0xDEADBEEF
            p_d[:] = -7.0
            p_c[:] = 99
            ix.search_batch(pq_, k, L, out=(p_ids, p_d, p_c))
            assert np.array_equal(p_c, ref_c) and np.array_equal(p_ids, ref_ids) and np.array_equal(bits(p_d), bits(ref_d))
        ix.search_batch(pq_, k, L, filters=filt, out=(p_ids, p_d, p_c))
        assert np.array_equal(p_c, f_c) and np.array_equal(p_ids, f_ids) and np.array_equal(bits(p_d), bits(f_d))
        # page-locked queries, pageable outputs -- and the other way round
        m_ids, m_d, m_c, _ = ix.search_batch(pq_, k, L)
        assert np.array_equal(m_ids, ref_ids) and np.array_equal(bits(m_d), bits(ref_d)) and np.array_equal(m_c, ref_c)
        ix.search_batch(queries, k, L, out=(p_ids, p_d, p_c))
        assert np.array_equal(p_ids, ref_ids) and np.array_equal(bits(p_d), bits(ref_d)) and np.array_equal(p_c, ref_c)
        # a slice in the middle of a block: the device's view of it is the block's plus the offset
        half = nq // 2
        ix.search_batch(pq_[half:], k, L, out=(p_ids[half:], p_d[half:], p_c[half:]))
        assert np.array_equal(p_ids[half:], ref_ids[half:]) and np.array_equal(p_c[half:], ref_c[half:])
    ix.set_tuning("wide_walk", 1)  # the one-wave kernel for a small call, too
    ix.search_batch(pq_, k, L, out=(p_ids, p_d, p_c))
    assert np.array_equal(p_ids, ref_ids) and np.array_equal(bits(p_d), bits(ref_d))
    ix.set_tuning("wide_walk", 0)
    # a quantized store: the table kernel reads the queries from a staged copy (host memory is not cached on the device),
    # the walk still writes its results in place
    if d % 8 == 0:
        from semadb_amd import vectorstore as vs
        pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(16, 8, 1000), d)
        pq.Fit(base[:1000].copy(), np.arange(8) * 11, alias=False)
        vs.attach(ix, pq)
        q_ids, q_d, q_c, _ = ix.search_batch(queries, k, L)  # pageable: staged both ways
        for no_zc in (0, 1):
            ix.set_tuning("no_zero_copy", no_zc)
            p_ids[:] = 7
            p_d[:] = -1.0
            p_c[:] = 99
            ix.search_batch(pq_, k, L, out=(p_ids, p_d, p_c))
            assert np.array_equal(p_c, q_c) and np.array_equal(p_ids, q_ids) and np.array_equal(bits(p_d), bits(q_d))
    ix.close()
    del pq_, p_ids, p_d, p_c
