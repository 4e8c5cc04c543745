"""Filtered greedy search on device (search.go:33-51,93-95): seeds, unsorted Add, result set with its own
visited set -- identical ids, distance bits, visit order and n_dist to the oracle."""
import numpy as np
import pytest

from tests.helpers import bits, build_oracle_index, unit_rows

pytestmark = pytest.mark.gpu


def _gpu(o, d, metric, R, L):
    from semadb_amd import vamana
    ids, vecs, off, edges = o.export()
    ix = vamana.NewIndexVamana("f", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    ix.load(ids, vecs, off, edges)
    return ix


def test_reference_filter_kats(oracle):
    # shard/index/search_test.go:196-244 (filter {47} -> one result, distance exactly 50) and
    # :246-288 (filter {42..46} -> exactly those five, first 42) on the dispatch_test.go:66-89 data
    from semadb_amd import vamana
    base = np.array([[ii, ii + 1] for ii in range(2, 102)], dtype=np.float32)
    o = build_oracle_index(oracle, base, "euclidean", R=64, L=75)
    ix = _gpu(o, 2, "euclidean", 64, 75)
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([42, 43], 75, 10), filter={47})
    assert rset == {47} and len(res) == 1 and res[0].Distance == np.float32(50)
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions([42, 43], 75, 10), filter={42, 43, 44, 45, 46})
    assert rset == {42, 43, 44, 45, 46} and len(res) == 5 and res[0].NodeId == 42
    ix.close()


def test_filter_search_property(oracle):
    # Test_FilterSearch vamana_test.go:254-276: filter of 3 ids -> 3 results, first is the query point
    from semadb_amd import vamana
    rng = np.random.default_rng(123)
    pts = rng.random((200, 2), dtype=np.float32)
    o = build_oracle_index(oracle, pts, "euclidean", R=64, L=75)
    ix = _gpu(o, 2, "euclidean", 64, 75)
    rset, res = ix.Search(vamana.SearchVectorVamanaOptions(pts[0], 75, 10), filter={2, 3, 4})
    assert len(res) == 3 and res[0].NodeId == 2
    ix.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("d,n,L,k", [(32, 1200, 50, 10), (96, 900, 30, 30), (384, 600, 75, 10)])
def test_filtered_batch_parity(oracle, metric, d, n, L, k):
    rng = np.random.default_rng(d + n + L)
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, metric, R=32, L=50)
    ix = _gpu(o, d, metric, 32, 50)
    nq = 24
    q = unit_rows(rng, nq, d)
    all_ids = np.arange(2, n + 2)
    filters = []
    for i in range(nq):
        kind = i % 6
        if kind == 0:
            f = rng.choice(all_ids, size=5, replace=False)            # tiny: fewer than k
        elif kind == 1:
            f = rng.choice(all_ids, size=L, replace=False)            # exactly searchSize seeds
        elif kind == 2:
            f = rng.choice(all_ids, size=n // 2, replace=False)       # large: most seeds ignored
        elif kind == 3:
            f = np.concatenate([rng.choice(all_ids, size=40, replace=False), [10 ** 7 + i, 10 ** 8]])  # unknown ids
        elif kind == 4:
            f = np.concatenate([[1], rng.choice(all_ids, size=20, replace=False)])  # contains the start id
        else:
            f = np.array([], dtype=np.int64)                          # empty filter (not nil): no results
        filters.append(set(int(v) for v in f))
    g_ids, g_d, g_c, tr = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    for i in range(nq):
        o_ids, o_d, o_vis, o_tr = o.search(q[i], k, L, filter_ids=sorted(filters[i]))
        assert int(g_c[i]) == len(o_ids), (i, g_c[i], len(o_ids))
        assert np.array_equal(g_ids[i, :len(o_ids)], o_ids), i
        assert np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d)), i
        assert int(tr.n_hop[i]) == o_tr.n_hop and int(tr.n_dist[i]) == o_tr.n_dist, i
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), i
        assert set(int(v) for v in o_ids) <= filters[i]
    # a call of 24 queries takes the workgroup-per-query walk (k_greedy_search_wide, filtered form); one wave per query
    # walks the same path
    ix.set_tuning("wide_walk", 1)
    w_ids, w_d, w_c, wtr = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    assert np.array_equal(w_ids, g_ids) and np.array_equal(bits(w_d), bits(g_d)) and np.array_equal(w_c, g_c)
    assert np.array_equal(wtr.visit_ids, tr.visit_ids) and np.array_equal(wtr.n_dist, tr.n_dist)
    ix.set_tuning("wide_walk", 0)
    # the table's ids are consecutive, so the filter ids above were resolved to slots on the device (k_filter_resolve);
    # the host's hash-map translation gives the same walk
    ix.set_tuning("host_filters", 1)
    h_ids, h_d, h_c, htr = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    assert np.array_equal(h_ids, g_ids) and np.array_equal(bits(h_d), bits(g_d)) and np.array_equal(h_c, g_c)
    assert np.array_equal(htr.visit_ids, tr.visit_ids) and np.array_equal(htr.n_dist, tr.n_dist)
    ix.close()


def test_filters_resolved_on_the_device_are_validated(oracle):
    """the device-side resolution keeps the ABI's contract: ids of a query strictly ascending (an error otherwise, and
    no search), offsets non-decreasing, unknown / out-of-range ids skipped, empty filters, ids of rows past the table"""
    from semadb_amd import SemaDBError
    rng = np.random.default_rng(77)
    n, d = 500, 32
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu(o, d, "cosine", 32, 50)
    q = unit_rows(rng, 3, d)
    off = np.array([0, 3, 3, 7], dtype=np.uint64)
    good = np.array([5, 9, 400, 2, 3, 10 ** 12, 2 ** 63], dtype=np.uint64)
    g_ids, g_d, g_c, _ = ix.search_batch(q, 10, 50, filters=(off, good))
    assert sorted(int(v) for v in g_ids[0, :int(g_c[0])]) == [5, 9, 400] and int(g_c[1]) == 0
    assert sorted(int(v) for v in g_ids[2, :int(g_c[2])]) == [2, 3]
    for i, f in enumerate([[5, 9, 400], [], [2, 3]]):
        o_ids, o_d, _, _ = o.search(q[i], 10, 50, filter_ids=f)
        assert np.array_equal(g_ids[i, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d))
    bad = good.copy()
    bad[1], bad[2] = 400, 9  # query 0: 5, 400, 9
    with pytest.raises(SemaDBError) as ei:
        ix.search_batch(q, 10, 50, filters=(off, bad))
    assert "not strictly ascending" in str(ei.value) and "query 0" in str(ei.value)
    dup = good.copy()
    dup[4] = 2  # query 2: 2, 2, ...
    with pytest.raises(SemaDBError):
        ix.search_batch(q, 10, 50, filters=(off, dup))
    with pytest.raises(SemaDBError):
        ix.search_batch(q, 10, 50, filters=(np.array([0, 3, 2, 7], dtype=np.uint64), good))
    g2 = ix.search_batch(q, 10, 50, filters=(off, good))  # and the next call is served
    assert np.array_equal(g2[0], g_ids)
    ix.close()


def test_filter_on_rows_stored_out_of_id_order(oracle):
    """the filter ids of a query translate to slots that are NOT ascending when the rows are not stored in id order
    (the translation sorts them only then): the exported graph is loaded with its rows shuffled (the start node stays
    first), and filtered searches must still answer like the oracle"""
    from semadb_amd import vamana
    d, n, L, k = 64, 800, 40, 10
    rng = np.random.default_rng(2024)
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ids, vecs, off, edges = o.export()
    perm = np.concatenate([[0], 1 + rng.permutation(len(ids) - 1)])
    deg = np.diff(off)
    p_off = np.zeros_like(off)
    p_off[1:] = np.cumsum(deg[perm])
    p_edges = np.concatenate([edges[off[i]:off[i + 1]] for i in perm]) if len(edges) else edges
    ix = vamana.NewIndexVamana("f", vamana.IndexVectorVamanaParameters(d, "cosine", L, 32, 1.2), strict=False)
    ix.load(ids[perm], vecs[perm], p_off, p_edges)
    nq = 12
    q = unit_rows(rng, nq, d)
    all_ids = np.arange(2, n + 2)
    filters = [set(int(v) for v in rng.choice(all_ids, size=s, replace=False)) for s in (3, L, 200, n // 2) * 3]
    g_ids, g_d, g_c, tr = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    for i in range(nq):
        o_ids, o_d, o_vis, o_tr = o.search(q[i], k, L, filter_ids=sorted(filters[i]))
        assert int(g_c[i]) == len(o_ids), (i, g_c[i], len(o_ids))
        assert np.array_equal(g_ids[i, :len(o_ids)], o_ids), i
        assert np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d)), i
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), i
    ix.close()


def test_filters_on_a_table_with_holes_resolve_on_the_device(oracle):
    """after deletes the ids are no longer consecutive: the filter ids are resolved by a probe of the committed view's
    id -> slot table on the device (sdb_index::IdMap), rebuilt for every published view.  Same answers as the host's
    translation (tuning host_filters) and as the oracle -- ids of deleted rows and unknown ids are skipped, rows added
    after the deletes resolve once committed, and in a batch for which ids and slots disagree on the order (an id that
    was deleted and inserted again sits behind later ids) the walk answers Contains from the ids themselves."""
    rng = np.random.default_rng(404)
    n, d, L, k = 2000, 32, 50, 10
    base = unit_rows(rng, n + 300, d)
    o = build_oracle_index(oracle, base[:n], "cosine", R=32, L=50)
    ix = _gpu(o, d, "cosine", 32, 50)
    gone = np.array(sorted(int(v) for v in rng.choice(np.arange(2, n + 2), size=150, replace=False)), dtype=np.uint64)
    ix.delete_batch(gone)
    assert o.delete(gone) == 0
    nq = 16
    q = unit_rows(rng, nq, d)

    def check(filters, label):
        g = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
        ix.set_tuning("host_filters", 1)
        h = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
        ix.set_tuning("host_filters", 0)
        assert np.array_equal(g[0], h[0]) and np.array_equal(bits(g[1]), bits(h[1])) and np.array_equal(g[2], h[2]), label
        assert np.array_equal(g[3].visit_ids, h[3].visit_ids) and np.array_equal(g[3].n_dist, h[3].n_dist), label
        for i in range(nq):
            o_ids, o_d, o_vis, o_tr = o.search(q[i], k, L, filter_ids=sorted(filters[i]))
            assert int(g[2][i]) == len(o_ids), (label, i)
            assert np.array_equal(g[0][i, :len(o_ids)], o_ids) and np.array_equal(bits(g[1][i, :len(o_ids)]), bits(o_d)), (label, i)
            assert np.array_equal(g[3].visit_ids[i, :o_tr.n_hop], o_vis), (label, i)

    live = np.setdiff1d(np.arange(2, n + 2), gone.astype(np.int64))
    sizes = (4, L, 300, n // 2) * 4
    filters = [set(int(v) for v in rng.choice(live, size=s, replace=False)) | set(int(v) for v in rng.choice(gone, size=5))
               | {n + 5000 + i, 2 ** 40 + i} for i, s in enumerate(sizes)]
    check(filters, "holes")
    # rows appended after the deletes: new, larger ids (ids and slots still agree on the order)
    new_ids = np.arange(n + 2, n + 202, dtype=np.uint64)
    ix.insert_batch(new_ids, base[n:n + 200], round_size=1)
    for i in range(200):
        assert o.insert(int(new_ids[i]), base[n + i]) == 0
    filters2 = [f | set(int(v) for v in rng.choice(new_ids, size=20, replace=False)) for f in filters]
    check(filters2, "appended")
    # ids that were deleted come back: their rows sit behind rows with larger ids
    back = gone[:100]
    ix.insert_batch(back, base[n + 200:n + 300], round_size=1)
    for i in range(100):
        assert o.insert(int(back[i]), base[n + 200 + i]) == 0
    filters3 = [f | set(int(v) for v in back[:30]) for f in filters2]
    check(filters3, "re-inserted ids")
    check([set(int(v) for v in live[:40])] * nq, "a batch that does not meet the re-inserted ids")
    ix.close()


def test_filter_argument_errors(oracle):
    from semadb_amd import vamana, SemaDBError, _lib
    import ctypes as C
    rng = np.random.default_rng(1)
    base = unit_rows(rng, 100, 16)
    o = build_oracle_index(oracle, base, "euclidean", R=16, L=30)
    ix = _gpu(o, 16, "euclidean", 16, 30)
    q = unit_rows(rng, 1, 16)
    off = np.array([0, 3], dtype=np.uint64)
    bad = np.array([5, 4, 9], dtype=np.uint64)  # not ascending
    ids = np.zeros((1, 5), np.uint64); d = np.zeros((1, 5), np.float32); c = np.zeros(1, np.uint32)
    rc = _lib.lib().sdb_index_search_batch(ix._h, 1, q.ctypes.data_as(C.c_void_p), 5, 30, off.ctypes.data_as(C.c_void_p),
                                           bad.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                           d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), None, 0, None)
    assert rc != 0 and b"ascending" in _lib.lib().sdb_last_error()
    ix.close()


def test_filtered_walk_is_the_same_under_every_visited_set(oracle):
    """Round 3: filtered walks keep both visited sets (the search set's and the result set's, search.go:37) in LDS hash
    tables.  Same answers, visit order and counters with the tables at their default size, with both forced to spill
    to their HBM bitsets after a handful of ids, and on the bitsets from the start -- and all equal the oracle."""
    rng = np.random.default_rng(2025)
    d, n, L, k = 64, 2500, 60, 10
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu(o, d, "cosine", 32, 50)
    nq = 16
    q = unit_rows(rng, nq, d)
    all_ids = np.arange(2, n + 2)
    filters = [set(int(v) for v in rng.choice(all_ids, size=s, replace=False)) for s in
               [3, 60, 61, 200, 1200, 2400, 10, 900] * 2]
    runs = {}
    for name, tune in [("hash", {}), ("spill", {"hash_limit": 24}), ("bitset", {"no_hash": 1})]:
        for key, v in {"hash_limit": 0, "no_hash": 0, **tune}.items():
            ix.set_tuning(key, v)
        runs[name] = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    for name, (g_ids, g_d, g_c, tr) in runs.items():
        for i in range(nq):
            o_ids, o_d, o_vis, o_tr = o.search(q[i], k, L, filter_ids=sorted(filters[i]))
            assert int(g_c[i]) == len(o_ids), (name, i)
            assert np.array_equal(g_ids[i, :len(o_ids)], o_ids), (name, i)
            assert np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d)), (name, i)
            assert int(tr.n_hop[i]) == o_tr.n_hop and int(tr.n_dist[i]) == o_tr.n_dist, (name, i)
            assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), (name, i)
    ix.close()


@pytest.mark.parametrize("dense", [True, False])
def test_filters_as_bitmaps_give_the_same_walk(oracle, dense):
    """sdb_index_search_batch_bitmap: the filter as the bitmap it is in the reference (roaring64, search.go:33-51,93) --
    {first_id + i : bit i} -- expanded to slots on the device: directly for a table with consecutive ids, through 64-bit ids
    and the view's id -> slot table otherwise (dense = False: a delete leaves a hole).  Same answers as the id lists and as the oracle: ids, distance bits, visit
    order, counters; unknown ids (bits outside the table, a window that starts below the first id), empty bitmaps,
    windows that are not word-aligned with each other, bits of the start node."""
    from semadb_amd import vamana
    rng = np.random.default_rng(91 if dense else 92)
    n, d, L, k = 1500, 32, 50, 10
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, "cosine", R=32, L=50)
    ix = _gpu(o, d, "cosine", 32, 50)
    if not dense:
        gone = np.array([7, 8, 900], dtype=np.uint64)
        ix.delete_batch(gone)
        assert o.delete(gone) == 0
    nq = 20
    q = unit_rows(rng, nq, d)
    all_ids = np.arange(2, n + 2)
    filters = []
    for i in range(nq):
        kind = i % 7
        if kind == 0:
            f = rng.choice(all_ids, size=6, replace=False)
        elif kind == 1:
            f = rng.choice(all_ids, size=n // 3, replace=False)                     # dense: most words full-ish
        elif kind == 2:
            f = np.concatenate([rng.choice(all_ids, size=30, replace=False), [n + 500, n + 4000]])  # past the table
        elif kind == 3:
            f = np.concatenate([[1], rng.choice(all_ids, size=25, replace=False)])  # the start node's id
        elif kind == 4:
            f = np.array([], dtype=np.int64)
        elif kind == 5:
            f = np.arange(600 + i, 600 + i + 130)                                   # a run: unaligned window
        else:
            f = rng.choice(all_ids, size=L, replace=False)                          # exactly searchSize seeds
        filters.append(set(int(v) for v in f))
    bm = vamana.FilterBitmaps.from_sets(filters, align=1 if not dense else 64)
    g_ids, g_d, g_c, tr = ix.search_batch(q, k, L, filters=bm, trace=True, visit_cap=1024)
    l_ids, l_d, l_c, ltr = ix.search_batch(q, k, L, filters=filters, trace=True, visit_cap=1024)
    assert np.array_equal(g_ids, l_ids) and np.array_equal(bits(g_d), bits(l_d)) and np.array_equal(g_c, l_c)
    assert np.array_equal(tr.visit_ids, ltr.visit_ids) and np.array_equal(tr.n_dist, ltr.n_dist)
    for i in range(nq):
        o_ids, o_d, o_vis, o_tr = o.search(q[i], k, L, filter_ids=sorted(filters[i]))
        assert int(g_c[i]) == len(o_ids), i
        assert np.array_equal(g_ids[i, :len(o_ids)], o_ids) and np.array_equal(bits(g_d[i, :len(o_ids)]), bits(o_d)), i
        assert int(tr.n_hop[i]) == o_tr.n_hop and int(tr.n_dist[i]) == o_tr.n_dist, i
        assert np.array_equal(tr.visit_ids[i, :o_tr.n_hop], o_vis), i
    # a window that starts far below the table's first id, and one word per query
    bm2 = vamana.FilterBitmaps(np.zeros(nq, dtype=np.uint64), np.arange(nq + 1, dtype=np.uint64),
                               np.full(nq, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))  # ids 0 .. 63
    b_ids, b_d, b_c, _ = ix.search_batch(q, k, L, filters=bm2)
    w_ids, w_d, w_c, _ = ix.search_batch(q, k, L, filters=[set(range(0, 64))] * nq)
    assert np.array_equal(b_ids, w_ids) and np.array_equal(b_c, w_c) and np.array_equal(bits(b_d), bits(w_d))
    ix.close()
