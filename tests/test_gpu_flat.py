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


class _Store:
    """the storage-order model of the device store: Set of a stored id drops its row and appends the new one, Delete
    drops the row, compaction keeps the order"""

    def __init__(self):
        self.ids, self.rows = [], []

    def set(self, i, v):
        self.delete(i)
        self.ids.append(int(i)), self.rows.append(np.asarray(v, dtype=np.float32))

    def delete(self, i):
        if int(i) in self.ids:
            k = self.ids.index(int(i))
            del self.ids[k], self.rows[k]

    def arrays(self):
        return np.array(self.ids, dtype=np.uint64), np.stack(self.rows)


def _check_flat(oracle, ix, store, q, metric, ks=(1, 10)):
    ids, base = store.arrays()
    for k in ks:
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, k)):
            assert int(g_c[i]) == len(e_ids)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))


@pytest.mark.parametrize("metric,d,n", [("euclidean", 2, 400), ("cosine", 33, 900), ("dot", 128, 36000)])
def test_flat_insert_update_delete(oracle, metric, d, n):
    """flat.go:41-74 on the device path: vecStore.Set replaces the vector of a stored id, vecStore.Delete removes the
    point, a delete of a missing id does nothing, the same id may appear more than once in one call (the last one
    wins).  After every call the exact scan equals the oracle's over the points that are left -- for the table of
    36 000 rows through the streaming scan, whose kernel skips tombstones -- and after compaction as well."""
    from semadb_amd import flat
    rng = np.random.default_rng(n)
    mk = (lambda m: unit_rows(rng, m, d)) if d > 2 else (lambda m: rng.integers(0, 6, size=(m, d)).astype(np.float32))
    base = mk(n)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric))
    st = _Store()
    ix.InsertUpdateDelete([flat.IndexVectorChange(i + 3, base[i]) for i in range(n)])
    for i in range(n):
        st.set(i + 3, base[i])
    q = np.concatenate([base[:6] + np.float32(0.01), mk(5)])
    _check_flat(oracle, ix, st, q, metric)
    # one mixed call: updates (among them the queries' nearest rows), deletes, a missing id, new points,
    # a point set twice and a point deleted and set again
    upd = mk(40)
    new = mk(30)
    ch = [flat.IndexVectorChange(3 + i, upd[i]) for i in range(20)]
    ch += [flat.IndexVectorChange(40 + i, None) for i in range(25)] + [flat.IndexVectorChange(10 ** 7, None)]
    ch += [flat.IndexVectorChange(n + 100 + i, new[i]) for i in range(30)]
    ch += [flat.IndexVectorChange(3, upd[20]), flat.IndexVectorChange(41, upd[21])]
    ch += [flat.IndexVectorChange(n + 100, None), flat.IndexVectorChange(n + 100, upd[22])]
    ix.InsertUpdateDelete(ch)
    for c in ch:
        st.delete(c.Id) if c.Vector is None else st.set(c.Id, c.Vector)
    assert ix.version_diff() == 0
    _check_flat(oracle, ix, st, q, metric)
    rows, dead = ix.row_usage()
    assert rows - dead == len(st.ids) and dead == 20 + 25 + 1 + 1  # updates, deletes, the second Set of 3, the deleted new point
    # a filter naming a deleted and an updated id
    allowed = [{45, 3, 4, 200, n + 100} for _ in range(q.shape[0])]
    g_ids, g_d, g_c = ix.search_batch(q, 10, filters=allowed)
    ids, rows_ = st.arrays()
    for i, (e_ids, e_d) in enumerate(_expected(oracle, q, rows_, ids, metric, 10, allowed)):
        assert int(g_c[i]) == len(e_ids) == 4
        assert np.array_equal(g_ids[i, :4], e_ids) and np.array_equal(bits(g_d[i, :4]), bits(e_d))
    ix.compact()
    assert ix.row_usage() == (len(st.ids), 0) and ix.version_diff() == 0
    _check_flat(oracle, ix, st, q, metric)
    # the store keeps working after compaction
    ix.InsertUpdateDelete([flat.IndexVectorChange(5, upd[23]), flat.IndexVectorChange(6, None)])
    st.set(5, upd[23]), st.delete(6)
    _check_flat(oracle, ix, st, q, metric)
    ix.close()


def test_flat_writes_are_invisible_until_commit(oracle):
    """a flat search between begin_write and commit answers from the committed rows"""
    from semadb_amd import flat
    rng = np.random.default_rng(2)
    d, n = 48, 500
    base = unit_rows(rng, n, d)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "cosine"))
    st = _Store()
    ix.set_vectors(np.arange(1, n + 1, dtype=np.uint64), base)
    for i in range(n):
        st.set(i + 1, base[i])
    q = base[:8] + np.float32(0.01)
    ix.begin_write()
    ix.set_vectors(np.arange(1, 5, dtype=np.uint64), unit_rows(rng, 4, d))
    ix.remove_vectors([5, 6, 7])
    ix.set_vectors(np.array([9000], dtype=np.uint64), q[:1])
    _check_flat(oracle, ix, st, q, "cosine")  # nothing of it is visible
    ix.commit()
    g_ids, g_d, _ = ix.search_batch(q, 1)
    assert int(g_ids[0, 0]) == 9000 and not ({int(v) for v in g_ids[:, 0]} & {5, 6, 7})
    assert ix.version_diff() == 0
    ix.close()


def test_flat_set_rules():
    from semadb_amd import flat
    from semadb_amd._lib import SemaDBError
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(4, "euclidean"))
    v = np.ones((2, 4), dtype=np.float32)
    with pytest.raises(SemaDBError):
        ix.set_vectors(np.array([0, 1], dtype=np.uint64), v)  # id 0 is no point
    with pytest.raises(SemaDBError):
        ix.set_vectors(np.array([7, 7], dtype=np.uint64), v)  # twice in one call: which one wins is the caller's order
    ix.set_vectors(np.array([7, 8], dtype=np.uint64), v)
    ix.remove_vectors([7, 7, 99])  # twice and unknown: one tombstone
    assert ix.row_usage() == (2, 1)
    ix.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_matrix_core_scan_equals_the_packed_fma_scan(oracle, metric):
    """dot / cosine tables are scanned on the matrix cores (k_flat_scan_mfma: one v_mfma_f32_16x16x1 = one fused
    multiply-add of the reference's chain for 1 024 (row, query, partial sum) triples).  Every row length the kernel is
    instantiated for (1..32 blocks of 32 floats) gives the same ids and the same distance bits as the scan without
    them (SDB_TUNE_NO_MFMA: the packed-FMA kernel up to 19 blocks, the block path beyond), values that underflow to
    denormals included; five of the lengths are also held to the oracle directly."""
    import ctypes as C
    from semadb_amd import flat
    from semadb_amd._lib import lib, check
    rng = np.random.default_rng(77)
    n, nq = 33000, 21
    for nblk in range(1, 33):
        d = 32 * nblk
        base = rng.standard_normal((n, d)).astype(np.float32)
        base[::7] *= np.float32(1e-22)   # products of two such values are denormal or underflow
        base[5::11, ::3] = 0.0
        if metric == "cosine":
            base[1::2] = base[1::2] / np.linalg.norm(base[1::2], axis=1, keepdims=True)
        q = base[rng.choice(n, nq, replace=False)] * np.float32(1.0 + 1e-3)
        q[3] *= np.float32(1e-20)
        ids = np.arange(3, n + 3, dtype=np.uint64)
        ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
        ix.set_vectors(ids, base)
        got = ix.search_batch(q, 10)
        check(lib().sdb_index_set_tuning(ix._h, 5, 1))  # SDB_TUNE_NO_MFMA
        ref = ix.search_batch(q, 10)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])), d
        assert np.array_equal(got[2], ref[2])
        if nblk in (1, 12, 16, 24, 32):
            for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, 10)):
                assert np.array_equal(got[0][i], e_ids) and np.array_equal(bits(got[1][i]), bits(e_d)), (d, i)
        ix.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_matrix_core_scan_with_a_tail(oracle, metric):
    """rows of d % 32 != 0: the tail elements are one more chain per pair (dot.s:35-43), run as a ninth accumulator
    set with one matrix instruction per tail element.  Same ids and distance bits as the block path (SDB_TUNE_NO_MFMA)
    for tails of 1, 18, 4, 12, 15, 8 and 31 elements, denormal products and all-zero rows included, and as the oracle."""
    from semadb_amd import flat
    from semadb_amd._lib import lib, check
    rng = np.random.default_rng(78)
    n, nq = 33100, 19
    for d in (33, 50, 100, 300, 527, 1000, 1055):
        base = rng.standard_normal((n, d)).astype(np.float32)
        base[::7] *= np.float32(1e-22)
        base[5::11, ::3] = 0.0
        base[9::500] = 0.0  # all-zero rows: distances of exactly zero, whose sign the extra chain must not change
        if metric == "cosine":
            nz = np.linalg.norm(base, axis=1) > 0
            base[nz] = base[nz] / np.linalg.norm(base[nz], axis=1, keepdims=True)
        q = base[rng.choice(n, nq, replace=False)] * np.float32(1.0 + 1e-3)
        q[3] *= np.float32(1e-20)
        q[4] = 0.0
        ids = np.arange(3, n + 3, dtype=np.uint64)
        ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
        ix.set_vectors(ids, base)
        got = ix.search_batch(q, 10)
        check(lib().sdb_index_set_tuning(ix._h, 5, 1))  # SDB_TUNE_NO_MFMA
        ref = ix.search_batch(q, 10)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])), d
        if d in (33, 300, 1055):
            for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, metric, 10)):
                assert np.array_equal(got[0][i], e_ids) and np.array_equal(bits(got[1][i]), bits(e_d)), (d, i)
        ix.close()


def test_flat_filter_that_names_no_stored_id(oracle):
    """a query whose filter holds only ids that are not stored has an empty candidate list: no answer for it, the other
    queries unaffected -- also when it is the last query, or when every query is like that (found by
    tools/fuzz_parity.py, seed 31337 trial 233: the distance kernel computed a dropped candidate on the slot word that
    follows the list, i.e. on whatever lies behind the buffer)"""
    from semadb_amd import flat
    rng = np.random.default_rng(12)
    n, d = 700, 2
    base = rng.standard_normal((n, d)).astype(np.float32)
    ids = rng.permutation(np.arange(1, 3 * n + 1))[:n].astype(np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "dot"))
    ix.set_vectors(ids, base)
    q = rng.standard_normal((6, d)).astype(np.float32)
    some = set(int(v) for v in ids[:25])
    for filters in ([some, {10 ** 9 + 1}, some, set(), some, {10 ** 9 + 1, 10 ** 9 + 2}],
                    [{10 ** 9 + 1}] * 6):
        g_ids, g_d, g_c = ix.search_batch(q, 10, filters=filters)
        for i, (e_ids, e_d) in enumerate(_expected(oracle, q, base, ids, "dot", 10, filters)):
            assert int(g_c[i]) == len(e_ids) == (10 if filters[i] is some else 0)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids) and np.array_equal(bits(g_d[i, :len(e_ids)]), bits(e_d))
    ix.close()


def test_flat_writes_under_a_concurrent_reader(oracle):
    """a writer thread applies 150 random Set / replace / Delete batches (and compactions) to a flat index while a reader
    thread keeps scanning: every answer is the exact answer over ONE committed state of the store"""
    import threading
    from semadb_amd import flat
    rng = np.random.default_rng(4)
    d, n0 = 24, 2500
    base = unit_rows(rng, n0, d)
    q = unit_rows(rng, 12, d)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "cosine"))
    st = _Store()
    ids0 = np.arange(1, n0 + 1, dtype=np.uint64)
    ix.set_vectors(ids0, base)
    for i in range(n0):
        st.set(i + 1, base[i])

    def answers():
        ids, rows = st.arrays()
        return tuple((tuple(int(v) for v in a), tuple(int(v) for v in bits(b))) for a, b in _expected(oracle, q, rows, ids, "cosine", 5))

    batches, allowed = [], {answers()}
    nxt = n0 + 1
    for b in range(150):
        ch = []
        for _ in range(int(rng.integers(1, 12))):
            op = int(rng.integers(0, 3))
            if op == 0:
                ch.append(flat.IndexVectorChange(nxt, unit_rows(rng, 1, d)[0]))
                nxt += 1
            elif op == 1:
                ch.append(flat.IndexVectorChange(int(rng.choice(st.ids)), unit_rows(rng, 1, d)[0]))
            else:
                ch.append(flat.IndexVectorChange(int(rng.choice(st.ids)), None))
        batches.append((ch, b % 20 == 19))
        for c in ch:
            st.delete(c.Id) if c.Vector is None else st.set(c.Id, c.Vector)
        allowed.add(answers())
    state = {"done": False, "err": None, "asked": 0}

    def writer():
        try:
            for ch, compact in batches:
                ix.InsertUpdateDelete(ch)
                if compact:
                    ix.compact()
        except Exception as e:  # pragma: no cover
            state["err"] = e
        state["done"] = True

    def reader():
        try:
            while not state["done"]:
                g = ix.search_batch(q, 5)
                got = tuple((tuple(int(v) for v in g[0][i]), tuple(int(v) for v in bits(g[1][i]))) for i in range(12))
                assert got in allowed, "an exact scan answered with no committed state's answers"
                state["asked"] += 1
        except Exception as e:
            state["err"] = e
            state["done"] = True

    ts = [threading.Thread(target=writer), threading.Thread(target=reader)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
        assert not t.is_alive(), "a thread hung"
    if state["err"] is not None:
        raise state["err"]
    assert state["asked"] >= 5 and ix.version_diff() == 0
    _check_flat(oracle, ix, st, q, "cosine")
    ix.close()
