"""Non-finite distances through the graph walk and the exact scan (GPU) against the oracle / the reference's loops.

The reference has no guard against NaN or Inf: a query with a NaN component makes every distance NaN, euclidean
distances on 1e20-scale rows overflow to +Inf (and tie there), `1 - dot` and `-dot` reach -Inf.  What happens then is
decided by the comparisons of DistSet.AddWithLimit (`distance > last` rejects, `<` bubbles: both FALSE for NaN,
distset.go:184,196), of greedySearch and of IndexFlat.Search (`dist >= tail` skips, `<` bubbles, flat.go:104,121).
The device walk must take the same branches: same result ids, same visit order, same counters, the same distance bits
(a NaN where the reference has a NaN: the payload of a NaN is the one thing the two machines do not share).
K1 alone was covered before (tests/test_gpu_distance.py::test_adversarial_values)."""
import numpy as np
import pytest

from tests.helpers import build_oracle_index, unit_rows

pytestmark = pytest.mark.gpu


def same_bits_or_both_nan(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb]))


def _gpu_index(o, d, metric, R, L):
    from semadb_amd import vamana
    ids, vecs, offsets, edges = o.export()
    ix = vamana.NewIndexVamana("nf", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    ix.load(ids, vecs, offsets, edges)
    return ix


def _check_walk(o, ix, queries, limit, L, visit_cap=2048):
    for mode in (1, 2):  # one wave per query; the workgroup-per-query walk of small calls
        ix.set_tuning("wide_walk", mode)
        out = _check_walk_mode(o, ix, queries, limit, L, visit_cap)
    ix.set_tuning("wide_walk", 0)
    return out


def _check_walk_mode(o, ix, queries, limit, L, visit_cap):
    g_ids, g_d, g_c, tr = ix.search_batch(queries, limit, L, trace=True, visit_cap=visit_cap)
    seen_nan = seen_inf = 0
    for q in range(queries.shape[0]):
        o_ids, o_d, o_vis, o_tr = o.search(queries[q], limit, L)
        n = len(o_ids)
        assert int(g_c[q]) == n, "query %d count" % q
        assert np.array_equal(g_ids[q, :n], o_ids), "query %d ids" % q
        assert same_bits_or_both_nan(g_d[q, :n], o_d), "query %d distances" % q
        assert int(tr.n_hop[q]) == o_tr.n_hop and int(tr.n_dist[q]) == o_tr.n_dist and int(tr.n_edges[q]) == o_tr.n_edges
        assert np.array_equal(tr.visit_ids[q, :o_tr.n_hop], o_vis), "query %d visit order" % q
        seen_nan += int(np.isnan(o_d).sum())
        seen_inf += int(np.isinf(o_d).sum())
    return seen_nan, seen_inf


def _poisoned_queries(rng, d, nq, scale):
    """queries that produce NaN (a NaN or an Inf component), +-Inf (1e20-scale) and ordinary distances"""
    q = (unit_rows(rng, nq, d) * np.float32(scale)).astype(np.float32)
    q[0, rng.integers(0, d)] = np.nan                 # every distance NaN
    q[1, :] = np.nan
    q[2, rng.integers(0, d)] = np.inf                 # Inf - x = Inf, Inf * x = +-Inf; sums of both signs -> NaN
    q[3, rng.integers(0, d)] = -np.inf
    q[4] = (q[4] * np.float32(1e20)).astype(np.float32)   # euclidean: every distance overflows to +Inf -> all ties
    q[5] = (q[5] * np.float32(3e19)).astype(np.float32)   # some overflow, some do not
    q[6, 0], q[6, 1] = np.inf, -np.inf
    q[7] = np.float32(0)                              # a plain one in between
    return q


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d,n", [(96, 700), (384, 600), (33, 500), (1024, 200), (2112, 100)])
def test_walk_with_nonfinite_queries(oracle, metric, d, n):
    """a sane graph, poisoned queries: the walk's candidate array fills with NaN / Inf and every comparison of
    AddWithLimit and of the hop selection goes the reference's way"""
    rng = np.random.default_rng(d * 7 + n + len(metric))
    base = unit_rows(rng, n, d)
    o = build_oracle_index(oracle, base, metric, R=32, L=50)
    ix = _gpu_index(o, d, metric, 32, 50)
    q = _poisoned_queries(rng, d, 12, 1.0)
    nan, inf = _check_walk(o, ix, q, 10, 50)
    assert nan > 0 and (inf > 0 or metric != "euclidean")
    _check_walk(o, ix, q, 50, 50)
    _check_walk(o, ix, q, 5, 25)
    ix.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
def test_walk_over_rows_that_overflow(oracle, metric):
    """poisoned ROWS: a third of the stored vectors sit at the 1e19 / 1e20 scale, a few hold an Inf or a NaN component.
    The oracle builds its graph over them (robustPrune on Inf / NaN distances included); the device walks that graph
    with ordinary and with large queries: Inf ties, NaN candidates next to finite ones, multi-accept hops whose points
    tie at +Inf."""
    rng = np.random.default_rng(1234 + len(metric))
    d, n = 64, 900
    base = unit_rows(rng, n, d)
    big = rng.choice(n, n // 3, replace=False)
    base[big[: n // 6]] *= np.float32(1e20)
    base[big[n // 6:]] *= np.float32(2e19)
    for r in rng.choice(n, 6, replace=False):
        base[r, rng.integers(0, d)] = np.inf
    for r in rng.choice(n, 6, replace=False):
        base[r, rng.integers(0, d)] = np.nan
    o = build_oracle_index(oracle, base, metric, R=24, L=40)
    ix = _gpu_index(o, d, metric, 24, 40)
    q = np.concatenate([unit_rows(rng, 8, d), (unit_rows(rng, 8, d) * np.float32(1e19)).astype(np.float32),
                        base[big[:4]], _poisoned_queries(rng, d, 8, 1.0)])
    nan, inf = _check_walk(o, ix, q, 10, 40)
    assert nan > 0 and inf > 0
    _check_walk(o, ix, q, 40, 40)
    _check_walk(o, ix, q, 10, 130, visit_cap=4096)  # the 512-entry candidate array
    ix.close()


def test_device_build_over_rows_that_overflow(oracle):
    """the sequential device build (round_size = 1) over poisoned rows makes the oracle's graph edge for edge:
    robustPrune's `alpha * d(i, j) < d(j)` with Inf and NaN on either side (search.go:132)"""
    from semadb_amd import vamana
    from tests.helpers import assert_same_graph, start_vector
    rng = np.random.default_rng(99)
    d, n = 32, 400
    base = unit_rows(rng, n, d)
    base[rng.choice(n, 60, replace=False)] *= np.float32(1e20)
    base[rng.choice(n, 40, replace=False)] *= np.float32(3e19)
    base[7, 3] = np.nan
    base[11, 5] = np.inf
    for metric in ("euclidean", "cosine"):
        o = build_oracle_index(oracle, base, metric, R=16, L=30)
        ix = vamana.NewIndexVamana("nb", vamana.IndexVectorVamanaParameters(d, metric, 30, 16, 1.2), strict=False)
        ix.set_start(start_vector(np.random.default_rng(20250622), d))
        ix.insert_batch(None, base, round_size=1)
        assert_same_graph(ix, o)
        ix.close()


def _flat_reference(dist, ids, k):
    """IndexFlat.Search's loop, flat.go:91-126, over one query's distances in storage order"""
    res = []
    for j in range(dist.shape[0]):
        dj = dist[j]
        if len(res) == k and dj >= res[-1][0]:  # False for NaN on either side
            continue
        if len(res) < k:
            res.append((dj, int(ids[j])))
        else:
            res[-1] = (dj, int(ids[j]))
        i = len(res) - 1
        while i > 0 and res[i][0] < res[i - 1][0]:
            res[i], res[i - 1] = res[i - 1], res[i]
            i -= 1
    return np.array([r[1] for r in res], dtype=np.uint64), np.array([r[0] for r in res], dtype=np.float32)


@pytest.mark.parametrize("metric,d,n", [("euclidean", 64, 2000), ("cosine", 64, 2000), ("dot", 96, 1500),
                                        ("euclidean", 64, 34000), ("cosine", 64, 34000)])
def test_exact_scan_with_inf_distances(oracle, metric, d, n):
    """the exact scan (block path below 32 768 rows, streaming scans above) with +-Inf distances: rows at the 1e20
    scale, queries at the 1e19 scale.  Inf compares like any number -- `dist >= tail` skips an Inf that meets an Inf
    tail, first seen stays -- so the reference's loop is still a stable sort, and the device must agree bit for bit."""
    from semadb_amd import flat
    rng = np.random.default_rng(d + n + len(metric))
    # euclidean: (x - y)^2 overflows whatever the signs.  cosine / dot: products of both signs would overflow to +Inf
    # AND -Inf and sum to NaN (the next test); with components of one sign the sums stay at +Inf, distances at -Inf
    sign = (lambda x: x) if metric == "euclidean" else np.abs
    base = sign(unit_rows(rng, n, d))
    base[rng.choice(n, n // 4, replace=False)] *= np.float32(1e20)
    base[rng.choice(n, n // 8, replace=False)] *= np.float32(3e19 if metric != "euclidean" else -3e19)
    ids = np.arange(3, n + 3, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
    ix.set_vectors(ids, base)
    q = np.concatenate([sign(unit_rows(rng, 5, d)), (sign(unit_rows(rng, 5, d)) * np.float32(1e19)).astype(np.float32),
                        (sign(unit_rows(rng, 3, d)) * np.float32(1e20)).astype(np.float32)])
    dm = oracle.distance_matrix(q, base, metric, oracle.IMPL_ASM)
    assert np.isinf(dm).any() and not np.isnan(dm).any()
    for k in (1, 10, 75):
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i in range(q.shape[0]):
            e_ids, e_d = _flat_reference(dm[i], ids, k)
            assert int(g_c[i]) == len(e_ids)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert same_bits_or_both_nan(g_d[i, :len(e_ids)], e_d)
    ix.close()


@pytest.mark.parametrize("metric,d,n", [("euclidean", 48, 1500), ("cosine", 64, 1200), ("euclidean", 64, 34000),
                                        ("dot", 64, 33500)])
def test_exact_scan_with_nan_distances(oracle, metric, d, n):
    """NaN distances in the exact scan.  The reference's loop is NOT a sort then: `dist >= tail` is false for a NaN on
    either side, so a NaN replaces the tail of a full list (evicting a finite answer) and any later row replaces a NaN
    tail, in storage order (flat.go:104-123).  The device reproduces that list, whatever it is worth."""
    from semadb_amd import flat
    rng = np.random.default_rng(d + n + 5)
    base = unit_rows(rng, n, d)
    for r in rng.choice(n, max(6, n // 200), replace=False):
        base[r, rng.integers(0, d)] = np.nan
    for r in rng.choice(n, 5, replace=False):
        base[r, rng.integers(0, d)] = np.inf       # Inf rows: NaN under cosine / dot when signs mix, Inf otherwise
    ids = np.arange(3, n + 3, dtype=np.uint64)
    ix = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, metric), capacity=n + 1)
    ix.set_vectors(ids, base)
    q = np.concatenate([unit_rows(rng, 6, d), _poisoned_queries(rng, d, 8, 1.0)])
    dm = oracle.distance_matrix(q, base, metric, oracle.IMPL_ASM)
    assert np.isnan(dm).any()
    for k in (1, 10, 75):
        g_ids, g_d, g_c = ix.search_batch(q, k)
        for i in range(q.shape[0]):
            e_ids, e_d = _flat_reference(dm[i], ids, k)
            assert int(g_c[i]) == len(e_ids), (k, i)
            assert np.array_equal(g_ids[i, :len(e_ids)], e_ids), (k, i)
            assert same_bits_or_both_nan(g_d[i, :len(e_ids)], e_d)
    ix.close()
