"""The two-precision hop (SDB_TUNE_SKETCH, search_kernel.h PlainDist::sketch_keep): a float16 copy of the rows is read
first and a neighbour's float32 row only when its float16 distance does not PROVE that AddWithLimit discards it
(distset.go:184).  Everything a search returns -- ids, distance bits, counts, visit order, n_dist / n_hop / n_edges --
must be what the default walk and the oracle return; the audit mode evaluates every discarded neighbour exactly as
well and counts decisions the exact distance contradicts (0)."""
import numpy as np
import pytest

from tests.helpers import bits, build_oracle_index, unit_rows

pytestmark = pytest.mark.gpu


def _gpu_index(o, d, metric, R, L):
    from semadb_amd import vamana
    ids, vecs, offsets, edges = o.export()
    ix = vamana.NewIndexVamana("t", vamana.IndexVectorVamanaParameters(d, metric, L, R, 1.2), strict=False)
    ix.load(ids, vecs, offsets, edges)
    ix.set_tuning("wide_walk", 1)  # the batch walk (one wave per query) is the one that has the stage
    return ix


def _answers(ix, queries, limit, L, visit_cap=512):
    ids, d, c, tr = ix.search_batch(queries, limit, L, trace=True, visit_cap=visit_cap)
    return ids.copy(), bits(d).copy(), c.copy(), tr.n_dist.copy(), tr.n_hop.copy(), tr.n_edges.copy(), tr.visit_ids.copy()


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("metric", ["cosine", "dot", "euclidean"])
@pytest.mark.parametrize("d,n,L", [(96, 1500, 25), (128, 1500, 40), (256, 1200, 30), (384, 1500, 25), (512, 800, 25), (768, 700, 25)])
def test_two_precision_hop_is_the_default_walk_bit_for_bit(oracle, metric, d, n, L):
    rng = np.random.default_rng(d + n)
    lat = rng.standard_normal((12, d)).astype(np.float32)
    base = rng.standard_normal((n, 12)).astype(np.float32) @ lat + 0.2 * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    o = build_oracle_index(oracle, base, metric, R=24, L=L)
    ix = _gpu_index(o, d, metric, 24, L)
    queries = np.vstack([unit_rows(rng, 40, d), base[:8]])
    ref = _answers(ix, queries, 10, L)
    for mode in (2, 1):
        ix.set_tuning("sketch", mode)
        discarded, contradicted, in_use = ix.sketch_stats()
        assert in_use and discarded == 0
        got = _answers(ix, queries, 10, L)
        assert _same(ref, got), "mode %d differs from the default walk" % mode
        discarded, contradicted, _ = ix.sketch_stats()
        assert discarded > 0, "the stage never discarded anything: not exercised"
        assert contradicted == 0
    # ... and the oracle's, for a sample
    for q in range(0, queries.shape[0], 6):
        o_ids, o_d, o_vis, o_tr = o.search(queries[q], 10, L)
        assert np.array_equal(got[0][q, :len(o_ids)], o_ids) and np.array_equal(got[1][q, :len(o_ids)], bits(o_d))
        assert int(got[3][q]) == o_tr.n_dist and int(got[4][q]) == o_tr.n_hop and np.array_equal(got[6][q, :o_tr.n_hop], o_vis)
    ix.set_tuning("sketch", 0)
    assert ix.sketch_stats()[2] is False and _same(ref, _answers(ix, queries, 10, L))
    ix.close()


def test_hostile_rows_discard_nothing_wrongly(oracle):
    """rows the float16 copy cannot hold (overflow -> the bound is infinite: nothing is discarded), rows of very
    different norms (a loose bound), values below the smallest normal half (flushed, measured), a NaN row"""
    d, n, L = 128, 1500, 30
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((48, d)).astype(np.float32)
    variants = {
        "norm spread": base * rng.choice(np.array([1e-3, 1.0, 30.0], np.float32), size=(n, 1)),
        "tiny values": base * np.float32(3e-6),
        "float16 overflow": np.where(np.arange(n)[:, None] == 7, np.float32(1e6), base),
        "a NaN row": np.where(np.arange(n)[:, None] == 11, np.float32(np.nan), base),
    }
    for name, rows in variants.items():
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        for metric in ("dot", "euclidean"):
            o = build_oracle_index(oracle, rows, metric, R=16, L=L)
            ix = _gpu_index(o, d, metric, 16, L)
            ref = _answers(ix, queries, 10, L)
            ix.set_tuning("sketch", 2)
            got = _answers(ix, queries, 10, L)
            discarded, contradicted, in_use = ix.sketch_stats()
            assert in_use and _same(ref, got), (name, metric)
            assert contradicted == 0, (name, metric)
            if name in ("float16 overflow", "a NaN row"):
                assert discarded == 0, (name, metric)  # an unbounded error proves nothing
            ix.close()


def test_copy_follows_the_committed_rows(oracle):
    from semadb_amd import vamana
    d, L = 96, 30
    rng = np.random.default_rng(9)
    base = unit_rows(rng, 2600, d)
    queries = unit_rows(rng, 64, d)
    ix = vamana.NewIndexVamana("t", vamana.IndexVectorVamanaParameters(d, "cosine", L, 24, 1.2), strict=False)
    ix.set_tuning("wide_walk", 1)
    ix.set_start(unit_rows(rng, 1, d)[0])
    ix.set_tuning("sketch", 2)  # before there is a row: the first commit builds the copy
    ix.insert_batch(np.arange(2, 1502, dtype=np.uint64), base[:1500])
    assert ix.sketch_stats()[2]

    def both():
        got = _answers(ix, queries, 10, L)
        ix.set_tuning("sketch", 0)
        ref = _answers(ix, queries, 10, L)
        ix.set_tuning("sketch", 2)
        assert ix.sketch_stats()[2] and _same(ref, got)
        return ref

    a = both()
    ix.insert_batch(np.arange(1502, 2602, dtype=np.uint64), base[1500:2600])  # grows the table: the copy is rebuilt at the new size
    b = both()
    assert not _same(a, b)  # the new rows are found
    # small commits: only the appended rows are converted (the maxima carry on), one point at a time and with an update
    more = unit_rows(rng, 12, d)
    for i in range(10):
        ix.insert_batch(np.array([5000 + i], dtype=np.uint64), more[i:i + 1], round_size=1)
    ix.InsertUpdateDelete([vamana.IndexVectorChange(5003, more[10].tolist()), vamana.IndexVectorChange(5004, None)])
    b = both()
    # an open transaction: searches walk the committed view with float32 rows only; abort: the copy is current again
    ix.begin_write()
    assert ix.sketch_stats()[2] is False
    assert _same(b, _answers(ix, queries, 10, L))
    ix.abort_write()
    assert ix.sketch_stats()[2] and _same(b, _answers(ix, queries, 10, L))
    # deletes (tombstones), then compaction (rows move)
    ix.delete_batch(np.arange(100, 700, dtype=np.uint64))
    c = both()
    ix.compact()
    assert _same(c[:3], both()[:3])  # same answers from the compacted table (slots moved: the counters may differ)
    assert ix.sketch_stats()[1] == 0
    ix.close()


def test_shapes_without_the_stage_are_untouched(oracle):
    rng = np.random.default_rng(3)
    for d, metric in ((100, "euclidean"), (100, "cosine")):  # rows with a tail chain (d % 32 != 0)
        base = unit_rows(rng, 600, d)
        o = build_oracle_index(oracle, base, metric, R=16, L=25)
        ix = _gpu_index(o, d, metric, 16, 25)
        q = unit_rows(rng, 16, d)
        ref = _answers(ix, q, 5, 25)
        ix.set_tuning("sketch", 1)
        assert ix.sketch_stats()[2] is False and _same(ref, _answers(ix, q, 5, 25))
        ix.close()


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
@pytest.mark.parametrize("noise", [1e-3, 1e-4, 1e-5, 0.0])
def test_near_ties_around_the_threshold(oracle, noise, metric):
    """rows in tight clusters: the candidate array's last distance sits inside a crowd of neighbours whose distances
    differ by less than the bound (or not at all) -- the stage must leave every such neighbour to the exact evaluation"""
    d, n, L = 128, 2400, 30
    rng = np.random.default_rng(int(noise * 1e6) + 17)
    centers = unit_rows(rng, 12, d)
    base = centers[rng.integers(0, 12, n)] + np.float32(noise) * rng.standard_normal((n, d)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    o = build_oracle_index(oracle, base, metric, R=24, L=L)
    ix = _gpu_index(o, d, metric, 24, L)
    queries = np.vstack([centers, base[:20], unit_rows(rng, 16, d)])
    ref = _answers(ix, queries, 10, L)
    ix.set_tuning("sketch", 2)
    got = _answers(ix, queries, 10, L)
    discarded, contradicted, in_use = ix.sketch_stats()
    assert in_use and contradicted == 0 and _same(ref, got)
    for q in range(0, queries.shape[0], 5):
        o_ids, o_d, o_vis, o_tr = o.search(queries[q], 10, L)
        assert np.array_equal(got[0][q, :len(o_ids)], o_ids) and np.array_equal(got[1][q, :len(o_ids)], bits(o_d))
        assert int(got[3][q]) == o_tr.n_dist and np.array_equal(got[6][q, :o_tr.n_hop], o_vis)
    ix.close()
