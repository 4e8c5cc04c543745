"""Searches during a write transaction walk the last committed graph (SURVEY 8b Threading; the reference serves a
reader that cannot get the cache's read lock from a cold index on the committed bucket, shard/cache/manager.go:159-181).
The device keeps two copies of what a walk reads and a write changes; these tests hold the visible behaviour to the
oracle: exactly the pre-transaction answers until commit, exactly the post-transaction answers after."""
import threading
import time

import numpy as np
import pytest

from tests.helpers import bits, start_vector

pytestmark = pytest.mark.gpu


def _rows(rng, n, d, lat):
    x = rng.standard_normal((n, lat.shape[0])).astype(np.float32) @ lat + 0.15 * rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _oracle_answers(o, q, k, L, filters=None):
    out = []
    for i in range(q.shape[0]):
        ids, d, _, _ = o.search(q[i], k, L) if filters is None else o.search(q[i], k, L, filter_ids=sorted(filters[i]))
        out.append((ids, d))
    return out


def _same(got, want):
    g_ids, g_d, g_c = got[:3]
    for i, (ids, d) in enumerate(want):
        if int(g_c[i]) != len(ids) or not np.array_equal(g_ids[i, :len(ids)], ids):
            return False
        if not np.array_equal(bits(g_d[i, :len(ids)]), bits(d)):
            return False
    return True


def test_transaction_is_invisible_until_commit(oracle):
    from semadb_amd import vamana
    rng = np.random.default_rng(2026)
    d, n0, m, R, L, k = 32, 12000, 12000, 32, 50, 10
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base0, base1, q = _rows(rng, n0, d, lat), _rows(rng, m, d, lat), _rows(rng, 64, d, lat)
    sv = start_vector(np.random.default_rng(4), d)
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    ids0 = np.arange(2, n0 + 2, dtype=np.uint64)
    ids1 = np.arange(n0 + 2, n0 + m + 2, dtype=np.uint64)
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=impl)
    o.set_start(sv)
    assert o.insert_rounds(ids0, base0, round_size=0) == 0
    ix = vamana.NewIndexVamana("ver", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(sv)
    ix.insert_batch(ids0, base0)
    assert ix.version_diff() == 0
    pre = _oracle_answers(o, q, k, L)
    assert _same(ix.search_batch(q, k, L), pre)
    filt = [set(int(v) for v in rng.choice(ids0, 30, replace=False)) | {int(ids1[i])} for i in range(64)]
    pre_f = _oracle_answers(o, q, 5, L, filt)

    # ---- an open transaction: 12 000 inserts have run on the device, nothing is visible
    ix.begin_write()
    ix.insert_batch(ids1, base1)
    assert ix.version_diff() > 0
    assert _same(ix.search_batch(q, k, L), pre), "a search saw an uncommitted insert"
    assert _same(ix.search_batch(q, 5, L, filters=filt), pre_f), "a filter resolved an uncommitted id"
    from semadb_amd import flat
    f_ids, _, f_c = flat.flat_search_batch(ix._h, d, q, 3)
    assert int(f_ids.max()) <= int(ids0.max()), "the exact scan saw uncommitted rows"
    ix.commit()
    assert ix.version_diff() == 0
    assert o.insert_rounds(ids1, base1, round_size=0) == 0
    post = _oracle_answers(o, q, k, L)
    assert _same(ix.search_batch(q, k, L), post)
    assert not _same(ix.search_batch(q, k, L), pre)  # the insert did change the answers
    assert _same(ix.search_batch(q, 5, L, filters=filt), _oracle_answers(o, q, 5, L, filt))

    # ---- a delete inside a transaction: the deleted points keep answering until commit
    gone = np.array(sorted(set(int(v) for row in post for v in row[0][:3])), dtype=np.uint64)  # the best answers
    ix.begin_write()
    ix.delete_batch(gone)
    assert ix.version_diff() > 0
    assert _same(ix.search_batch(q, k, L), post), "a search saw an uncommitted delete"
    filt2 = [set(int(v) for v in gone[:20]) for _ in range(64)]
    assert _same(ix.search_batch(q, 5, L, filters=filt2), _oracle_answers(o, q, 5, L, filt2))
    ix.commit()
    assert ix.version_diff() == 0
    assert o.delete(gone) == 0
    after = _oracle_answers(o, q, k, L)
    got = ix.search_batch(q, k, L)
    assert _same(got, after)
    assert not set(int(v) for v in got[0].ravel()) & set(int(v) for v in gone)

    # ---- an update (delete + re-insert), twice for the same ids, inside one transaction: a filter that names the
    # ids still resolves them to their committed rows (found by tools/fuzz_parity.py flat trials, seed 77 trial 7: the
    # record of the committed row was overwritten by the second removal, and a replaced id resolved to its new,
    # uncommitted row)
    upd = np.array(sorted(set(int(v) for row in after for v in row[0][:2]))[:40], dtype=np.uint64)
    filt3 = [set(int(v) for v in upd) for _ in range(64)]
    pre_u = _oracle_answers(o, q, 5, L, filt3)
    new1, new2 = _rows(rng, len(upd), d, lat), _rows(rng, len(upd), d, lat)
    ix.begin_write()
    ix.delete_batch(upd)
    ix.insert_batch(upd, new1, round_size=1)
    assert _same(ix.search_batch(q, 5, L, filters=filt3), pre_u), "a filter lost the committed row of an updated id"
    ix.delete_batch(upd)
    ix.insert_batch(upd, new2, round_size=1)
    assert _same(ix.search_batch(q, 5, L, filters=filt3), pre_u), "a second update hid the committed row"
    assert _same(ix.search_batch(q, k, L), after)
    ix.commit()
    assert o.delete(upd) == 0
    for i in range(len(upd)):
        assert o.insert(int(upd[i]), new1[i]) == 0
    assert o.delete(upd) == 0
    for i in range(len(upd)):
        assert o.insert(int(upd[i]), new2[i]) == 0
    assert _same(ix.search_batch(q, k, L), _oracle_answers(o, q, k, L))
    assert _same(ix.search_batch(q, 5, L, filters=filt3), _oracle_answers(o, q, 5, L, filt3))
    ix.close()


def test_searches_run_while_a_writer_inserts(oracle):
    """a writer thread inserts (one call = one transaction) while this thread keeps searching: every answer is the
    pre-insert oracle answer or the post-insert one, never a mix; at least one search ran during the write"""
    from semadb_amd import vamana
    rng = np.random.default_rng(77)
    d, n0, m, R, L, k = 48, 8000, 40000, 32, 50, 10
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base0, base1, q = _rows(rng, n0, d, lat), _rows(rng, m, d, lat), _rows(rng, 128, d, lat)
    sv = start_vector(np.random.default_rng(5), d)
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    ids0 = np.arange(2, n0 + 2, dtype=np.uint64)
    ids1 = np.arange(n0 + 2, n0 + m + 2, dtype=np.uint64)
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=impl)
    o.set_start(sv)
    assert o.insert_rounds(ids0, base0, round_size=0) == 0
    pre = _oracle_answers(o, q, k, L)
    assert o.insert_rounds(ids1, base1, round_size=0) == 0
    post = _oracle_answers(o, q, k, L)
    ix = vamana.NewIndexVamana("conc", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(sv)
    ix.insert_batch(ids0, base0)
    state = {"done": False, "err": None}

    def writer():
        try:
            ix.insert_batch(ids1, base1)
        except Exception as e:  # pragma: no cover
            state["err"] = e
        state["done"] = True

    t = threading.Thread(target=writer)
    seen_pre = seen_post = during = 0
    t.start()
    while True:
        finished_before = state["done"]
        got = ix.search_batch(q, k, L)
        if not state["done"]:
            during += 1
        is_pre, is_post = _same(got, pre), _same(got, post)
        assert is_pre or is_post, "a search returned neither the committed nor the new version's answers"
        seen_pre += is_pre
        seen_post += is_post
        assert not (finished_before and is_pre and not is_post), "stale answers after the commit"
        if finished_before:
            break
    t.join()
    assert state["err"] is None
    assert during >= 1 and seen_pre >= 1 and seen_post >= 1
    assert ix.version_diff() == 0
    ix.close()


def test_random_writes_under_concurrent_readers(oracle):
    """One writer applies 120 random write batches (inserts in rounds, deletes, updates, now and then a compaction)
    while three reader threads keep asking: graph walks with and without filters, the exact scan, K1.  Every answer a
    reader gets must be the answer of ONE committed version -- the oracle replays the same batches one by one and the
    answers to the fixed questions after each commit are the allowed set -- and nothing may crash, hang or tear.  At
    the end the device graph equals the oracle's edge for edge."""
    from semadb_amd import flat, vamana
    from tests.helpers import assert_same_graph
    rng = np.random.default_rng(909)
    d, n0, R, L, k = 24, 3000, 24, 40, 5
    lat = rng.standard_normal((6, d)).astype(np.float32)
    base0 = _rows(rng, n0, d, lat)
    q = _rows(rng, 16, d, lat)
    sv = start_vector(np.random.default_rng(5), d)
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    ids0 = np.arange(2, n0 + 2, dtype=np.uint64)
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=impl)
    o.set_start(sv)
    for i in range(n0):
        assert o.insert(int(ids0[i]), base0[i]) == 0
    ix = vamana.NewIndexVamana("rw", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(sv)
    ix.insert_batch(ids0, base0, round_size=1)
    filt = [set(int(v) for v in rng.choice(ids0, 25, replace=False)) for _ in range(16)]

    def o_version():
        plain = tuple((tuple(int(v) for v in a), tuple(int(v) for v in bits(b))) for a, b in _oracle_answers(o, q, k, L))
        filtered = tuple((tuple(int(v) for v in a), tuple(int(v) for v in bits(b))) for a, b in _oracle_answers(o, q, 3, L, filt))
        return plain, filtered

    # ---- the write batches, replayed on the oracle first: the answers of every committed version
    batches, versions = [], [o_version()]
    live = [int(v) for v in ids0]
    nxt = n0 + 2
    for b in range(120):
        n_ins, n_del, n_upd = int(rng.integers(0, 30)), int(rng.integers(0, 12)), int(rng.integers(0, 4))
        dels = [int(v) for v in rng.choice(live, n_del, replace=False)] if n_del else []
        rest = [v for v in live if v not in set(dels)]
        upds = [int(v) for v in rng.choice(rest, n_upd, replace=False)] if n_upd else []
        new_ids = list(range(nxt, nxt + n_ins))
        nxt += n_ins
        nv, uv = _rows(rng, max(1, n_ins), d, lat), _rows(rng, max(1, n_upd), d, lat)
        ch = [vamana.IndexVectorChange(i, nv[j]) for j, i in enumerate(new_ids)]
        ch += [vamana.IndexVectorChange(i, None) for i in dels]
        ch += [vamana.IndexVectorChange(i, uv[j]) for j, i in enumerate(upds)]
        batches.append((ch, b % 9 == 8))
        for j, i in enumerate(new_ids):
            assert o.insert(i, nv[j]) == 0
        if dels or upds:
            assert o.delete(np.array(dels + upds, dtype=np.uint64)) == 0
        for j, i in enumerate(upds):
            assert o.insert(i, uv[j]) == 0
        live = sorted((set(live) - set(dels)) | set(new_ids))
        versions.append(o_version())
    plain_ok, filt_ok = set(v[0] for v in versions), set(v[1] for v in versions)

    state = {"done": False, "err": None, "asked": 0}

    def writer():
        try:
            for ch, compact in batches:
                ix.InsertUpdateDelete(ch, round_size=1)
                if compact:
                    ix.compact()
        except Exception as e:  # pragma: no cover
            state["err"] = e
        state["done"] = True

    def reader(kind):
        try:
            while not state["done"]:
                if kind == 0:
                    g = ix.search_batch(q, k, L)
                    got = tuple((tuple(int(v) for v in g[0][i, :int(g[2][i])]), tuple(int(v) for v in bits(g[1][i, :int(g[2][i])])))
                                for i in range(16))
                    assert got in plain_ok, "a walk answered with no committed version's answers"
                elif kind == 1:
                    g = ix.search_batch(q, 3, L, filters=filt)
                    got = tuple((tuple(int(v) for v in g[0][i, :int(g[2][i])]), tuple(int(v) for v in bits(g[1][i, :int(g[2][i])])))
                                for i in range(16))
                    assert got in filt_ok, "a filtered walk answered with no committed version's answers"
                else:
                    f_ids, f_d, f_c = flat.flat_search_batch(ix._h, d, q, 4)
                    assert (np.asarray(f_c) == 4).all() and np.isfinite(np.asarray(f_d)).all()
                    assert (np.diff(np.asarray(f_d), axis=1) >= 0).all(), "the exact scan tore"
                    ix.distance_batch(q, np.tile(ids0[:8], (16, 1)))
                state["asked"] += 1
        except Exception as e:
            state["err"] = e
            state["done"] = True

    threads = [threading.Thread(target=writer)] + [threading.Thread(target=reader, args=(kind,)) for kind in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "a thread hung"
    if state["err"] is not None:
        raise state["err"]
    print("questions answered while the writer ran: %d" % state["asked"])
    assert state["asked"] >= 10
    assert ix.version_diff() == 0
    assert_same_graph(ix, o)
    ix.close()


def test_abort_write_rolls_the_transaction_back(oracle):
    """sdb_index_abort_write: inserts, deletes and updates of an open transaction are undone -- the exported graph, the
    id tables and every answer are what they were at begin_write -- and the writer's state is restored well enough that
    the SAME writes, applied again and committed, build exactly the oracle's graph (the restored rows have no clean
    prefix and no cached edge distances: both only ever save work)."""
    from semadb_amd import flat, vamana
    rng = np.random.default_rng(77)
    d, n0, R, L, k = 32, 6000, 32, 50, 10
    lat = rng.standard_normal((8, d)).astype(np.float32)
    base0, extra, q = _rows(rng, n0, d, lat), _rows(rng, 900, d, lat), _rows(rng, 48, d, lat)
    sv = start_vector(np.random.default_rng(5), d)
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    ids0 = np.arange(2, n0 + 2, dtype=np.uint64)
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=impl)
    o.set_start(sv)
    assert o.insert_rounds(ids0, base0, round_size=0) == 0
    ix = vamana.NewIndexVamana("rb", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(sv)
    ix.insert_batch(ids0, base0)
    snap = ix.export()
    stats0, usage0 = ix.stats(), ix.row_usage()
    pre = ix.search_batch(q, k, L, trace=True, visit_cap=512)
    assert _same(pre, _oracle_answers(o, q, k, L))
    new_ids = np.arange(n0 + 2, n0 + 2 + 600, dtype=np.uint64)
    gone = rng.choice(ids0, 150, replace=False).astype(np.uint64)
    upd = rng.choice(np.setdiff1d(ids0, gone), 40, replace=False).astype(np.uint64)

    def writes(target, commit):
        target.begin_write()
        target.insert_batch(new_ids, extra[:600])
        target.delete_batch(np.concatenate([gone, new_ids[:50]]))
        target.delete_batch(upd)
        target.insert_batch(upd, extra[600:640], round_size=1)
        if commit:
            target.commit()

    for attempt in range(2):  # twice: an aborted transaction leaves nothing behind that a second one trips over
        writes(ix, commit=False)
        assert ix.version_diff() > 0 and ix.exists(int(new_ids[100])) and not ix.exists(int(gone[0]))
        assert _same(ix.search_batch(q, k, L), _oracle_answers(o, q, k, L))  # invisible while open
        assert ix.abort_write() is True
        assert ix.version_diff() == 0
        assert ix.stats() == stats0 and ix.row_usage() == usage0
        assert not ix.exists(int(new_ids[100])) and ix.exists(int(gone[0])) and ix.exists(int(upd[0]))
        again = ix.export()
        for a, b in zip(snap, again):
            assert np.array_equal(a, b)
        got = ix.search_batch(q, k, L, trace=True, visit_cap=512)
        assert np.array_equal(got[0], pre[0]) and np.array_equal(bits(got[1]), bits(pre[1]))
        assert np.array_equal(got[3].visit_ids, pre[3].visit_ids) and np.array_equal(got[3].n_dist, pre[3].n_dist)
        f = [set(int(v) for v in new_ids[:30]) | set(int(v) for v in gone[:10]) for _ in range(48)]
        assert _same(ix.search_batch(q, 5, L, filters=f), _oracle_answers(o, q, 5, L, f))
    assert ix.abort_write() is True  # nothing open: fine
    # the same writes, for real, on both sides
    writes(ix, commit=True)
    assert o.insert_rounds(new_ids, extra[:600], round_size=0) == 0
    assert o.delete(np.concatenate([gone, new_ids[:50]])) == 0
    assert o.delete(upd) == 0
    for i in range(len(upd)):
        assert o.insert(int(upd[i]), extra[600 + i]) == 0
    from tests.helpers import assert_same_graph
    assert_same_graph(ix, o)
    assert _same(ix.search_batch(q, k, L), _oracle_answers(o, q, k, L))
    ix.close()
    # a flat index: Set (insert and replace) and Delete rolled back the same way
    fx = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "euclidean"))
    fx.set_vectors(ids0[:500], base0[:500])
    want = flat.flat_search_batch(fx._h, d, q, 5)
    fx.begin_write()
    fx.set_vectors(np.concatenate([ids0[:20], new_ids[:20]]), extra[:40])  # 20 replaced, 20 new
    fx.remove_vectors(ids0[100:140])
    assert fx.abort_write() is True
    assert fx.row_usage() == (500, 0)
    got = flat.flat_search_batch(fx._h, d, q, 5)
    assert np.array_equal(np.asarray(got[0]), np.asarray(want[0])) and np.array_equal(bits(np.asarray(got[1])), bits(np.asarray(want[1])))
    fx.set_vectors(new_ids[:5], extra[:5])  # and it takes the next write
    assert fx.row_usage() == (505, 0)
    fx.close()
