"""The shard exchange behind the C ABI (csrc/cluster.hip: sdb_cluster_*).  RCCL is driven with a single rank -- all a
1-GPU box can hold; RCCL refuses two ranks on one device.  The whole N > 1 protocol -- tickets, tags, a failing shard,
several host batches in flight per rank, the merge over several shards -- runs on one GPU over the library's
shared-device transport (sdb_cluster_create_local with the same device for every rank), which differs from the RCCL
path in the gather call alone.  The torch.distributed form of the exchange is covered in tests/test_cluster.py."""
import threading
import ctypes as C

import numpy as np
import pytest


def test_block_layout_is_what_the_header_says():
    from semadb_amd import cluster
    off_d, off_c, off_t, total = cluster.block_layout(1024, 10)
    assert off_d == 1024 * 10 * 8 and off_c == off_d + 1024 * 10 * 4
    assert off_t >= off_c + 1024 * 4 and off_t % 16 == 0 and total == off_t + 64
    # odd sizes still put the next shard's uint64 ids on an 8-byte boundary, and the tag on a 16-byte one
    for nq, per in [(1, 1), (3, 7), (5, 11), (1023, 13)]:
        a, b, tg, t = cluster.block_layout(nq, per)
        assert a == nq * per * 8 and b == a + nq * per * 4 and tg % 16 == 0 and tg >= b + nq * 4 and t == tg + 64


def test_cluster_calls_fail_loudly_without_a_gpu():
    from semadb_amd import _lib, cluster
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.SemaDBError):
        cluster.Cluster.create(0, 1, bytes(128), 0)
    with pytest.raises(_lib.SemaDBError):
        cluster.Cluster.create_local([0])


def _index(rng, n=3000, d=48, R=32, L=50):
    from semadb_amd import vamana
    from tests.helpers import start_vector, unit_rows
    lat = rng.standard_normal((8, d)).astype(np.float32)
    x = rng.standard_normal((n, 8)).astype(np.float32) @ lat + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    base = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    ix = vamana.NewIndexVamana("cl", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), strict=False)
    ix.set_start(start_vector(rng, d))
    ix.insert_batch(None, base)
    q = unit_rows(rng, 96, d)
    return ix, base, q


@pytest.mark.gpu
@pytest.mark.parametrize("local", [False, True])
def test_cluster_single_rank_end_to_end(oracle, local):
    """unique id -> communicator -> search_batch (search into the ring block, ncclAllGather on the exchange stream,
    merge) equals the direct search: with one shard the merge is the identity (actions.go:357: no sort)."""
    import torch
    from semadb_amd import cluster
    rng = np.random.default_rng(5)
    ix, base, q = _index(rng)
    if local:
        cl = cluster.Cluster.create_local([0])[0]
    else:
        uid = cluster.Cluster.unique_id()
        assert len(uid) == 128
        cl = cluster.Cluster.create(0, 1, uid, 0)
    assert (cl.rank, cl.world) == (0, 1)
    want_ids, want_d, want_c, _ = ix.search_batch(q, 10, 50)
    # host memory: synchronous
    ids, d, sh, c = cl.search_batch(ix, q, 10, 50)
    assert np.array_equal(ids, want_ids) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(c, want_c) and not sh.any()
    # device memory: several batches in flight (more than the ring holds), joined once
    qd = torch.from_numpy(q).cuda()
    outs = [cl.search_batch(ix, qd, 10, 50) for _ in range(7)]
    cl.wait()
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o[0].cpu().numpy().view(np.uint64), want_ids)
        assert np.array_equal(o[1].cpu().numpy().view(np.uint32), want_d.view(np.uint32))
        assert np.array_equal(o[3].cpu().numpy().view(np.uint32), want_c)
    # the primitive on a caller-built block, against the oracle's merge rule
    blk = cluster.PackedTopK(q.shape[0], 10, "cuda:0")
    ix.search_batch(qd, 10, 50, out=blk.out())
    m = cl.allgather_merge(blk, 7)
    cl.synchronize()
    m_ids = m[0].cpu().numpy().view(np.uint64)
    for i in range(q.shape[0]):
        w_ids, w_d, w_s = oracle.cluster_merge(want_ids[None, i, :], want_d[None, i, :], want_c[None, i].astype(np.int32), 7)
        assert np.array_equal(m_ids[i, :len(w_ids)], w_ids)
    h = cl.allgather_merge(blk, 7, host_out=True)
    assert np.array_equal(h[0], m_ids)
    # searchSize < limit is the reference's error (search.go:23-25), checked against the query's own limit
    from semadb_amd._lib import SemaDBError
    with pytest.raises(SemaDBError):
        cl.search_batch(ix, q, 60, 50)
    cl.close()
    ix.close()


@pytest.mark.gpu
def test_exists_batch_tuning_and_build_stats():
    from semadb_amd import vamana
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(9)
    ix, base, q = _index(rng, n=1500)
    e = ix.exists_batch([1, 2, 1501, 1502, 0, 99999])
    assert e.tolist() == [True, True, True, False, False, False]
    ix.delete_batch(np.array([2], dtype=np.uint64))
    assert ix.exists_batch([2, 3]).tolist() == [False, True] and not ix.exists(2)
    assert ix.row_usage() == (1501, 1) and ix.stats()[0] == 1500  # the tombstone still occupies a row
    # GetMany (plain.go:26-45): request order, missing ids skipped, bit-exact rows; d = 48 has a tail chain region
    want_ids = [7, 10 ** 9, 3, 1500, 2, 1]
    vecs, found = ix.GetMany(want_ids)
    assert found.tolist() == [True, False, True, True, False, True] and vecs.shape == (4, 48)
    assert np.array_equal(vecs[0].view(np.uint32), base[7 - 2].view(np.uint32))
    assert np.array_equal(vecs[1].view(np.uint32), base[3 - 2].view(np.uint32))
    assert np.array_equal(vecs[2].view(np.uint32), base[1500 - 2].view(np.uint32))
    st = ix.build_stats()
    assert st["rounds"] > 0 and st["search_n_dist"] > 1500 and st["prune_pairs"] > 0
    assert st["requests"] > 0 and st["requests"] >= st["appends"]
    with pytest.raises(SemaDBError):
        ix.set_tuning("hub_min", 1)
    with pytest.raises(SemaDBError):
        ix.set_tuning("hash_limit", 10 ** 6)
    ix.close()


@pytest.mark.gpu
def test_host_batcher_answers_equal_direct_calls():
    """semadb_host.hpp's SearchBatcher through libsemadb_hostbench.so: 16 threads x 8 outstanding single-query
    requests coalesce into device batches and every answer equals the direct batch call."""
    import os
    from semadb_amd import _lib
    rng = np.random.default_rng(12)
    ix, base, q = _index(rng)
    so = os.path.join(os.path.dirname(_lib.SO_PATH), "libsemadb_hostbench.so")
    hb = C.CDLL(so)
    hb.sdb_hostbench_batcher.restype = C.c_int
    hb.sdb_hostbench_batcher.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    nq, d, k = q.shape[0], q.shape[1], 10
    first_ids = np.zeros((nq, k), dtype=np.uint64)
    first_c = np.zeros(nq, dtype=np.uint32)
    qps, nb, served = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    p50, p99 = C.c_double(0), C.c_double(0)
    rc = hb.sdb_hostbench_batcher(ix._h, d, q.ctypes.data, nq, k, 50, 16, 8, 64, 500, 2, 0.3, first_ids.ctypes.data,
                                  first_c.ctypes.data, C.byref(qps), C.byref(nb), C.byref(served), C.byref(p50), C.byref(p99))
    assert rc == 0 and qps.value > 0 and served.value >= nq and 0 < p50.value <= p99.value
    assert served.value / nb.value > 4, "requests were not coalesced"
    want_ids, _, want_c, _ = ix.search_batch(q, k, 50)
    assert np.array_equal(first_c, want_c) and np.array_equal(first_ids, want_ids)
    ix.close()


# ---------------------------------------------------------------------------------------------------------------
# N > 1 on one GPU: the shared-device transport
# ---------------------------------------------------------------------------------------------------------------
def _shards(rng, world, n=1500, d=48, nq=64):
    from semadb_amd import vamana
    from tests.helpers import start_vector, unit_rows
    lat = rng.standard_normal((8, d)).astype(np.float32)
    ixs, bases = [], []
    for s in range(world):
        x = rng.standard_normal((n, 8)).astype(np.float32) @ lat + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
        base = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
        ix = vamana.NewIndexVamana("sh%d" % s, vamana.IndexVectorVamanaParameters(d, "cosine", 50, 32, 1.2), strict=False)
        ix.set_start(start_vector(rng, d))
        ix.insert_batch(None, base)
        ixs.append(ix)
        bases.append(base)
    qs = [unit_rows(rng, nq, d) for _ in range(6)]
    return ixs, bases, qs


def _expected(oracle, ixs, q, limit, L):
    """the reference's rule on the shards' own answers: per-shard limit (actions.go:291-299), then the merge (:357-376)"""
    from semadb_amd import cluster
    per = cluster.shard_limit(limit, len(ixs), 75)
    res = [ix.search_batch(q, per, L) for ix in ixs]
    ids = np.stack([r[0] for r in res]); d = np.stack([r[1] for r in res]); c = np.stack([r[2] for r in res])
    out = []
    for i in range(q.shape[0]):
        out.append(oracle.cluster_merge(ids[:, i, :], d[:, i, :], c[:, i].astype(np.int32), limit))
    return out


def _check(got, want):
    ids, d, sh, c = got
    for i, (w_ids, w_d, w_s) in enumerate(want):
        n = len(w_ids)
        assert int(c[i]) == n
        assert np.array_equal(ids[i, :n], w_ids) and np.array_equal(d[i, :n].view(np.uint32), w_d.view(np.uint32))
        assert np.array_equal(sh[i, :n].astype(np.int32), w_s)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_shards_on_one_gpu_fanout_equals_reference_merge(oracle, world):
    """ClusterNode.SearchPoints over `world` shards that share the GPU: requests from racing threads, one ticket each,
    every rank's blocking host-memory call on a thread of its own -- every answer is the reference's merge of the
    shards' own answers, whatever order the threads ran in."""
    from semadb_amd import cluster
    rng = np.random.default_rng(40 + world)
    ixs, bases, qs = _shards(rng, world)
    ranks = cluster.Cluster.create_local([0] * world)
    assert [(r.rank, r.world) for r in ranks] == [(i, world) for i in range(world)]
    assert "shared-device: %d ranks" % world in ranks[0].transport()
    assert cluster.shard_limit(10, world, 75) == min(10, int(10 / world * 1.42 + 10))  # actions.go:291-299: 10 at 8 shards
    fan = cluster.Fanout(ranks, ixs)
    want = [_expected(oracle, ixs, q, 10, 50) for q in qs]
    results, errors = {}, []

    def client(j):
        try:
            for rep in range(3):
                b = (j + rep) % len(qs)
                results[(j, rep)] = (b, fan.search_points(qs[b], 10, 50))
        except Exception as e:  # pragma: no cover - reported below
            errors.append(e)

    ts = [threading.Thread(target=client, args=(j,)) for j in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(results) == 18
    seen = set()
    for b, got in results.values():
        _check(got, want[b])
        seen.update(int(v) for i in range(got[0].shape[0]) for v in got[2][i, :int(got[3][i])])
    assert sorted(seen) == list(range(world))  # every shard's block reached the merge
    assert ranks[0].next_ticket() == 19 and ranks[-1].next_ticket() == 19
    for r in ranks:
        r.close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_misordered_collectives_are_refused_not_merged(oracle):
    """The race the reference's fan-out allows (requests A and B reach shard 0 as A,B and shard 1 as B,A) with the
    ordering switched off (ticket 0): the all-gathers pair A's block with B's.  The tags disagree on the query hash,
    so NO rank returns an answer -- every call fails with SDB_ERR_STATE -- and the handles stay in step: the next,
    ordered request is answered."""
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(77)
    ixs, bases, qs = _shards(rng, 2)
    ranks = cluster.Cluster.create_local([0, 0])
    A, B = qs[0], qs[1]
    errs = {}

    def run(r, order):
        for name, q in order:
            try:
                ranks[r].search_batch(ixs[r], q, 10, 50)
                errs[(r, name)] = None
            except SemaDBError as e:
                errs[(r, name)] = e

    t0 = threading.Thread(target=run, args=(0, [("A", A), ("B", B)]))
    t1 = threading.Thread(target=run, args=(1, [("B", B), ("A", A)]))
    t0.start(); t1.start(); t0.join(); t1.join()
    assert len(errs) == 4
    for key, e in errs.items():
        assert e is not None and e.code == 3, key  # SDB_ERR_STATE
        assert "query hash" in str(e) and "different requests" in str(e)
    # in step again: an ordered request goes through
    fan = cluster.Fanout(ranks, ixs)
    _check(fan.search_points(A, 10, 50), _expected(oracle, ixs, A, 10, 50))
    # the same race with tickets cannot happen: whichever thread arrives first, ticket order wins
    t = ranks[0].next_ticket()
    outs = {}

    def run_t(r, order):
        for name, q, tk in order:
            outs[(r, name)] = ranks[r].search_batch(ixs[r], q, 10, 50, ticket=tk)

    t0 = threading.Thread(target=run_t, args=(0, [("A", A, t), ("B", B, t + 1)]))
    t1 = threading.Thread(target=run_t, args=(1, [("B", B, t + 1)]))
    t2 = threading.Thread(target=run_t, args=(1, [("A", A, t)]))
    t1.start()  # rank 1's B arrives first and must wait for rank 1's A
    import time
    time.sleep(0.2)
    t0.start(); t2.start()
    for th in (t0, t1, t2):
        th.join()
    wa, wb = _expected(oracle, ixs, A, 10, 50), _expected(oracle, ixs, B, 10, 50)
    for r in (0, 1):
        _check(outs[(r, "A")], wa)
        _check(outs[(r, "B")], wb)
    # a ticket that has already entered is refused
    with pytest.raises(SemaDBError):
        ranks[0].search_batch(ixs[0], A, 10, 50, ticket=t)
    for r in ranks:
        r.close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_failing_shard_enters_the_exchange_and_fails_the_request_everywhere(oracle):
    """One shard cannot search (no start node: sdb_index_search_batch returns SDB_ERR_STATE).  Its rank still joins the
    all-gather with the status in its tag, so the other rank is not left waiting inside the collective; both calls
    return an error for THIS request (actions.go:339-353) and the next request, on healthy shards, is served."""
    from semadb_amd import cluster, vamana
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(91)
    ixs, bases, qs = _shards(rng, 2)
    bad = vamana.NewIndexVamana("bad", vamana.IndexVectorVamanaParameters(48, "cosine", 50, 32, 1.2), strict=False)
    ranks = cluster.Cluster.create_local([0, 0])
    fan = cluster.Fanout(ranks, [ixs[0], bad])
    with pytest.raises(SemaDBError) as ei:
        fan.search_points(qs[0], 10, 50)
    assert ei.value.code == 3
    # each rank on its own: the healthy one names the failing shard, the failing one reports its own error
    out = {}

    def run(r, ix, tk):
        try:
            ranks[r].search_batch(ix, qs[0], 10, 50, ticket=tk)
            out[r] = None
        except SemaDBError as e:
            out[r] = e

    tk = ranks[0].next_ticket()
    ths = [threading.Thread(target=run, args=(0, ixs[0], tk)), threading.Thread(target=run, args=(1, bad, tk))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert out[0] is not None and "shard 1" in str(out[0]) and "status 3" in str(out[0])
    assert out[1] is not None and "start point" in str(out[1])
    good = cluster.Fanout(ranks, ixs)
    _check(good.search_points(qs[1], 10, 50), _expected(oracle, ixs, qs[1], 10, 50))
    for r in ranks:
        r.close()
    for ix in ixs + [bad]:
        ix.close()


@pytest.mark.gpu
def test_device_memory_exchanges_in_flight_and_their_verdicts(oracle):
    """Asynchronous (device-memory) calls over two shards driven from ONE thread, several batches in flight; the
    verdict of a mis-paired exchange arrives with synchronize(), the counts of that request are zero."""
    import torch
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(123)
    ixs, bases, qs = _shards(rng, 2)
    ranks = cluster.Cluster.create_local([0, 0])
    qd = [torch.from_numpy(q).cuda() for q in qs]
    outs = []
    for b in range(5):
        outs.append([ranks[r].search_batch(ixs[r], qd[b], 10, 50) for r in range(2)])
    for r in ranks:
        r.synchronize()
    for b in range(5):
        want = _expected(oracle, ixs, qs[b], 10, 50)
        for r in range(2):
            o = outs[b][r]
            _check((o[0].cpu().numpy().view(np.uint64), o[1].cpu().numpy(), o[2].cpu().numpy().view(np.uint32),
                    o[3].cpu().numpy().view(np.uint32)), want)
    # mis-paired: rank 0 searches batch 0, rank 1 batch 1
    o0 = ranks[0].search_batch(ixs[0], qd[0], 10, 50)
    o1 = ranks[1].search_batch(ixs[1], qd[1], 10, 50)
    for r in ranks:
        with pytest.raises(SemaDBError) as ei:
            r.synchronize()
        assert ei.value.code == 3 and "query hash" in str(ei.value)
    assert not o0[3].cpu().numpy().any() and not o1[3].cpu().numpy().any()
    for r in ranks:
        r.synchronize()  # reported once
    # the caller's own block through allgather_merge, two shards
    per = cluster.shard_limit(10, 2, 75)
    blks = []
    for r in range(2):
        blk = cluster.PackedTopK(qs[2].shape[0], per, "cuda:0")
        ixs[r].search_batch(qd[2], per, 50, out=blk.out())
        blks.append(blk)
    ms = [ranks[r].allgather_merge(blks[r], 10) for r in range(2)]
    for r in ranks:
        r.synchronize()
    want = _expected(oracle, ixs, qs[2], 10, 50)
    for m in ms:
        _check((m[0].cpu().numpy().view(np.uint64), m[1].cpu().numpy(), m[2].cpu().numpy().view(np.uint32),
                m[3].cpu().numpy().view(np.uint32)), want)
    for r in ranks:
        r.close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_mixed_device_lists_are_rejected():
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError, device_count
    if device_count() < 2:
        with pytest.raises(SemaDBError):
            cluster.Cluster.create_local([0, 1])  # device 1 does not exist here
    with pytest.raises(SemaDBError):
        cluster.Cluster.create_local([0, 0, 1] if device_count() >= 2 else [0, 0, 7])


# ---------------------------------------------------------------------------------------------------------------
# a way out of the turnstile (sdb_cluster_set_deadline / sdb_cluster_skip_ticket): the reference fails one request
# and serves the next (cluster/actions.go:339-353)
# ---------------------------------------------------------------------------------------------------------------
def _call(rank, ix, q, ticket, out, key, want=True):
    from semadb_amd._lib import SemaDBError
    try:
        out[key] = rank.search_batch(ix, q, 10, 50, ticket=ticket, want=want)
    except SemaDBError as e:
        out[key] = e


@pytest.mark.gpu
def test_a_ticket_that_is_never_presented_does_not_wedge_the_ranks(oracle):
    """Ticket t is drawn and presented to NO rank (the request died in the host).  The next request waits at the
    turnstile for the deadline, fails with SDB_ERR_STATE having done nothing, the fan-out skips the lost ticket, and
    the SAME request (same ticket) is then answered -- as is the one after it."""
    import time
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(311)
    ixs, bases, qs = _shards(rng, 2)
    ranks = cluster.Cluster.create_local([0, 0])
    for r in ranks:
        r.set_deadline(300)
    fan = cluster.Fanout(ranks, ixs)
    _check(fan.search_points(qs[0], 10, 50), _expected(oracle, ixs, qs[0], 10, 50))  # ticket 1
    lost = fan._ticket()  # ticket 2: never presented
    nxt = fan._ticket()  # ticket 3
    out = {}
    t0 = time.time()
    ths = [threading.Thread(target=_call, args=(ranks[r], ixs[r], qs[1], nxt, out, r)) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert 0.25 < time.time() - t0 < 5
    for r in range(2):
        assert isinstance(out[r], SemaDBError) and out[r].code == 3, out[r]
        assert "ticket %d" % lost in str(out[r]) and "did nothing" in str(out[r])
        assert ranks[r].next_ticket() == lost  # nothing moved
    for r in ranks:
        r.skip_ticket(lost)
        assert r.next_ticket() == nxt
    with pytest.raises(SemaDBError):  # a skipped ticket cannot come back
        ranks[0].search_batch(ixs[0], qs[1], 10, 50, ticket=lost)
    ths = [threading.Thread(target=_call, args=(ranks[r], ixs[r], qs[1], nxt, out, r)) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    want = _expected(oracle, ixs, qs[1], 10, 50)
    for r in range(2):
        _check(out[r], want)
    # a ticket skipped BEFORE its turn: the turnstile passes over it when it gets there
    a, b, c = fan._ticket(), fan._ticket(), fan._ticket()
    for r in ranks:
        r.skip_ticket(b)
    for tk, q in ((a, qs[2]), (c, qs[3])):
        ths = [threading.Thread(target=_call, args=(ranks[r], ixs[r], q, tk, out, r)) for r in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for r in range(2):
            _check(out[r], _expected(oracle, ixs, q, 10, 50))
    assert ranks[0].next_ticket() == c + 1 == ranks[1].next_ticket()
    for r in ranks:
        r.close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_a_ticket_dropped_on_one_rank_fails_that_request_and_the_next_is_served(oracle):
    """The reference's failure model (actions.go:339-353): the request reaches shard 0 and never shard 1 (its goroutine
    died).  (a) The fan-out notices and skips the ticket on rank 1 WITH the request's shape: rank 1 enters the exchange
    with an empty answer under an error flag, rank 0's call fails naming shard 1, and the next request is served.
    (b) Nobody notices: rank 0's call gives up after the deadline, the request is withdrawn, the handles stay in
    step, and once rank 1's turnstile has been told (skip, nq = 0) the next request is served."""
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(312)
    ixs, bases, qs = _shards(rng, 2)
    ranks = cluster.Cluster.create_local([0, 0])
    for r in ranks:
        r.set_deadline(400)
    fan = cluster.Fanout(ranks, ixs)
    # (a)
    tk = fan._ticket()
    out = {}
    th = threading.Thread(target=_call, args=(ranks[0], ixs[0], qs[0], tk, out, 0))
    th.start()
    ranks[1].skip_ticket(tk, nq=qs[0].shape[0], limit=10)
    th.join()
    assert isinstance(out[0], SemaDBError) and out[0].code == 3
    assert "shard 1" in str(out[0]) and "failed" in str(out[0])
    _check(fan.search_points(qs[1], 10, 50), _expected(oracle, ixs, qs[1], 10, 50))
    # (b)
    tk = fan._ticket()
    _call(ranks[0], ixs[0], qs[2], tk, out, 0)
    assert isinstance(out[0], SemaDBError) and out[0].code == 3 and "did not join" in str(out[0])
    assert "out of step" not in str(out[0])
    ranks[1].skip_ticket(tk)
    _check(fan.search_points(qs[3], 10, 50), _expected(oracle, ixs, qs[3], 10, 50))
    _check(fan.search_points(qs[4], 10, 50), _expected(oracle, ixs, qs[4], 10, 50))
    for r in ranks:
        r.close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_destroying_a_rank_fails_its_peers_pending_exchanges(oracle):
    """a rank of a shared-device group goes away while its peer waits for it inside an exchange: the peer's call fails
    (no answer, SDB_ERR_STATE) instead of waiting for ever, and later calls on the peer are refused"""
    import time
    from semadb_amd import cluster
    from semadb_amd._lib import SemaDBError
    rng = np.random.default_rng(313)
    ixs, bases, qs = _shards(rng, 2)
    ranks = cluster.Cluster.create_local([0, 0])
    out = {}
    th = threading.Thread(target=_call, args=(ranks[0], ixs[0], qs[0], 1, out, 0))
    th.start()
    time.sleep(0.3)
    ranks[1].close()
    th.join(timeout=20)
    assert not th.is_alive()
    assert isinstance(out[0], SemaDBError) and out[0].code == 3 and "shard 1" in str(out[0])
    with pytest.raises(SemaDBError) as ei:
        ranks[0].search_batch(ixs[0], qs[0], 10, 50, ticket=2)
    assert "destroyed" in str(ei.value)
    ranks[0].close()
    for ix in ixs:
        ix.close()


@pytest.mark.gpu
def test_rccl_transport_names_the_library_and_the_communicator():
    from semadb_amd import cluster
    cl = cluster.Cluster.create_local([0])[0]
    t = cl.transport()
    assert t.startswith("rccl ") and "librccl" in t and "communicator of 1 ranks" in t and "rank 0" in t
    cl.close()
