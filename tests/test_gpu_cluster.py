"""The RCCL exchange behind the C ABI (csrc/cluster.hip: sdb_cluster_*), driven with a single rank -- all a 1-GPU
box can hold; RCCL refuses two ranks on one device -- plus the CPU-side checks of its layout and argument
handling.  The N > 1 logic (shard-major gather, merge rule) is covered under gloo in tests/test_cluster.py."""
import ctypes as C

import numpy as np
import pytest


def test_block_layout_is_what_the_header_says():
    from semadb_amd import cluster
    off_d, off_c, total = cluster.block_layout(1024, 10)
    assert off_d == 1024 * 10 * 8 and off_c == off_d + 1024 * 10 * 4
    assert total >= off_c + 1024 * 4 and total % 16 == 0
    # odd sizes still put the next shard's uint64 ids on an 8-byte boundary
    for nq, per in [(1, 1), (3, 7), (5, 11), (1023, 13)]:
        a, b, t = cluster.block_layout(nq, per)
        assert a == nq * per * 8 and b == a + nq * per * 4 and t % 16 == 0 and t >= b + nq * 4


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
                                         C.POINTER(C.c_uint64)]
    nq, d, k = q.shape[0], q.shape[1], 10
    first_ids = np.zeros((nq, k), dtype=np.uint64)
    first_c = np.zeros(nq, dtype=np.uint32)
    qps, nb, served = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    rc = hb.sdb_hostbench_batcher(ix._h, d, q.ctypes.data, nq, k, 50, 16, 8, 64, 500, 2, 0.3, first_ids.ctypes.data,
                                  first_c.ctypes.data, C.byref(qps), C.byref(nb), C.byref(served))
    assert rc == 0 and qps.value > 0 and served.value >= nq
    assert served.value / nb.value > 4, "requests were not coalesced"
    want_ids, _, want_c, _ = ix.search_batch(q, k, 50)
    assert np.array_equal(first_c, want_c) and np.array_equal(first_ids, want_ids)
    ix.close()
