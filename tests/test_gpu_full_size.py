"""The hot path at BASELINE.json's full sizes (C2/C3: 1M x 384 cosine, searchSize 75, degreeBound 64, alpha 1.2,
batch 1024; C4: 10M x 768 + product quantizer K = 256, M = 8), held to properties that do not need the oracle
to walk a graph of that size: graph invariants the reference asserts after its own builds
(shard_vector_test.go:129-245, vamana_test.go:63-75), determinism of the round schedule, sortedness, idempotence,
independence of a query's answer from the batch it travels in, recall against the exact scan -- and, where the
oracle is cheap per element, bit-exact checks against it (every returned distance recomputed by the oracle from
the stored row or code; the walk itself on a sample of queries for C2).

Sizes: SDB_TEST_C2_ROWS / SDB_TEST_C4_ROWS override the row counts (defaults are the BASELINE sizes)."""
import os
import sys
import types

import numpy as np
import pytest

from tests.helpers import bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C2_ROWS = int(os.environ.get("SDB_TEST_C2_ROWS", 1_000_000))
C4_ROWS = int(os.environ.get("SDB_TEST_C4_ROWS", 10_000_000))
R, L, K = 64, 75, 10


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench  # the synthetic data generator of the measured workload (SURVEY 8d seeds)
    return bench


def _build(n, d):
    import torch
    from semadb_amd import vamana
    bench = _bench()
    base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
    ix = vamana.NewIndexVamana("full", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    ix.insert_batch(None, base)  # ids 2..n+1, full-size rounds
    torch.cuda.synchronize()
    return ix, base


@pytest.fixture(scope="module")
def c2():
    import torch
    bench = _bench()
    ix, base = _build(C2_ROWS, 384)
    queries = bench.gen_rows(2048, 384, 20250621, "latent:24", "cuda:0")
    ns = types.SimpleNamespace(ix=ix, base=base, queries=queries, n=C2_ROWS, d=384)
    yield ns
    ix.close()
    del base
    torch.cuda.empty_cache()


def _graph_invariants(ids, off, edges, n):
    """shard_vector_test.go:129-245 checkNodeCount / checkNoReferences / checkConnectivity, vectorised"""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order
    assert np.array_equal(ids, np.arange(1, n + 2, dtype=np.uint64))  # start node 1, then 2..n+1
    deg = np.diff(off.astype(np.int64))
    assert deg.min() >= 1 and deg.max() <= R
    assert int(edges.min()) >= 1 and int(edges.max()) <= n + 1  # every edge names a stored node
    src = np.repeat(np.arange(n + 1, dtype=np.int64), deg)
    dst = edges.astype(np.int64) - 1
    assert not (src == dst).any(), "self loop"
    key = src * (n + 1) + dst
    assert np.unique(key).size == key.size, "duplicate edge in a node's list"
    g = csr_matrix((np.ones(dst.size, np.int8), dst, off.astype(np.int64)), shape=(n + 1, n + 1))
    order = breadth_first_order(g, 0, directed=True, return_predecessors=False)
    assert order.size == n + 1, "%d nodes unreachable from the start node" % (n + 1 - order.size)


def test_c3_build_graph_invariants(c2):
    ids, _, off, edges = c2.ix.export(with_vectors=False)
    _graph_invariants(ids, off, edges, c2.n)
    assert edges.size / (c2.n + 1) > 0.9 * R  # alpha = 1.2 on this data fills nearly every list


def test_c3_build_is_deterministic(c2):
    """the same rows give the same graph, edge for edge (the reference's worker race does not)"""
    _, _, off, edges = c2.ix.export(with_vectors=False)
    ix2, _ = _build(c2.n, c2.d)
    _, _, off2, edges2 = ix2.export(with_vectors=False)
    ix2.close()
    assert np.array_equal(off, off2) and np.array_equal(edges, edges2)


def _search(ix, q, k=K, search_size=L):
    import torch
    ids, d, c, tr = ix.search_batch(q, k, search_size, trace=True)
    torch.cuda.synchronize()
    return (ids.cpu().numpy().view(np.uint64), d.cpu().numpy(), c.cpu().numpy().view(np.uint32),
            tr.n_dist.cpu().numpy().view(np.uint32), tr.n_hop.cpu().numpy().view(np.uint32))


def _row_properties(ids, d, c, n, k=K):
    assert (c == k).all()
    assert (np.diff(d, axis=1) >= 0).all(), "results not ascending by distance"
    assert ids.min() >= 2 and ids.max() <= n + 1, "start node or unknown id returned"
    s = np.sort(ids, axis=1)
    assert (np.diff(s.astype(np.int64), axis=1) > 0).all(), "an id twice in one result"


def test_c2_search_properties(c2, oracle):
    import torch
    q = c2.queries[:1024]
    ids, d, c, nd, nh = _search(c2.ix, q)
    _row_properties(ids, d, c, c2.n)
    # every returned distance is the store's distance to that row: K1 on the device, and the oracle on the host
    k1 = c2.ix.distance_batch(q, ids).cpu().numpy()
    assert np.array_equal(bits(k1), bits(d))
    rows = c2.base[torch.from_numpy((ids.astype(np.int64) - 2).ravel()).cuda()].cpu().numpy().reshape(1024, K, c2.d)
    qh = q.cpu().numpy()
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    for i in range(1024):
        want = oracle.distance_matrix(qh[i:i + 1], rows[i], "cosine", impl)[0]
        assert np.array_equal(bits(want), bits(d[i])), "query %d: distance bits differ from the oracle" % i
    # idempotent
    ids2, d2, c2_, nd2, nh2 = _search(c2.ix, q)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(d), bits(d2))
    assert np.array_equal(nd, nd2) and np.array_equal(nh, nh2)
    # a query's answer does not depend on the batch it travels in: 4 x 256, and the batch reversed
    for s in range(0, 1024, 256):
        a_ids, a_d, _, a_nd, _ = _search(c2.ix, q[s:s + 256])
        assert np.array_equal(a_ids, ids[s:s + 256]) and np.array_equal(bits(a_d), bits(d[s:s + 256]))
        assert np.array_equal(a_nd, nd[s:s + 256])
    r_ids, r_d, _, _, _ = _search(c2.ix, torch.flip(q, dims=[0]).contiguous())
    assert np.array_equal(r_ids[::-1], ids) and np.array_equal(bits(r_d[::-1]), bits(d))
    # a smaller limit is a prefix of a larger one (vamana.go:285-306 truncates the same search set)
    p_ids, p_d, _, _, _ = _search(c2.ix, q[:256], k=3)
    assert np.array_equal(p_ids, ids[:256, :3]) and np.array_equal(bits(p_d), bits(d[:256, :3]))


def test_c2_recall_against_exact_scan(c2):
    import torch
    from semadb_amd import flat
    q = c2.queries[:1024]
    ids, d, _, _, _ = _search(c2.ix, q)
    f_ids, f_d, f_c = flat.flat_search_batch(c2.ix._h, c2.d, q, K, device=0)
    torch.cuda.synchronize()
    f_ids, f_d = f_ids.cpu().numpy().view(np.uint64), f_d.cpu().numpy()
    assert (np.diff(f_d, axis=1) >= 0).all()
    hit = (ids[:, :, None] == f_ids[:, None, :]).any(2)
    assert hit.mean() >= 0.95, "recall@10 %.4f" % hit.mean()  # BASELINE metric's gate
    # the exact scan bounds the walk from below, and where both name the same point the distance bits agree
    assert (f_d[:, 0] <= d[:, 0]).all()
    same = ids[:, 0] == f_ids[:, 0]
    assert np.array_equal(bits(d[same, 0]), bits(f_d[same, 0]))
    # the scan agrees with a plain matmul top-k (ids; fp32 rounding can swap near-ties)
    t_ids = (_bench().exact_topk(q, c2.base, K)[1] + 2).cpu().numpy().astype(np.uint64)
    assert (f_ids[:, :, None] == t_ids[:, None, :]).any(2).mean() > 0.995


def test_c2_self_retrieval(c2):
    """vamana_test.go:230-252 at 1M: a stored vector finds itself first"""
    import torch
    pick = torch.arange(0, c2.n, c2.n // 1024, device="cuda:0")[:1024]
    ids, d, c, _, _ = _search(c2.ix, c2.base[pick].contiguous())
    _row_properties(ids, d, c, c2.n)
    want = (pick.cpu().numpy() + 2).astype(np.uint64)
    assert (ids[:, 0] == want).mean() >= 0.99
    hit = ids[:, 0] == want
    k1 = c2.ix.distance_batch(c2.base[pick].contiguous(), want.reshape(-1, 1)).cpu().numpy()[:, 0]
    assert np.array_equal(bits(d[hit, 0]), bits(k1[hit]))


def test_c2_oracle_walks_the_same_path(c2, oracle):
    """the 1M graph exported to the oracle: result ids, distance bits, visit order and counters of a sample"""
    ids, vecs, off, edges = c2.ix.export()
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    o = oracle.Index(c2.d, "cosine", R, L, 1.2, impl=impl)
    assert o.load(ids, vecs, off, edges) == 0
    del vecs, edges
    q = c2.queries[1024:1024 + 96]
    g_ids, g_d, g_c, tr = c2.ix.search_batch(q, K, L, trace=True, visit_cap=512)
    import torch
    torch.cuda.synchronize()
    g_ids, g_d = g_ids.cpu().numpy().view(np.uint64), g_d.cpu().numpy()
    vis = tr.visit_ids.cpu().numpy().view(np.uint64)
    qh = q.cpu().numpy()
    for i in range(qh.shape[0]):
        o_ids, o_d, o_vis, o_tr = o.search(qh[i], K, L)
        assert np.array_equal(g_ids[i], o_ids) and np.array_equal(bits(g_d[i]), bits(o_d))
        assert int(tr.n_dist[i]) == o_tr.n_dist and int(tr.n_hop[i]) == o_tr.n_hop
        assert np.array_equal(vis[i, :o_tr.n_hop], o_vis), "query %d visit order" % i


def test_c5_two_shards_merge(c2, oracle):
    """C5's shape on one GPU: two 1M-row shards, every shard answers every query (actions.go:316-351), the merged
    top-k equals the oracle's restatement of actions.go:357-376 and recalls the exact top-k of the union"""
    import torch
    from semadb_amd import cluster, flat, vamana
    bench = _bench()
    base1 = bench.gen_rows(c2.n, c2.d, 20250620 + 1, "latent:24", "cuda:0")  # shard r = seed 20250620 + r
    ix1 = vamana.NewIndexVamana("s1", vamana.IndexVectorVamanaParameters(c2.d, "cosine", L, R, 1.2), capacity=c2.n + 1)
    ix1.set_start(bench.start_vector(c2.d))
    ix1.insert_batch(None, base1)
    q = c2.queries[:1024]
    per = cluster.shard_limit(K, 2, 75)
    assert per == oracle.shard_limit(K, 2, 75) == 10
    parts = [ix.search_batch(q, per, L)[:3] for ix in (c2.ix, ix1)]
    g_ids = torch.stack([p[0] for p in parts])
    g_d = torch.stack([p[1] for p in parts])
    g_c = torch.stack([p[2] for p in parts])
    m_ids, m_d, m_s, m_c = cluster.topk_merge(g_ids, g_d, g_c, K)
    torch.cuda.synchronize()
    m_ids, m_d, m_s = m_ids.cpu().numpy().view(np.uint64), m_d.cpu().numpy(), m_s.cpu().numpy()
    h_ids, h_d, h_c = g_ids.cpu().numpy().view(np.uint64), g_d.cpu().numpy(), g_c.cpu().numpy()
    assert (m_c.cpu().numpy() == K).all() and (np.diff(m_d, axis=1) >= 0).all()
    for i in range(1024):
        o_ids, o_d, o_s = oracle.cluster_merge(h_ids[:, i], h_d[:, i], h_c[:, i], K)
        assert np.array_equal(bits(o_d), bits(m_d[i]))
        if np.unique(o_d).size == o_d.size:  # the reference's sort is unstable on ties
            assert np.array_equal(o_ids, m_ids[i]) and np.array_equal(o_s, m_s[i])
    # recall of the merged answer against the exact scan of both shards
    truth = []
    for r, ix in enumerate((c2.ix, ix1)):
        f_ids, f_d, _ = flat.flat_search_batch(ix._h, c2.d, q, K, device=0)
        truth.append((f_d.cpu().numpy(), f_ids.cpu().numpy().view(np.uint64) + (np.uint64(r) << np.uint64(40))))
    td = np.concatenate([t[0] for t in truth], 1)
    ti = np.concatenate([t[1] for t in truth], 1)
    sel = np.argsort(td, axis=1, kind="stable")[:, :K]
    want = np.take_along_axis(ti, sel, 1)
    got = m_ids + (m_s.astype(np.uint64) << np.uint64(40))
    assert (got[:, :, None] == want[:, None, :]).any(2).mean() >= 0.95
    # the same through the library's exchange -- ClusterNode.SearchPoints for the two shards as two ranks that share
    # this GPU (device copies in place of ncclAllGather, everything else the code an 8-GPU node runs): tickets, tagged
    # blocks, tag check + merge, device-memory calls with four batches in flight, then a host-memory fan-out
    ranks = cluster.Cluster.create_local([0, 0])
    outs = []
    for b in range(4):
        qb = c2.queries[b * 1024:(b + 1) * 1024]
        outs.append([ranks[r].search_batch(ix, qb, K, L, ticket=b + 1) for r, ix in enumerate((c2.ix, ix1))])
    for r in ranks:
        r.synchronize()
    e_ids, e_d, e_s, e_c = (t.cpu().numpy() for t in outs[0][0])
    assert np.array_equal(e_ids.view(np.uint64), m_ids) and np.array_equal(bits(e_d), bits(m_d))
    assert np.array_equal(e_s.view(np.uint32), m_s.astype(np.uint32)) and (e_c.view(np.uint32) == K).all()
    for b in range(4):  # both ranks hold the same merged answer
        for j in range(4):
            assert torch.equal(outs[b][0][j], outs[b][1][j])
    fan = cluster.Fanout(ranks, [c2.ix, ix1])
    h = fan.search_points(q.cpu().numpy(), K, L)
    assert np.array_equal(h[0], m_ids) and np.array_equal(bits(h[1]), bits(m_d)) and np.array_equal(h[2], m_s.astype(np.uint32))
    for r in ranks:
        r.close()
    ix1.close()


def test_c4_quantized_walk_matches_oracle_at_1M(oracle):
    """C4's store at a size the oracle can hold in host memory (1M x 768 = 3 GB): graph built on device, quantizer
    fitted on the first 10 000 rows and attached, everything exported to the oracle -- the quantized walk gives
    the same ids, LUT-distance bits, visit order and counters on a sample of queries."""
    import torch
    from semadb_amd import vectorstore as vs
    bench = _bench()
    n, d, M, Kc = int(os.environ.get("SDB_TEST_C4_ORACLE_ROWS", 1_000_000)), 768, 8, 256
    ix, base = _build(n, d)
    train = base[:10000].cpu().numpy().copy()
    first = np.arange(M) * 7 % 10000
    pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(Kc, M, 10000), d)
    pq.Fit(train, first, alias=True)
    vs.attach(ix, pq)
    ids, vecs, off, edges = ix.export()
    del base
    torch.cuda.empty_cache()
    impl = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    o = oracle.Index(d, "cosine", R, L, 1.2, impl=impl)
    assert o.load(ids, vecs, off, edges) == 0
    del vecs, edges
    opq = oracle.PQ(d, "cosine", M, Kc, impl=impl)
    opq.set_codebook(pq.codebook()[0])
    assert o.attach_pq(opq, vs.get_codes(ix, ids)) == 0
    q = bench.gen_rows(64, d, 20250621, "latent:24", "cuda:0")
    g_ids, g_d, g_c, tr = ix.search_batch(q, K, L, trace=True, visit_cap=512)
    torch.cuda.synchronize()
    g_ids, g_d = g_ids.cpu().numpy().view(np.uint64), g_d.cpu().numpy()
    vis = tr.visit_ids.cpu().numpy().view(np.uint64)
    qh = q.cpu().numpy()
    for i in range(64):
        o_ids, o_d, o_vis, o_tr = o.search(qh[i], K, L)
        assert np.array_equal(g_ids[i], o_ids) and np.array_equal(bits(g_d[i]), bits(o_d)), "query %d" % i
        assert int(tr.n_dist[i]) == o_tr.n_dist and int(tr.n_hop[i]) == o_tr.n_hop
        assert np.array_equal(vis[i, :o_tr.n_hop], o_vis), "query %d visit order" % i
    ix.close()
    pq.close()


# ---- C4: 10M x 768 + product quantizer -----------------------------------------------------------

@pytest.fixture(scope="module")
def c4():
    import torch
    from semadb_amd import vectorstore as vs
    bench = _bench()
    torch.cuda.empty_cache()
    ix, base = _build(C4_ROWS, 768)
    queries = bench.gen_rows(1024, 768, 20250621, "latent:24", "cuda:0")
    ns = types.SimpleNamespace(ix=ix, base=base, queries=queries, n=C4_ROWS, d=768, M=8, K=256)
    ns.full = _search(ix, queries)  # full-precision walk, before the quantizer is attached
    train = base[:10000].cpu().numpy().copy()  # TriggerThreshold maximum (models/quantizer.go:62)
    ns.pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(ns.K, ns.M, 10000), 768)
    ns.pq.Fit(train, np.arange(ns.M) * 7 % 10000, alias=True)
    yield ns
    ix.close()
    del base
    torch.cuda.empty_cache()


def test_c4_full_precision_walk(c4):
    import torch
    from semadb_amd import flat
    ids, d, c, nd, nh = c4.full
    _row_properties(ids, d, c, c4.n)
    f_ids = flat.flat_search_batch(c4.ix._h, c4.d, c4.queries, K, device=0)[0]
    torch.cuda.synchronize()
    hit = (ids[:, :, None] == f_ids.cpu().numpy().view(np.uint64)[:, None, :]).any(2)
    assert hit.mean() >= 0.95, "recall@10 %.4f" % hit.mean()
    n_nodes, n_edges, _ = c4.ix.stats()
    assert n_nodes == c4.n + 1 and n_edges <= n_nodes * R


def test_c4_quantized_search(c4, oracle):
    from semadb_amd import vectorstore as vs
    vs.attach(c4.ix, c4.pq)  # K6: encodes all rows
    fc, _ = c4.pq.codebook()
    opq = oracle.PQ(c4.d, "cosine", c4.M, c4.K)
    opq.set_codebook(fc)
    # codes of a spread of rows equal the oracle's encode (product.go:136-159)
    import torch
    pick = np.arange(0, c4.n, c4.n // 512)[:512]
    codes = vs.get_codes(c4.ix, (pick + 2).astype(np.uint64))
    rows = c4.base[torch.from_numpy(pick).cuda()].cpu().numpy()
    for j in range(pick.size):
        assert np.array_equal(codes[j], opq.encode(rows[j])), "row %d" % pick[j]
    # the quantized walk: sorted, unique, idempotent, batch independent
    ids, d, c, nd, nh = _search(c4.ix, c4.queries)
    _row_properties(ids, d, c, c4.n)
    ids2, d2, _, nd2, _ = _search(c4.ix, c4.queries)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(d), bits(d2)) and np.array_equal(nd, nd2)
    a_ids, a_d, _, _, _ = _search(c4.ix, c4.queries[512:768])
    assert np.array_equal(a_ids, ids[512:768]) and np.array_equal(bits(a_d), bits(d[512:768]))
    # every returned distance is the oracle's LUT sum over that point's code (product.go:255-275), bit for bit
    qh = c4.queries.cpu().numpy()
    rc = vs.get_codes(c4.ix, ids[:128].ravel()).reshape(128, K, c4.M)
    for i in range(128):
        lut = opq.lut(qh[i])
        want = np.array([opq.dist_lut(lut, rc[i, j]) for j in range(K)], dtype=np.float32)
        assert np.array_equal(bits(want), bits(d[i])), "query %d" % i
    # the quantized walk does far fewer HBM bytes per query than the full-precision one, on the same graph
    assert nd.mean() * c4.M < 0.05 * c4.full[3].mean() * c4.d * 4


C5_ROWS = int(os.environ.get("SDB_TEST_C5_ROWS", 12_500_000))


def test_c5_one_rank_at_size():
    """BASELINE configs[4] is 8 shards x 12.5M x 384, one per MI355X; a 1-GPU box holds one of them.  One rank's shard
    at full size (19.2 GB slab; seed 20250620 + rank): built on the device, graph invariants checked where
    the graph lies (degree bound, no dangling / self edges: export without vectors), 1 024 queries: rows sorted, ids
    unique, every returned distance bit-equal to K1's recomputation from the stored row, recall@10 >= 0.95 against the
    exact scan of the same 12.5M rows (k_flat_scan), per-shard limit of the 8-shard cluster rule applied."""
    import torch
    from semadb_amd import cluster, flat, vamana
    bench = _bench()
    rank, d, n = 3, 384, C5_ROWS
    base = bench.gen_rows(n, d, 20250620 + rank, "latent:24", "cuda:0")
    queries = bench.gen_rows(1024, d, 20250621, "latent:24", "cuda:0")
    ix = vamana.NewIndexVamana("c5", vamana.IndexVectorVamanaParameters(d, "cosine", L, R, 1.2), capacity=n + 1)
    ix.set_start(bench.start_vector(d))
    ix.insert_batch(None, base)
    torch.cuda.synchronize()
    del base
    torch.cuda.empty_cache()
    nn, ne, mx = ix.stats()
    assert nn == n + 1 and mx == n + 1 and ne / nn > 0.9 * R
    ids, _, off, edges = ix.export(with_vectors=False)
    deg = np.diff(off.astype(np.int64))
    assert deg.min() >= 1 and deg.max() <= R and int(edges.min()) >= 1 and int(edges.max()) <= n + 1
    src = np.repeat(np.arange(n + 1, dtype=np.int64), deg)
    assert not (src == edges.astype(np.int64) - 1).any(), "self loop"
    del ids, off, edges, src, deg
    per_shard = cluster.shard_limit(K, 8)  # actions.go:291-299 for the 8-shard collection
    g_ids, g_d, g_c, _ = ix.search_batch(queries, per_shard, L)
    f_ids, f_d, f_c = flat.flat_search_batch(ix._h, d, queries, per_shard)
    torch.cuda.synchronize()
    gi, gd = g_ids.cpu().numpy().view(np.uint64), g_d.cpu().numpy()
    assert (g_c.cpu().numpy() == per_shard).all()
    assert (np.diff(gd, axis=1) >= 0).all(), "rows not sorted by distance"
    assert all(len(set(row)) == per_shard for row in gi[:128])
    hits = (g_ids.to(torch.int64).unsqueeze(2) == f_ids.to(torch.int64).unsqueeze(1)).any(2).float().mean().item()
    assert hits >= 0.95, hits
    # every returned distance is what K1 computes from the stored row (bit for bit)
    k1 = ix.distance_batch(queries[:64], gi[:64])
    torch.cuda.synchronize()
    assert np.array_equal(bits(k1.cpu().numpy()), bits(gd[:64]))
    # and the exact scan's own distances likewise
    k1f = ix.distance_batch(queries[:64], f_ids[:64].cpu().numpy().view(np.uint64))
    assert np.array_equal(bits(k1f.cpu().numpy()), bits(f_d[:64].cpu().numpy()))
    ix.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_headline_protocol_on_the_reference_schedule_graph():
    """tools/reference_schedule_graph.py at a size the suite can afford: the sequential device build (the reference's
    insertSinglePoint loop, insert.go:16-68) equals the oracle's on a prefix of the bench's own rows edge for edge, and
    the batched rounds' graph answers the same batches with the same recall and walk length as the sequential graph
    (profiles/r06_refsched_1m.json is the same comparison at 1M x 384)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "reference_schedule_graph.py"), "--rows", "30000",
                          "--prefix", "6000", "--chunk", "12000", "--timed-batches", "4"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["prefix"] == dict(j["prefix"], rows=6000, equal_to_oracle_edge_for_edge=True)
    assert j["rows_reached"] == 30000
    seq, bat = j["reference_schedule_graph"], j["batched_graph"]
    assert seq["recall_at_10"] >= 0.95 and bat["recall_at_10"] >= 0.95
    assert abs(seq["recall_at_10"] - bat["recall_at_10"]) <= 0.01
    assert 0.95 <= bat["mean_n_dist"] / seq["mean_n_dist"] <= 1.05
    assert 0.95 <= bat["mean_degree"] / seq["mean_degree"] <= 1.05
