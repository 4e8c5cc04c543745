"""Mirror of the shard fan-out in ClusterNode.SearchPoints (cluster/actions.go:275-379).

The reference scatters a query to every shard over msgpack net/rpc, gathers the per-shard results
and sorts/truncates them (actions.go:316-376).  On one 8 x MI355X node the shards live one per GPU;
the gather step is a single RCCL all-gather over xGMI of the fixed-size per-shard top-k blocks and the
merge runs on device (merge.hip).

Transports for the gather step:
  Cluster        the product path: a binding of sdb_cluster_* (csrc/cluster.hip), where the library itself
                 owns the exchange stream and the transport -- RCCL between GPUs (create / create_local with
                 distinct devices), device copies behind a host rendezvous for shards that share one GPU
                 (create_local with the same device repeated) -- what a Go host calls.
  allgather_topk / PackedTopK.allgather / PackedTopK.exchange
                 the same exchange over any torch.distributed backend; exists so that the N > 1 logic
                 (block layout, tags, shard-major gather, merge rule) runs under gloo on machines without
                 N GPUs (tests/test_cluster.py, BENCH_BACKEND=gloo).  PackedTopK.exchange stamps the block's tag
                 and runs the library's tag check + merge on the gathered buffer (sdb_cluster_stamp_block /
                 sdb_cluster_merge_gathered), so a mis-ordered collective is refused on this path too.

Order: collective calls take a `ticket` (1, 2, 3, ... drawn once per request by whoever fans it out, see Fanout):
a rank's calls enter the exchange in ticket order whichever thread arrives first, and the tags every block carries
are compared after the gather (include/semadb_amd.h "Collective calls, order and failure").
"""
import threading

import ctypes as C

import numpy as np

from . import _buf
from ._lib import MEM_DEVICE, SemaDBError, check, lib


def shard_limit(limit, n_shards, max_search_limit=75):
    """per-shard target limit, cluster/actions.go:291-299"""
    out = C.c_uint32(0)
    check(lib().sdb_shard_limit(limit, n_shards, max_search_limit, C.byref(out)))
    return out.value


def topk_merge(ids, dists, counts, limit, device=0):
    """ids/dists [n_shards, nq, per_shard], counts [n_shards, nq] -> merged (ids, dists, shards, counts).

    numpy in -> numpy out; torch CUDA in -> torch out on the current stream.  Ties break by
    (shard, id) ascending (the reference's sort is unstable there, actions.go:357-364)."""
    if _buf.is_torch_cuda(ids):
        import torch
        n_shards, nq, per = ids.shape
        ids, dists, counts = ids.contiguous(), dists.contiguous(), counts.contiguous()
        o_ids = torch.zeros((nq, limit), dtype=torch.int64, device=ids.device)
        o_d = torch.zeros((nq, limit), dtype=torch.float32, device=ids.device)
        o_s = torch.zeros((nq, limit), dtype=torch.int32, device=ids.device)
        o_c = torch.zeros((nq,), dtype=torch.int32, device=ids.device)
        check(lib().sdb_topk_merge(n_shards, nq, per, C.c_void_p(ids.data_ptr()), C.c_void_p(dists.data_ptr()),
                                   C.c_void_p(counts.data_ptr()), limit, C.c_void_p(o_ids.data_ptr()),
                                   C.c_void_p(o_d.data_ptr()), C.c_void_p(o_s.data_ptr()),
                                   C.c_void_p(o_c.data_ptr()), MEM_DEVICE, device, _buf.current_stream(MEM_DEVICE)))
        return o_ids, o_d, o_s, o_c
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_shards, nq, per = ids.shape
    o_ids = np.zeros((nq, limit), dtype=np.uint64)
    o_d = np.zeros((nq, limit), dtype=np.float32)
    o_s = np.zeros((nq, limit), dtype=np.uint32)
    o_c = np.zeros((nq,), dtype=np.uint32)
    check(lib().sdb_topk_merge(n_shards, nq, per, _buf.np_ptr(ids), _buf.np_ptr(dists), _buf.np_ptr(counts), limit,
                               _buf.np_ptr(o_ids), _buf.np_ptr(o_d), _buf.np_ptr(o_s), _buf.np_ptr(o_c), 0, device,
                               None))
    return o_ids, o_d, o_s, o_c


def allgather_topk(ids, dists, counts):
    """The exchange step: every rank contributes its shard's [nq, per_shard] block and receives the
    shard-major [world, nq, per_shard] buffers the merge consumes.  Works on any torch.distributed
    backend (nccl = RCCL over xGMI on the GPU box, gloo for CPU tests).  Message size per rank:
    nq * per_shard * 12 B + nq * 4 B (120 KB at 1024 x 10): latency-bound, one step."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    ids = ids.contiguous()
    dists = dists.contiguous()
    counts = counts.contiguous()
    g_ids = torch.empty((world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
    g_d = torch.empty((world,) + tuple(dists.shape), dtype=dists.dtype, device=dists.device)
    g_c = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
    # concatenated-along-dim-0 views of the shard-major buffers (the form every backend accepts)
    dist.all_gather_into_tensor(g_ids.view((-1,) + tuple(ids.shape[1:])), ids)
    dist.all_gather_into_tensor(g_d.view((-1,) + tuple(dists.shape[1:])), dists)
    dist.all_gather_into_tensor(g_c.view((-1,) + tuple(counts.shape[1:])), counts)
    return g_ids, g_d, g_c


def allgather_merge(ids, dists, counts, limit, device=0):
    """all-gather the per-shard results and merge them on this rank's GPU (every rank ends up with the
    merged answer, like every SemaDB server can answer the REST call)."""
    g_ids, g_d, g_c = allgather_topk(ids, dists, counts)
    return topk_merge(g_ids, g_d, g_c, limit, device=device)


class PackedTopK:
    """One shard's [nq, per_shard] result block laid out as a single byte buffer (ids | dists | counts), so the
    search kernel writes straight into the message of ONE all-gather per batch instead of three."""

    def __init__(self, nq, per_shard, device):
        import torch
        self.nq, self.per = nq, per_shard
        self.b_ids, self.b_d, self.b_c = nq * per_shard * 8, nq * per_shard * 4, nq * 4
        off_d, off_c, off_t, total = block_layout(nq, per_shard)  # the layout the C ABI defines (semadb_amd.h)
        assert off_d == self.b_ids and off_c == self.b_ids + self.b_d and total == off_t + 64
        self.off_tag = off_t
        # the search kernel writes every element (short rows are zero-padded by the kernel); the padding
        # behind the counts and the tag travel too, so the buffer starts out cleared
        self.buf = torch.zeros(total, dtype=torch.uint8, device=device)
        self.ids = self.buf[:self.b_ids].view(torch.int64).view(nq, per_shard)
        self.dists = self.buf[self.b_ids:self.b_ids + self.b_d].view(torch.float32).view(nq, per_shard)
        self.counts = self.buf[self.b_ids + self.b_d:self.b_ids + self.b_d + self.b_c].view(torch.int32)

    def out(self):
        return self.ids, self.dists, self.counts

    def allgather(self):
        """-> shard-major (ids [W,nq,per], dists, counts [W,nq]) on every rank"""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size()
        g = torch.empty(world * self.buf.numel(), dtype=torch.uint8, device=self.buf.device)
        dist.all_gather_into_tensor(g, self.buf)
        g = g.view(world, -1)
        ids = g[:, :self.b_ids].contiguous().view(torch.int64).view(world, self.nq, self.per)
        d = g[:, self.b_ids:self.b_ids + self.b_d].contiguous().view(torch.float32).view(world, self.nq, self.per)
        c = g[:, self.b_ids + self.b_d:self.b_ids + self.b_d + self.b_c].contiguous().view(torch.int32).view(world, self.nq)
        return ids, d, c


    def exchange(self, limit, seq, ticket=0, queries=None, status=0):
        """The torch.distributed form of sdb_cluster_search_batch's exchange step with the library's guards: stamp
        this block's tag (sequence number, ticket, shape, hash of the queries this rank searched), all-gather the
        blocks, then tag check + merge on this rank's GPU.  Raises SemaDBError(SDB_ERR_STATE) when the gathered blocks
        belong to different requests or a shard reported a failure.  Blocking.  CUDA buffers only."""
        import torch
        import torch.distributed as dist
        dev = self.buf.device
        stream = _buf.current_stream(MEM_DEVICE)
        qp, dim = None, 0
        if queries is not None:
            queries = queries.contiguous()
            qp, dim = C.c_void_p(queries.data_ptr()), queries.shape[1]
        check(lib().sdb_cluster_stamp_block(C.c_void_p(self.buf.data_ptr()), self.nq, self.per, limit, dist.get_rank(),
                                            seq, ticket, status, qp, dim, dev.index or 0, stream))
        world = dist.get_world_size()
        if dist.get_backend() == "gloo":  # gloo moves host tensors
            mine = self.buf.cpu()
            g = torch.empty(world * mine.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(g, mine)
            g = g.to(dev)
        else:
            g = torch.empty(world * self.buf.numel(), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(g, self.buf)
        o_ids = torch.zeros((self.nq, limit), dtype=torch.int64, device=dev)
        o_d = torch.zeros((self.nq, limit), dtype=torch.float32, device=dev)
        o_s = torch.zeros((self.nq, limit), dtype=torch.int32, device=dev)
        o_c = torch.zeros((self.nq,), dtype=torch.int32, device=dev)
        check(lib().sdb_cluster_merge_gathered(world, self.nq, self.per, C.c_void_p(g.data_ptr()), limit,
                                               C.c_void_p(o_ids.data_ptr()), C.c_void_p(o_d.data_ptr()),
                                               C.c_void_p(o_s.data_ptr()), C.c_void_p(o_c.data_ptr()), dev.index or 0,
                                               stream))
        return o_ids, o_d, o_s, o_c


def block_layout(nq, per_shard):
    """(off_dists, off_counts, off_tag, bytes) of one shard's result block, sdb_cluster_block_layout"""
    a, b, t, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    check(lib().sdb_cluster_block_layout(nq, per_shard, C.byref(a), C.byref(b), C.byref(t), C.byref(c)))
    return a.value, b.value, t.value, c.value


class Cluster:
    """One rank (one shard, one GPU) of the RCCL exchange, sdb_cluster_* of the C ABI.  The library owns
    the communicator and the exchange stream; torch only supplies device buffers here."""

    def __init__(self, handle, rank, world, device):
        self._h, self.rank, self.world, self.device = handle, rank, world, device

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().sdb_cluster_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, rank, world, uid, device):
        h = C.c_void_p()
        raw = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(lib().sdb_cluster_create(rank, world, raw, device, C.byref(h)))
        return cls(h, rank, world, device)

    @classmethod
    def create_local(cls, devices):
        """all ranks of a single-process node (the shape of a Go server owning every GPU)"""
        n = len(devices)
        hs = (C.c_void_p * n)()
        devs = (C.c_int * n)(*devices)
        check(lib().sdb_cluster_create_local(n, devs, hs))
        return [cls(C.c_void_p(hs[i]), i, n, devices[i]) for i in range(n)]

    @classmethod
    def from_torch_distributed(cls, device):
        """bootstrap over an existing torch.distributed group: rank 0 draws the unique id, everybody gets it"""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.create(rank, world, box[0], device)

    def close(self):
        if getattr(self, "_h", None):
            lib().sdb_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _outs(self, nq, limit, mem):
        o_ids, p_ids = _buf.empty_like_mem(mem, (nq, limit), "uint64", self.device)
        o_d, p_d = _buf.empty_like_mem(mem, (nq, limit), "float32", self.device)
        o_s, p_s = _buf.empty_like_mem(mem, (nq, limit), "uint32", self.device)
        o_c, p_c = _buf.empty_like_mem(mem, (nq,), "uint32", self.device)
        return (o_ids, o_d, o_s, o_c), (p_ids, p_d, p_s, p_c)

    def allgather_merge(self, block, limit, host_out=False, ticket=0):
        """block: PackedTopK on this rank's GPU, written by work on the current stream.  Returns merged
        (ids, dists, shards, counts): device tensors valid after wait()/synchronize(), or numpy if host_out."""
        mem = 0 if host_out else MEM_DEVICE
        outs, ptrs = self._outs(block.nq, limit, mem)
        check(lib().sdb_cluster_allgather_merge(self._h, ticket, block.nq, block.per, C.c_void_p(block.buf.data_ptr()),
                                                limit, ptrs[0], ptrs[1], ptrs[2], ptrs[3], mem,
                                                _buf.current_stream(MEM_DEVICE)))
        return outs

    def search_batch(self, ix, queries, limit, search_size, ticket=0, want=True):
        """ClusterNode.SearchPoints for this rank's shard: search -> all-gather -> merge.  numpy queries ->
        numpy results (synchronous); torch CUDA queries -> device results, valid after wait()/synchronize()
        (synchronize raises if the exchange failed its tag check).  want=False (host memory only): take part in the
        exchange and return its verdict, but do not copy this rank's copy of the merged answer back (-> None)."""
        k, qp, mem, shape = _buf.as_f32(queries)
        if not want and mem != MEM_DEVICE:
            check(lib().sdb_cluster_search_batch(self._h, ix._h, ticket, shape[0], qp, limit, search_size, None, None,
                                                 None, None, mem, _buf.current_stream(mem)))
            return None
        outs, ptrs = self._outs(shape[0], limit, mem)
        check(lib().sdb_cluster_search_batch(self._h, ix._h, ticket, shape[0], qp, limit, search_size, ptrs[0], ptrs[1],
                                             ptrs[2], ptrs[3], mem, _buf.current_stream(mem)))
        return outs

    def next_ticket(self):
        t = C.c_uint64(0)
        check(lib().sdb_cluster_next_ticket(self._h, C.byref(t)))
        return t.value

    def set_deadline(self, milliseconds):
        """longest wait at the turnstile / for the peers (0: for ever), sdb_cluster_set_deadline"""
        check(lib().sdb_cluster_set_deadline(self._h, int(milliseconds)))

    def skip_ticket(self, ticket, nq=0, limit=0, per_shard=0):
        """the fan-out gives `ticket` up for this rank: nq == 0 -> the turnstile passes over it; nq > 0 -> this rank
        enters the request's exchange with an empty answer under an error flag so that its peers fail that request and
        serve the next (sdb_cluster_skip_ticket)"""
        check(lib().sdb_cluster_skip_ticket(self._h, ticket, nq, per_shard, limit))

    def transport(self):
        buf = C.create_string_buffer(512)
        check(lib().sdb_cluster_transport(self._h, buf, 512))
        return buf.value.decode()

    def wait(self):
        """the current torch stream waits (on the device) for every exchange enqueued so far"""
        check(lib().sdb_cluster_wait(self._h, _buf.current_stream(MEM_DEVICE)))

    def synchronize(self):
        check(lib().sdb_cluster_synchronize(self._h))



class Fanout:
    """ClusterNode.SearchPoints for the shards of one process (cluster/actions.go:316-376): one request = one ticket,
    handed to every rank's sdb_cluster_search_batch from a thread of its own (the calls are collective and, with
    host memory, blocking).  The Python twin of integration/go/cluster/fanout_mi355x.go and of
    semadb::cluster::GpuFanout (semadb_host.hpp); requests may be issued from many threads at once."""

    def __init__(self, ranks, indexes):
        assert len(ranks) == len(indexes)
        self.ranks, self.indexes = ranks, indexes
        self._mu = threading.Lock()
        self._next = ranks[0].next_ticket()

    def _ticket(self):
        with self._mu:
            t = self._next
            self._next += 1
            return t

    def search_points(self, queries, limit, search_size):
        """numpy queries [nq, dim] -> merged (ids, dists, shards, counts) as numpy; raises the first rank's error.
        Every rank's GPU holds the same merged answer after the exchange; only rank 0 copies it to the host (the other
        ranks' calls run with device outputs that are dropped).  A rank whose call could not even be made has the
        ticket skipped for it, so that neither its peers (inside the exchange) nor its later requests (at the
        turnstile) wait for it: the reference fails one request and serves the next (actions.go:339-353)."""
        ticket = self._ticket()
        n = len(self.ranks)
        outs, errs, called = [None] * n, [None] * n, [False] * n
        nq = int(np.asarray(queries).shape[0]) if hasattr(queries, "shape") else len(queries)

        def run(r):
            try:
                called[r] = True
                outs[r] = self.ranks[r].search_batch(self.indexes[r], queries, limit, search_size, ticket=ticket,
                                                     want=(r == 0))
            except SemaDBError as e:  # every rank sees the failure of any rank
                errs[r] = e
            except BaseException as e:  # the binding itself failed before the library saw the ticket
                called[r] = False
                errs[r] = e

        ts = [threading.Thread(target=run, args=(r,)) for r in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for r in range(n):
            if not called[r]:
                try:
                    self.ranks[r].skip_ticket(ticket, nq, limit)
                except SemaDBError:
                    pass
        for e in errs:
            if e is not None:
                raise e
        return outs[0]  # every rank holds the same merged answer
