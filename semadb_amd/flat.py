"""Mirror of the reference's flat package (shard/index/flat/flat.go) over the C ABI: exact scan."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _buf
from ._lib import IndexParams, METRICS, SemaDBError, check, lib
from .vamana import IndexVectorChange, SearchResult


@dataclass
class IndexVectorFlatParameters:
    """models.IndexVectorFlatParameters (models/index.go:238-246)"""
    VectorSize: int
    DistanceMetric: str


@dataclass
class SearchVectorFlatOptions:
    """models.SearchVectorFlatOptions (models/search.go:308-314)"""
    Vector: object
    Limit: int = 10
    Weight: Optional[float] = None


def flat_search_batch(handle, dim, queries, limit, filters=None, device=0):
    """exact kNN over the rows of any device index (also the recall ground truth of a graph index)"""
    k, qp, mem, shape = _buf.as_f32(queries)
    if len(shape) != 2 or shape[1] != dim:
        raise SemaDBError(1, "query vector length must be %d" % dim)
    nq = shape[0]
    f_off = f_ids = None
    if filters is not None:
        flat, off = [], [0]
        for f in filters:
            flat.extend(sorted(int(v) for v in f))
            off.append(len(flat))
        f_off = np.array(off, dtype=np.uint64)
        f_ids = np.array(flat if flat else [0], dtype=np.uint64)
    ids, idp = _buf.empty_like_mem(mem, (nq, limit), "uint64", device)
    dists, dp = _buf.empty_like_mem(mem, (nq, limit), "float32", device)
    counts, cp = _buf.empty_like_mem(mem, (nq,), "uint32", device)
    check(lib().sdb_index_flat_search(handle, nq, qp, limit, _buf.np_ptr(f_off), _buf.np_ptr(f_ids), idp, dp, cp, mem,
                                      _buf.current_stream(mem)))
    return ids, dists, counts


class IndexFlat:
    """flat.IndexFlat (flat.go:17-19): a vector store and nothing else."""

    def __init__(self, params: IndexVectorFlatParameters, bucket=None, device=0, capacity=0):
        if params.DistanceMetric not in METRICS:
            raise SemaDBError(1, "failed to create vector store: unknown float32 distance function: %s" %
                              params.DistanceMetric)
        self.parameters, self.device = params, device
        p = IndexParams(params.VectorSize, METRICS[params.DistanceMetric], 75, 64, 1.2, device, capacity, 0)
        h = C.c_void_p()
        check(lib().sdb_index_create(C.byref(p), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().sdb_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def InsertUpdateDelete(self, points):
        """flat.go:41-74: vecStore.Set for a point with a vector (insert or replace), vecStore.Delete for one
        without, in the order given; one write transaction."""
        ops = []  # runs of consecutive sets / deletes keep the reference's order between the two kinds
        for ch in points:
            kind = "del" if ch.Vector is None else "set"
            if ops and ops[-1][0] == kind and (kind == "del" or ch.Id not in ops[-1][3]):
                ops[-1][1].append(ch.Id)
                ops[-1][3].add(ch.Id)
            else:
                ops.append([kind, [ch.Id], [], {ch.Id}])
            if kind == "set":
                ops[-1][2].append(np.asarray(ch.Vector, dtype=np.float32))
        if not ops:
            return
        # everything the caller can get wrong is checked before the transaction opens (ragged vectors raise here)
        batches = []
        for kind, ids, vecs, _ in ops:
            if kind == "set":
                m = np.stack(vecs)
                if m.ndim != 2 or m.shape[1] != self.parameters.VectorSize:
                    raise SemaDBError(1, "failed to insert/update/delete: vector length mismatch")
                batches.append((kind, np.array(ids, dtype=np.uint64), m))
            else:
                batches.append((kind, ids, None))
        check(lib().sdb_index_begin_write(self._h))
        try:
            for kind, ids, m in batches:
                if kind == "set":
                    self.set_vectors(ids, m)
                else:
                    self.remove_vectors(ids)
            check(lib().sdb_index_commit(self._h, None))
        except Exception:
            self.abort_write()  # never leave the transaction open: the next write would be refused
            raise

    def set_vectors(self, ids, vectors):
        k, vp, mem, shape = _buf.as_f32(vectors)
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
        check(lib().sdb_index_set_vectors(self._h, shape[0], _buf.np_ptr(ids_a), vp, mem))

    def remove_vectors(self, ids):
        ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
        if ids_a.size:
            check(lib().sdb_index_remove_vectors(self._h, ids_a.size, _buf.np_ptr(ids_a)))

    def begin_write(self):
        check(lib().sdb_index_begin_write(self._h))

    def commit(self):
        check(lib().sdb_index_commit(self._h, None))

    def abort_write(self):
        """leave an open transaction without committing: rolled back, the store is what it was at begin_write"""
        return lib().sdb_index_abort_write(self._h) == 0

    def row_usage(self):
        """(storage rows in use, of which tombstones)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().sdb_index_row_usage(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def compact(self):
        check(lib().sdb_index_compact(self._h))

    def version_diff(self):
        v = C.c_uint64(0)
        check(lib().sdb_index_version_diff(self._h, C.byref(v)))
        return v.value

    def Search(self, options: SearchVectorFlatOptions, filter=None):
        vec = np.ascontiguousarray(options.Vector, dtype=np.float32).reshape(1, -1)
        ids, dists, counts = flat_search_batch(self._h, self.parameters.VectorSize, vec, options.Limit,
                                               None if filter is None else [filter], self.device)
        weight = np.float32(1) if options.Weight is None else np.float32(options.Weight)
        res = [SearchResult(int(ids[0, i]), np.float32(dists[0, i]), np.float32(-1) * weight * np.float32(dists[0, i]))
               for i in range(int(counts[0]))]  # flat.go:114 (-1 * weight * dist)
        return set(r.NodeId for r in res), res

    def search_batch(self, queries, limit, filters=None):
        return flat_search_batch(self._h, self.parameters.VectorSize, queries, limit, filters, self.device)


def NewIndexFlat(params, bucket=None, **kw):
    """flat.NewIndexFlat (flat.go:21-32)"""
    return IndexFlat(params, bucket, **kw)
