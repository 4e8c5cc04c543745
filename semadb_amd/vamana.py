"""Mirror of the reference's vamana package (shard/index/vamana/) over the C ABI.

Same exported surface as the Go package a maintainer would swap out:
STARTID (vamana.go:28), IndexVectorChange (vamana.go:122-125), NewIndexVamana (vamana.go:54),
IndexVamana.Search (vamana.go:278), InsertUpdateDelete (vamana.go:127), SizeInMemory (vamana.go:83).
The index state lives in HBM; this module only marshals.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _buf
from ._lib import (IndexParams, MEM_DEVICE, MEM_HOST, METRICS, SearchTrace, SemaDBError, check, lib)

STARTID = 1  # vamana.go:28


@dataclass
class IndexVectorVamanaParameters:
    """models.IndexVectorVamanaParameters (models/index.go:275-282)."""
    VectorSize: int
    DistanceMetric: str
    SearchSize: int = 75
    DegreeBound: int = 64
    Alpha: float = 1.2
    Quantizer: object = None  # vectorstore.Quantizer (models/index.go:281)


@dataclass
class SearchVectorVamanaOptions:
    """models.SearchVectorVamanaOptions (models/search.go:268-275)."""
    Vector: object
    SearchSize: int = 75
    Limit: int = 10
    Weight: Optional[float] = None


@dataclass
class SearchResult:
    """models.SearchResult fields the index fills (vamana.go:300-304)."""
    NodeId: int
    Distance: np.float32
    HybridScore: np.float32


@dataclass
class IndexVectorChange:
    """vamana.IndexVectorChange (vamana.go:122-125); Vector None means delete."""
    Id: int
    Vector: object = None


@dataclass
class BatchTrace:
    n_dist: np.ndarray
    n_hop: np.ndarray
    n_edges: np.ndarray
    visit_ids: Optional[np.ndarray] = None


class FilterBitmaps:
    """The filters of a batch as bitmaps (sdb_index_search_batch_bitmap): query q's filter is {first_id[q] + i : bit i of
    its words}; words[word_offsets[q]:word_offsets[q + 1]] are its 64-bit words.  from_sets() builds the form a roaring
    bitmap's dense containers already have."""

    def __init__(self, first_id, word_offsets, words):
        self.first_id = np.ascontiguousarray(first_id, dtype=np.uint64)
        self.word_offsets = np.ascontiguousarray(word_offsets, dtype=np.uint64)
        self.words = np.ascontiguousarray(words if len(words) else [0], dtype=np.uint64)

    @classmethod
    def from_sets(cls, filters, align=64):
        first, off, chunks = [], [0], []
        for f in filters:
            ids = np.unique(np.asarray(sorted(int(v) for v in f), dtype=np.uint64)) if not isinstance(f, np.ndarray) \
                else np.unique(f.astype(np.uint64))
            if ids.size == 0:
                first.append(0)
                off.append(off[-1])
                continue
            f0 = int(ids[0]) // align * align
            rel = (ids - np.uint64(f0)).astype(np.int64)
            nwords = int(rel[-1]) // 64 + 1
            w = np.zeros(nwords, dtype=np.uint64)
            np.bitwise_or.at(w, rel // 64, np.uint64(1) << (rel % 64).astype(np.uint64))
            first.append(f0)
            chunks.append(w)
            off.append(off[-1] + nwords)
        words = np.concatenate(chunks) if chunks else np.zeros(1, dtype=np.uint64)
        return cls(np.array(first, dtype=np.uint64), np.array(off, dtype=np.uint64), words)


class IndexVamana:
    """vamana.IndexVamana (vamana.go:36-52) with its state pinned in one MI355X's HBM."""

    def __init__(self, name, params: IndexVectorVamanaParameters, bucket=None, device=0, capacity=0, strict=True,
                 fit_seed=None):
        if params.DistanceMetric not in METRICS:
            raise SemaDBError(1, "unknown distance metric %s" % params.DistanceMetric)
        self.name, self.parameters, self.device = name, params, device
        # vectorstore.New (vamana.go:64): None = plain store (the slab); a ProductQuantizer waits for Fit
        from . import vectorstore
        self._store = vectorstore.New(getattr(params, "Quantizer", None), params.DistanceMetric, params.VectorSize, device)
        self._pq = None
        self._fit_rng = np.random.default_rng(fit_seed)
        self.last_fit_first_idx = None
        p = IndexParams(params.VectorSize, METRICS[params.DistanceMetric], params.SearchSize, params.DegreeBound,
                        params.Alpha, device, capacity, 1 if strict else 0)
        h = C.c_void_p()
        check(lib().sdb_index_create(C.byref(p), C.byref(h)))
        self._h = h
        # test harness only (the library itself never reads the environment): run a whole suite with the two-precision
        # hop switched on -- 2 = with its audit -- on every index this mirror creates, and have close() check the audit
        self._forced_sketch = int(os.environ.get("SEMADB_AMD_TEST_SKETCH", "0") or 0)
        if self._forced_sketch:
            self.set_tuning("sketch", self._forced_sketch)
            if os.environ.get("SEMADB_AMD_TEST_WIDE_WALK"):  # ... and the batch walk (which has the stage) for small calls too
                self.set_tuning("wide_walk", int(os.environ["SEMADB_AMD_TEST_WIDE_WALK"]))

    def close(self):
        if getattr(self, "_h", None):
            bad = self.sketch_stats()[1] if getattr(self, "_forced_sketch", 0) else 0
            lib().sdb_index_destroy(self._h)
            self._h = None
            if bad:
                raise AssertionError("two-precision hop: %d discarded neighbours had an exact distance that would have been kept" % bad)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setupStartNode vamana.go:93-120 ------------------------------------------------------
    def set_start(self, vec):
        k, p, mem, shape = _buf.as_f32(vec)
        if int(np.prod(shape)) != self.parameters.VectorSize:
            raise SemaDBError(1, "start vector has wrong length")
        check(lib().sdb_index_set_start(self._h, p, mem))

    # ---- bucket -> HBM -------------------------------------------------------------------------
    def load(self, ids, vectors, offsets, edges):
        k, vp, mem, shape = _buf.as_f32(vectors)
        n = shape[0]
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ed = np.ascontiguousarray(edges if len(edges) else [0], dtype=np.uint64)
        check(lib().sdb_index_load(self._h, n, _buf.np_ptr(ids_a), vp, _buf.np_ptr(off), _buf.np_ptr(ed), mem))

    def export(self, with_vectors=True):
        n, ne, _ = self.stats()
        ids = np.zeros(n, dtype=np.uint64)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        edges = np.zeros(max(ne, 1), dtype=np.uint64)
        vecs = np.zeros((n, self.parameters.VectorSize), dtype=np.float32) if with_vectors else None
        check(lib().sdb_index_export(self._h, _buf.np_ptr(ids), _buf.np_ptr(vecs), _buf.np_ptr(offsets),
                                     _buf.np_ptr(edges)))
        return ids, vecs, offsets, edges[:ne]

    def stats(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().sdb_index_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    TUNING = {"hub_min": 1, "hash_limit": 2, "no_hash": 3, "no_tile": 4, "no_mfma": 5, "wide_hash": 6, "hash16_probes": 7,
              "pq_narrow": 8, "wide_walk": 9, "host_filters": 10, "no_defer": 11, "no_zero_copy": 12,
              "sketch": 13}  # SDB_TUNE_* (semadb_amd.h)

    def sketch_stats(self):
        """(neighbours discarded on their float16 distance, contradicted by the exact distance [audit], copy in use)"""
        out = (C.c_uint64 * 3)()
        check(lib().sdb_index_sketch_stats(self._h, out))
        return int(out[0]), int(out[1]), bool(out[2])

    def set_tuning(self, key, value):
        """test / measurement knobs of this index (sdb_index_set_tuning); none changes a result"""
        check(lib().sdb_index_set_tuning(self._h, self.TUNING[key], int(value)))

    BUILD_STATS = ("search_n_dist", "search_n_edges", "prune_pairs", "backedge_pairs", "backedge_cached",
                   "requests", "reprunes", "appends", "staged_rows", "rounds", "hubs")

    def build_stats(self):
        """counters of the most recent insert_batch (sdb_index_build_stats)"""
        out = np.zeros(len(self.BUILD_STATS), dtype=np.uint64)
        check(lib().sdb_index_build_stats(self._h, _buf.np_ptr(out), out.size))
        return dict(zip(self.BUILD_STATS, (int(v) for v in out)))

    def GetMany(self, ids):
        """vecStore.GetMany (plain.go:26-45): (vectors of the ids that are stored, in request order; found mask)"""
        ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.zeros((ids_a.size, self.parameters.VectorSize), dtype=np.float32)
        found = np.zeros(ids_a.size, dtype=np.uint8)
        if ids_a.size:
            check(lib().sdb_index_get_vectors(self._h, ids_a.size, _buf.np_ptr(ids_a), _buf.np_ptr(out), _buf.np_ptr(found)))
        return out[found.astype(bool)], found.astype(bool)

    def exists_batch(self, ids):
        """vecStore.Exists (plain.go:21-24) for many ids: host-side table lookup, no device work"""
        ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.zeros(ids_a.size, dtype=np.uint8)
        if ids_a.size:
            check(lib().sdb_index_exists_batch(self._h, ids_a.size, _buf.np_ptr(ids_a), _buf.np_ptr(out)))
        return out.astype(bool)

    def row_usage(self):
        """(storage rows in use, of which tombstones) -- sdb_index_row_usage"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().sdb_index_row_usage(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def compact(self):
        """drop the tombstones of deleted / updated points (sdb_index_compact); ids, graph and answers are unchanged"""
        check(lib().sdb_index_compact(self._h))

    def set_profiling(self, enabled=True):
        check(lib().sdb_index_set_profiling(self._h, 1 if enabled else 0))

    def last_search_ms(self):
        v = C.c_float(0)
        check(lib().sdb_index_last_search_ms(self._h, C.byref(v)))
        return v.value

    def profile_read(self):
        """kernel durations (ms) of the profiled K2 launches since the last read, oldest first"""
        buf = np.zeros(256, dtype=np.float32)
        n = C.c_uint32(0)
        check(lib().sdb_index_profile_read(self._h, _buf.np_ptr(buf), 256, C.byref(n)))
        return buf[:n.value].copy()

    def SizeInMemory(self):
        v = C.c_int64(0)
        check(lib().sdb_index_size_in_memory(self._h, C.byref(v)))
        return v.value

    # ---- InsertUpdateDelete vamana.go:127-263 (insert branch on device) --------------------------
    def InsertUpdateDelete(self, points, round_size=0, _between=None):
        """points: iterable of IndexVectorChange.  Same classification and order as vamana.go:149-251: new
        ids are inserted first; then the inbound edges of deleted AND updated ids are removed in one scan
        and the deleted nodes dropped; then the updated points are re-inserted one by one.  `_between` (tests): called
        after every step inside the open transaction."""
        ins_ids, ins_vecs, upd_ids, upd_vecs, del_ids = [], [], [], [], []
        known = set()
        points = list(points)
        stored = self.exists_batch([ch.Id for ch in points])  # one table lookup for the whole change list
        for ch, in_store in zip(points, stored):
            if ch.Id == STARTID:
                raise SemaDBError(1, "cannot modify point with start id: %d" % STARTID)  # vamana.go:150-153
            if ch.Id == 0:
                raise SemaDBError(1, "invalid point id: %d" % ch.Id)  # vamana.go:154-157
            exists = bool(in_store) or ch.Id in known
            if ch.Vector is None:
                if exists:
                    del_ids.append(ch.Id)  # :175-179
                continue  # !exists && nil: nothing to do (:161-163)
            v = np.asarray(ch.Vector, dtype=np.float32)
            if exists:
                upd_ids.append(ch.Id)  # :170-174
                upd_vecs.append(v)
            else:
                ins_ids.append(ch.Id)
                ins_vecs.append(v)
                known.add(ch.Id)
        if not (ins_ids or del_ids or upd_ids):
            return
        # what the caller can get wrong is checked before the transaction opens (models/index.go:182-184)
        for v in ins_vecs + upd_vecs:
            if v.ndim != 1 or v.shape[0] != self.parameters.VectorSize:
                raise SemaDBError(1, "vector length mismatch: expected %d" % self.parameters.VectorSize)
        ins_mat = np.stack(ins_vecs) if ins_ids else None
        self.begin_write()  # one transaction, like the shard's (searches see it whole or not at all)
        try:
            if ins_ids:
                self.insert_batch(np.array(ins_ids, dtype=np.uint64), ins_mat, round_size)
                if _between:
                    _between("inserts")
            if del_ids or upd_ids:
                self.delete_batch(np.array(del_ids + upd_ids, dtype=np.uint64))  # removeInboundEdges :223-233
                if _between:
                    _between("deletes")
            for i, v in zip(upd_ids, upd_vecs):  # :247-251 re-inserted sequentially
                self.insert_batch(np.array([i], dtype=np.uint64), v.reshape(1, -1), 1)
            if _between and upd_ids:
                _between("updates")
            self.commit()
        except Exception:
            self.abort_write()  # never leave the transaction open: the next write would be refused
            raise
        self.Fit()  # vamana.go:257-260

    def Fit(self):
        """vecStore.Fit (vamana.go:258) = productQuantizer.Fit (product.go:175-236): once, when the store holds
        TriggerThreshold points (the start node is one of them).  k-means per sub-quantizer over all stored
        vectors in storage order (the reference walks a Go map), first centroid drawn at random
        (kmeans.go:61-63); the labels become the points' centroid ids (:216-218)."""
        pq = self._store
        if pq is None or self._pq is not None:
            return False
        n = self.stats()[0]
        if n < pq.params.TriggerThreshold:
            return False
        from . import vectorstore
        ids, vecs, _, _ = self.export()
        first = self._fit_rng.integers(0, n, pq.M)
        self.last_fit_first_idx = first
        codes = pq.Fit(vecs, first, alias=True)  # `vecs` is a private copy: the aliasing write-through stays in it
        vectorstore.attach(self, pq, ids, codes)
        return True

    def union_prune(self, node_id, extra_ids, chip_wide=False):
        """insert.go:47-58 over the node's neighbours + several candidates at once (Add, Sort, robustPrune)"""
        ex = np.ascontiguousarray(extra_ids, dtype=np.uint64)
        check(lib().sdb_index_union_prune(self._h, int(node_id), ex.size, _buf.np_ptr(ex), 1 if chip_wide else 0, None))

    def exists(self, node_id):
        """vecStore.Exists (plain.go:21-24)"""
        return bool(self.exists_batch([node_id])[0])

    def begin_write(self):
        """open a write transaction: until commit() searches keep walking the graph as it is now"""
        check(lib().sdb_index_begin_write(self._h))

    def commit(self):
        check(lib().sdb_index_commit(self._h, None))

    def abort_write(self):
        """leave an open transaction without committing (the error path of InsertUpdateDelete): the transaction is
        rolled back and the index is what it was at begin_write (True); False only for a handle a device failure had
        already left unusable"""
        return lib().sdb_index_abort_write(self._h) == 0

    def version_diff(self):
        """test support: rows on which the committed and the writer's copy differ"""
        v = C.c_uint64(0)
        check(lib().sdb_index_version_diff(self._h, C.byref(v)))
        return v.value

    def EdgeScan(self, deleteSet):
        """IndexVamana.EdgeScan (node.go:142-199): (toPrune, toSave) for a set of ids about to be deleted"""
        ids_a = np.ascontiguousarray(sorted(int(v) for v in deleteSet), dtype=np.uint64)
        n = self.stats()[0] + 1
        tp, ts = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().sdb_index_edge_scan(self._h, ids_a.size, _buf.np_ptr(ids_a) if ids_a.size else None, _buf.np_ptr(tp),
                                        n, C.byref(a), _buf.np_ptr(ts), n, C.byref(b), None))
        return tp[:a.value].copy(), ts[:b.value].copy()

    def delete_batch(self, ids):
        ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
        if ids_a.size:
            check(lib().sdb_index_delete_batch(self._h, ids_a.size, _buf.np_ptr(ids_a), None))

    def insert_batch(self, ids, vectors, round_size=0):
        k, vp, mem, shape = _buf.as_f32(vectors)
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
        check(lib().sdb_index_insert_batch(self._h, shape[0], _buf.np_ptr(ids_a), vp, mem, round_size,
                                           _buf.current_stream(mem)))

    # ---- Search vamana.go:278-310 -------------------------------------------------------------------
    def Search(self, options: SearchVectorVamanaOptions, filter=None):
        """Returns (set of node ids, [SearchResult]) like (*roaring64.Bitmap, []models.SearchResult)."""
        vec = np.ascontiguousarray(options.Vector, dtype=np.float32).reshape(1, -1)
        if vec.shape[1] != self.parameters.VectorSize:  # models/search.go:198-200 rejects this upstream
            raise SemaDBError(1, "query vector length %d does not match index %d" %
                              (vec.shape[1], self.parameters.VectorSize))
        ids, dists, counts, _ = self.search_batch(vec, options.Limit, options.SearchSize,
                                                  filters=None if filter is None else [filter])
        weight = np.float32(1) if options.Weight is None else np.float32(options.Weight)
        results = []
        for i in range(int(counts[0])):
            d = np.float32(dists[0, i])
            results.append(SearchResult(int(ids[0, i]), d, np.float32(-1) * d * weight))  # vamana.go:303
        return set(r.NodeId for r in results), results

    def search_batch(self, queries, limit, search_size, filters=None, trace=False, visit_cap=0, out=None):
        """nq queries at once.  numpy in -> numpy out (synchronous); torch CUDA in -> torch out, enqueued
        on the current stream.  Returns (ids [nq,limit] uint64, dists, counts, BatchTrace|None)."""
        k, qp, mem, shape = _buf.as_f32(queries)
        if len(shape) != 2 or shape[1] != self.parameters.VectorSize:
            raise SemaDBError(1, "query vector length must be %d" % self.parameters.VectorSize)
        nq = shape[0]
        f_off = f_ids = None
        bitmaps = None
        if isinstance(filters, FilterBitmaps):  # sdb_index_search_batch_bitmap
            bitmaps, filters = filters, None
            if bitmaps.word_offsets.size != nq + 1 or bitmaps.first_id.size != nq:
                raise SemaDBError(1, "one filter per query expected")
        if isinstance(filters, tuple):  # (offsets [nq + 1], ascending ids) already in the ABI's form
            f_off = np.ascontiguousarray(filters[0], dtype=np.uint64)
            f_ids = np.ascontiguousarray(filters[1], dtype=np.uint64)
            if f_off.size != nq + 1:
                raise SemaDBError(1, "one filter per query expected")
            if f_ids.size == 0:
                f_ids = np.zeros(1, dtype=np.uint64)
        elif filters is not None:
            if len(filters) != nq:
                raise SemaDBError(1, "one filter per query expected")
            flat, off = [], [0]
            for f in filters:
                flat.extend(sorted(int(v) for v in f))
                off.append(len(flat))
            f_off = np.array(off, dtype=np.uint64)
            f_ids = np.array(flat if flat else [0], dtype=np.uint64)
        if out is not None:  # caller-provided buffers: device tensors (ids int64 [nq,limit], dists f32, counts
            ids, dists, counts = out  # int32) for device queries, numpy arrays (e.g. pinned) for host queries
            if mem == MEM_DEVICE:
                idp, dp, cp = (C.c_void_p(t.data_ptr()) for t in out)
            else:
                idp, dp, cp = (_buf.np_ptr(t) for t in out)
        else:
            ids, idp = _buf.empty_like_mem(mem, (nq, limit), "uint64", self.device)
            dists, dp = _buf.empty_like_mem(mem, (nq, limit), "float32", self.device)
            counts, cp = _buf.empty_like_mem(mem, (nq,), "uint32", self.device)
        tr_struct, tr_out = None, None
        if trace:
            nd, ndp = _buf.empty_like_mem(mem, (nq,), "uint32", self.device)
            nh, nhp = _buf.empty_like_mem(mem, (nq,), "uint32", self.device)
            ne, nep = _buf.empty_like_mem(mem, (nq,), "uint32", self.device)
            vis, visp = (None, None)
            if visit_cap:
                vis, visp = _buf.empty_like_mem(mem, (nq, visit_cap), "uint64", self.device, zero=True)
            tr_struct = SearchTrace(ndp, nhp, nep, visp, visit_cap)
            tr_out = BatchTrace(nd, nh, ne, vis)
        if bitmaps is not None:
            check(lib().sdb_index_search_batch_bitmap(self._h, nq, qp, limit, search_size, _buf.np_ptr(bitmaps.first_id),
                                                      _buf.np_ptr(bitmaps.word_offsets), _buf.np_ptr(bitmaps.words), idp,
                                                      dp, cp, C.byref(tr_struct) if tr_struct is not None else None, mem,
                                                      _buf.current_stream(mem)))
            return ids, dists, counts, tr_out
        check(lib().sdb_index_search_batch(self._h, nq, qp, limit, search_size, _buf.np_ptr(f_off),
                                           _buf.np_ptr(f_ids), idp, dp, cp,
                                           C.byref(tr_struct) if tr_struct is not None else None, mem,
                                           _buf.current_stream(mem)))
        return ids, dists, counts, tr_out

    # ---- plainStore.DistanceFromFloat plain.go:76-85, batched ------------------------------------------
    def distance_batch(self, queries, cand_ids):
        k, qp, mem, shape = _buf.as_f32(queries)
        cand = np.ascontiguousarray(cand_ids, dtype=np.uint64)
        nq, nc = cand.shape
        out, op = _buf.empty_like_mem(mem, (nq, nc), "float32", self.device)
        check(lib().sdb_index_distance_batch(self._h, nq, qp, nc, _buf.np_ptr(cand), op, mem,
                                             _buf.current_stream(mem)))
        return out


def NewIndexVamana(name, params, bucket=None, **kw):
    """vamana.NewIndexVamana (vamana.go:54-81)."""
    return IndexVamana(name, params, bucket, **kw)
