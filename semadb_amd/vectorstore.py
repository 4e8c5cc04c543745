"""Mirror of the product quantizer of shard/vectorstore/product.go over the C ABI (K5, K6, K8).

The plain store's device form is the index slab itself (vamana.IndexVamana.load / distance_batch);
this module carries the quantizer: newProductQuantizer (product.go:42-88), Fit (:175-236), encode
(:136-159), DistanceFromFloat (:238-277) and DistanceFromPoint (:279-305) in batched form.
"""
import ctypes as C

import numpy as np

from . import _buf
from ._lib import MEM_HOST, METRICS, SemaDBError, check, lib


class ProductQuantizerParameters:
    """models.ProductQuantizerParameters (models/quantizer.go:51-63)"""

    def __init__(self, NumCentroids, NumSubVectors, TriggerThreshold=10000):
        self.NumCentroids, self.NumSubVectors, self.TriggerThreshold = NumCentroids, NumSubVectors, TriggerThreshold


class ProductQuantizer:
    def __init__(self, distFnName, params: ProductQuantizerParameters, vectorLen, device=0):
        if distFnName not in METRICS:  # product.go:48-50
            raise SemaDBError(1, "distance function %s not supported for product quantisation" % distFnName)
        self.params, self.vectorLen, self.device = params, vectorLen, device
        h = C.c_void_p()
        check(lib().sdb_pq_create(vectorLen, METRICS[distFnName], params.NumSubVectors, params.NumCentroids, device,
                                  C.byref(h)))
        self._h = h
        self.M, self.K = params.NumSubVectors, params.NumCentroids
        self.subVectorLen = vectorLen // params.NumSubVectors

    def close(self):
        if getattr(self, "_h", None):
            lib().sdb_pq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Fit(self, X, first_idx, alias=True):
        """productQuantizer.Fit (product.go:175-236) over the rows of X in the given order; returns codes."""
        assert isinstance(X, np.ndarray) and X.dtype == np.float32 and X.flags.c_contiguous
        fi = np.ascontiguousarray(first_idx, dtype=np.uint32)
        assert fi.size == self.M
        codes = np.zeros((X.shape[0], self.M), dtype=np.uint8)
        check(lib().sdb_pq_fit(self._h, _buf.np_ptr(X), X.shape[0], _buf.np_ptr(fi), 1 if alias else 0,
                               _buf.np_ptr(codes), MEM_HOST, None))
        return codes

    def set_codebook(self, flat_centroids):
        fc = np.ascontiguousarray(flat_centroids, dtype=np.float32)
        assert fc.size == self.M * self.K * self.subVectorLen
        check(lib().sdb_pq_set_codebook(self._h, _buf.np_ptr(fc), MEM_HOST))

    def codebook(self):
        fc = np.zeros((self.M, self.K, self.subVectorLen), dtype=np.float32)
        cd = np.zeros((self.M, self.K, self.K), dtype=np.float32)
        check(lib().sdb_pq_get_codebook(self._h, _buf.np_ptr(fc), _buf.np_ptr(cd)))
        return fc, cd

    def encode(self, vectors):
        k, vp, mem, shape = _buf.as_f32(vectors)
        codes, cp = _buf.empty_like_mem(mem, (shape[0], self.M), "uint8", self.device)
        check(lib().sdb_pq_encode(self._h, vp, shape[0], cp, mem, _buf.current_stream(mem)))
        return codes

    def lut_distance(self, queries, codes):
        """DistanceFromFloat batched: out[q, c] (product.go:250-277)"""
        k, qp, mem, qs = _buf.as_f32(queries)
        if mem == MEM_HOST:
            cd = np.ascontiguousarray(codes, dtype=np.uint8)
            cptr, nc = _buf.np_ptr(cd), cd.shape[0]
        else:
            cd = codes.contiguous()
            cptr, nc = C.c_void_p(cd.data_ptr()), cd.shape[0]
        out, op = _buf.empty_like_mem(mem, (qs[0], nc), "float32", self.device)
        check(lib().sdb_pq_lut_distance(self._h, qp, qs[0], cptr, nc, op, mem, _buf.current_stream(mem)))
        return out

    def sym_distance(self, codes_x, codes_y):
        """DistanceFromPoint batched over pairs (product.go:293-304)"""
        cx = np.ascontiguousarray(codes_x, dtype=np.uint8)
        cy = np.ascontiguousarray(codes_y, dtype=np.uint8)
        out = np.zeros(cx.shape[0], dtype=np.float32)
        check(lib().sdb_pq_sym_distance(self._h, _buf.np_ptr(cx), _buf.np_ptr(cy), cx.shape[0], _buf.np_ptr(out),
                                        MEM_HOST, None))
        return out


def attach(index, pq: ProductQuantizer, ids=None, codes=None):
    """Switch a vamana.IndexVamana to the fitted quantizer (what a fitted productQuantizer store does).
    Every stored vector is encoded; `ids`/`codes` then overwrite the centroid ids of those points -- the
    k-means labels Fit leaves on its training points (product.go:216-218) or codes read back from a bucket."""
    check(lib().sdb_index_attach_pq(index._h, pq._h, None))
    index._pq = pq  # keep alive
    if ids is not None:
        set_codes(index, ids, codes)


def set_codes(index, ids, codes):
    ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
    cd = np.ascontiguousarray(codes, dtype=np.uint8)
    assert cd.shape == (ids_a.size, index._pq.M)
    check(lib().sdb_index_set_codes(index._h, ids_a.size, _buf.np_ptr(ids_a), _buf.np_ptr(cd)))


def get_codes(index, ids):
    ids_a = np.ascontiguousarray(ids, dtype=np.uint64)
    cd = np.zeros((ids_a.size, index._pq.M), dtype=np.uint8)
    check(lib().sdb_index_get_codes(index._h, ids_a.size, _buf.np_ptr(ids_a), _buf.np_ptr(cd)))
    return cd


QuantizerNone, QuantizerProduct = "none", "product"


class Quantizer:
    """models.Quantizer (models/quantizer.go:5-28); the binary quantizer is out of scope (hamming/jaccard)"""

    def __init__(self, Type=QuantizerNone, Product: ProductQuantizerParameters = None):
        self.Type, self.Product = Type, Product


def New(params, distFnName, vectorLength, device=0):
    """vectorstore.New (vectorstore.go:47-96): None for the plain store (the index slab), a ProductQuantizer
    for `product`."""
    if distFnName not in METRICS:
        raise SemaDBError(1, "unknown float32 distance function: %s" % distFnName)
    if params is None or params.Type == QuantizerNone:
        return None
    if params.Type != QuantizerProduct:
        raise SemaDBError(1, "unknown vector store type %s" % params.Type)
    if params.Product is None:
        raise SemaDBError(1, "product quantizer parameters are nil")
    pp = params.Product
    if vectorLength % pp.NumSubVectors != 0:  # product.go:44-46
        raise SemaDBError(1, "vector length %d must be divisible by num subvectors %d" % (vectorLength, pp.NumSubVectors))
    if pp.NumCentroids > 256:  # product.go:63-65
        raise SemaDBError(1, "number of centroids %d cannot exceed 256" % pp.NumCentroids)
    return ProductQuantizer(distFnName, pp, vectorLength, device)
