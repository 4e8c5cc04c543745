"""Mirror of utils.KMeans (utils/kmeans.go:16-31,34-150) over the C ABI (K7)."""
import ctypes as C

import numpy as np

from . import _buf
from ._lib import MEM_HOST, check, lib


class KMeans:
    """utils.KMeans: fields K, MaxIter, Offset, VectorLen; Fit fills Centroids and Labels.

    first_idx replaces the reference's unseeded rand.IntN (kmeans.go:61).  alias=True keeps the
    reference's behaviour that centroids are views into X and the mean update overwrites those rows
    (kmeans.go:63,82,144): X is modified in place."""

    def __init__(self, K, MaxIter, Offset, VectorLen, first_idx=0, alias=True, device=0):
        self.K, self.MaxIter, self.Offset, self.VectorLen = K, MaxIter, Offset, VectorLen
        self.first_idx, self.alias, self.device = first_idx, alias, device
        self.Centroids = None
        self.Labels = None
        self.iters = 0

    def Fit(self, X):
        assert isinstance(X, np.ndarray) and X.dtype == np.float32 and X.flags.c_contiguous and X.ndim == 2
        n, stride = X.shape
        cent = np.zeros((self.K, self.VectorLen), dtype=np.float32)
        labels = np.zeros(n, dtype=np.uint8)
        iters = C.c_uint32(0)
        check(lib().sdb_kmeans_fit(_buf.np_ptr(X), n, stride, self.Offset, self.VectorLen, self.K, self.MaxIter,
                                   self.first_idx, 1 if self.alias else 0, _buf.np_ptr(cent), _buf.np_ptr(labels),
                                   C.byref(iters), MEM_HOST, self.device, None))
        self.Centroids, self.Labels, self.iters = cent, labels, iters.value
        return self
