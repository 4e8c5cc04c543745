"""Mirror of the reference's distance package (distance/distance.go).

GetFloatDistanceFn(name) returns a FloatDistFunc(x, y) -> float32, exactly the reference's seam
(distance/distance.go:11,70-83), computed by the K1 HIP kernel with the AVX2 assembly's summation
order (distance/asm/dot.s, euclidean.s).  distance_batch is the batched form the GPU wants.
"""
import ctypes as C

import numpy as np

from . import _buf
from ._lib import METRICS, SemaDBError, check, lib

DistanceEuclidean, DistanceCosine, DistanceDot = "euclidean", "cosine", "dot"


def distance_batch(name, queries, candidates, device=0):
    """out[q, c] = dist(queries[q], candidates[c]); numpy in -> numpy out, torch CUDA in -> torch out."""
    if name not in METRICS:
        raise SemaDBError(1, "unknown float32 distance function: %s" % name)  # distance.go:81
    qk, qp, qm, qs = _buf.as_f32(queries)
    ck, cp, cm, cs = _buf.as_f32(candidates)
    if qm != cm:
        raise SemaDBError(1, "queries and candidates must live in the same memory space")
    if len(qs) != 2 or len(cs) != 2 or qs[1] != cs[1]:
        raise SemaDBError(1, "vector length mismatch: %r vs %r" % (qs, cs))
    out, op = _buf.empty_like_mem(qm, (qs[0], cs[0]), "float32", device)
    check(lib().sdb_distance_batch(METRICS[name], qs[1], qp, qs[0], cp, cs[0], op, qm, device,
                                   _buf.current_stream(qm)))
    return out


def GetFloatDistanceFn(name, device=0):
    """distance.GetFloatDistanceFn (distance/distance.go:70-83)."""
    if name not in METRICS:
        raise SemaDBError(1, "unknown float32 distance function: %s" % name)

    def dist_fn(x, y):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(1, -1)
        y = np.ascontiguousarray(y, dtype=np.float32).reshape(1, -1)
        # like asm.Dot, the length comes from x only (dot.s:10); y must be at least as long
        y = y[:, :x.shape[1]]
        return np.float32(distance_batch(name, x, y, device)[0, 0])

    return dist_fn
