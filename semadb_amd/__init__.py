"""semadb_amd -- MI355X-native Vamana ANN search path for SemaDB.

Host-side mirror of the reference's Go interfaces for the hot path, sitting directly on the C ABI
(include/semadb_amd.h -> libsemadb_amd.so, hand-written HIP for gfx950):

    semadb_amd.distance    <-> distance/             (GetFloatDistanceFn, FloatDistFunc)
    semadb_amd.vamana      <-> shard/index/vamana/   (IndexVamana.Search / InsertUpdateDelete)
    semadb_amd.vectorstore <-> shard/vectorstore/    (plain store, product quantizer)
    semadb_amd.kmeans      <-> utils/kmeans.go
    semadb_amd.cluster     <-> cluster/actions.go    (per-shard limit, top-k merge, RCCL all-gather)

Nothing here computes on the CPU and nothing imports the oracle: without the built library the
import fails, without a GPU every call raises.
"""
from . import _lib
from ._lib import SemaDBError, device_count

__all__ = ["_lib", "SemaDBError", "device_count"]
