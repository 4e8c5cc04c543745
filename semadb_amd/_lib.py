"""ctypes loader for libsemadb_amd.so (the C ABI declared in include/semadb_amd.h).

There is no CPU fallback anywhere in this package: if the shared library is missing the import of
anything that needs it raises, and every compute entry point fails with SDB_ERR_DEVICE when no
MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SEMADB_AMD_LIB") or os.path.join(_HERE, "libsemadb_amd.so")

SDB_OK = 0
MEM_HOST, MEM_DEVICE = 0, 1
METRICS = {"euclidean": 0, "cosine": 1, "dot": 2}
STARTID = 1

f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class IndexParams(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("search_size", C.c_uint32),
                ("degree_bound", C.c_uint32), ("alpha", C.c_float), ("device", C.c_int32),
                ("capacity", C.c_uint64), ("strict", C.c_uint32)]


class SearchTrace(C.Structure):
    _fields_ = [("n_dist", C.c_void_p), ("n_hop", C.c_void_p), ("n_edges", C.c_void_p),
                ("visit_ids", C.c_void_p), ("visit_cap", C.c_uint32)]


# name -> (restype, argtypes); one entry per function declared in include/semadb_amd.h
SIGNATURES = {
    "sdb_last_error": (C.c_char_p, []),
    "sdb_abi_version": (C.c_int, []),
    "sdb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sdb_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "sdb_host_free": (C.c_int, [C.c_void_p]),
    "sdb_distance_batch": (C.c_int, [C.c_int, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdb_index_create": (C.c_int, [C.POINTER(IndexParams), C.POINTER(C.c_void_p)]),
    "sdb_index_destroy": (C.c_int, [C.c_void_p]),
    "sdb_index_set_start": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "sdb_index_load": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sdb_index_insert_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32,
                                         C.c_void_p]),
    "sdb_index_delete_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "sdb_index_begin_write": (C.c_int, [C.c_void_p]),
    "sdb_index_commit": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdb_index_abort_write": (C.c_int, [C.c_void_p]),
    "sdb_index_version_diff": (C.c_int, [C.c_void_p, u64p]),
    "sdb_index_edge_scan": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_void_p,
                                      C.c_uint64, u64p, C.c_void_p]),
    "sdb_index_search_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SearchTrace),
                                         C.c_int, C.c_void_p]),
    "sdb_index_search_batch_bitmap": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(SearchTrace), C.c_int, C.c_void_p]),
    "sdb_index_distance_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p]),
    "sdb_index_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64]),
    "sdb_index_sketch_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "sdb_index_build_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "sdb_index_get_vectors": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdb_index_exists_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "sdb_index_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "sdb_index_last_search_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sdb_index_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, u32p]),
    "sdb_index_flat_search": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sdb_index_set_vectors": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]),
    "sdb_index_remove_vectors": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "sdb_index_size_in_memory": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "sdb_index_stats": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "sdb_index_compact": (C.c_int, [C.c_void_p]),
    "sdb_index_row_usage": (C.c_int, [C.c_void_p, u64p, u64p]),
    "sdb_index_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdb_topk_merge": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdb_shard_limit": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
    "sdb_cluster_unique_id": (C.c_int, [C.c_void_p]),
    "sdb_cluster_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "sdb_cluster_create_local": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "sdb_cluster_destroy": (C.c_int, [C.c_void_p]),
    "sdb_cluster_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdb_cluster_block_layout": (C.c_int, [C.c_uint64, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sdb_cluster_allgather_merge": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sdb_cluster_search_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32,
                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p]),
    "sdb_cluster_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdb_cluster_synchronize": (C.c_int, [C.c_void_p]),
    "sdb_cluster_next_ticket": (C.c_int, [C.c_void_p, u64p]),
    "sdb_cluster_set_deadline": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sdb_cluster_skip_ticket": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
    "sdb_cluster_transport": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "sdb_cluster_stamp_block": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                          C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "sdb_cluster_merge_gathered": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sdb_kmeans_fit": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, u32p, C.c_int, C.c_int, C.c_void_p]),
    "sdb_pq_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "sdb_pq_destroy": (C.c_int, [C.c_void_p]),
    "sdb_pq_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                             C.c_void_p]),
    "sdb_pq_set_codebook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "sdb_pq_get_codebook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdb_pq_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]),
    "sdb_pq_lut_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                      C.c_int, C.c_void_p]),
    "sdb_pq_sym_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    "sdb_index_attach_pq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdb_index_union_prune": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]),
    "sdb_index_set_codes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "sdb_index_get_codes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
}


class SemaDBError(RuntimeError):
    """An error returned through the C ABI; mirrors the Go `error` values of the reference."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "semadb_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C semadb_amd/csrc` (there is no CPU fallback)" % SO_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both
        # own the GPU, so load torch's first and let our NEEDED libamdhip64.so.7 resolve to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != SDB_OK:
        msg = lib().sdb_last_error()
        raise SemaDBError(rc, msg.decode("utf-8", "replace") if msg else "semadb_amd error %d" % rc)


def device_count():
    n = C.c_int(0)
    rc = lib().sdb_device_count(C.byref(n))
    return n.value if rc == SDB_OK else 0
