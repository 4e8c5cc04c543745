"""Workload for PMC comparisons between the search kernel and the dependency-free gather probe
(tools/probe/gather_probe.hip): builds the bench index, then launches 3 search batches and 3 probe launches
that read the same number of rows of the same size."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from semadb_amd import vamana

here = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(here, "probe", "libgather_probe.so"))
probe.gather_probe.restype = ctypes.c_float
probe.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                               ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
n, d, nq = int(os.environ.get("PMC_N", 1000000)), 384, 1024
dev = "cuda:0"
base = bench.gen_rows(n, d, 20250620, "latent:24", dev)
queries = bench.gen_rows(3 * nq, d, 20250621, "latent:24", dev).view(3, nq, d)
ix = vamana.NewIndexVamana("pmc", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
torch.cuda.synchronize()
for b in range(3):
    ix.search_batch(queries[b], 10, 75)
    torch.cuda.synchronize()
sink = torch.zeros(1 << 20, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ms = probe.gather_probe(base.data_ptr(), n, d, d // 128, 1024, 128, 16, sink.data_ptr(), stream)
    print("probe ms", ms, "GB/s", 1024 * 128 * 32 * d * 4 / ms / 1e6)
