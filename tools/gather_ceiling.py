"""Practical ceiling for the search kernel's access shape: random slab rows, half-wave per row, no dependencies
(tools/probe/gather_probe.hip).  Prints achieved GB/s for the C2 (d = 384) and C4 (d = 768) row sizes at several
occupancies, next to a plain streaming read of the same slab."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe", "libgather_probe.so"))
lib.gather_probe.restype = ctypes.c_float
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
out = {}
sink = torch.zeros(1 << 20, device=dev)
shapes = ((384, 1000000), (768, 1000000), (128, 2000000))
if len(sys.argv) > 1 and sys.argv[1] == "--big":  # the C5-rank and C4 slabs: how far the ceiling itself falls with size
    shapes = ((384, 4000000), (384, 12500000), (768, 10000000))
for d, n in shapes:
    ng = d // 128
    slab = torch.empty(n, d, device=dev)
    for i in range(0, n, 1000000):
        slab[i:i + 1000000].normal_()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for waves in (1024, 2048, 4096, 16384):
        for U in (4, 16):
            iters = max(1, (4096 * 1024) // (waves * 2 * U))  # ~4M rows per launch, like one C2 batch
            lib.gather_probe(slab.data_ptr(), n, d, ng, waves, iters, U, sink.data_ptr(), stream)
            ms = min(lib.gather_probe(slab.data_ptr(), n, d, ng, waves, iters, U, sink.data_ptr(), stream)
                     for _ in range(5))
            rows = waves * iters * 2 * min(U, 8 if ng == 6 else U)
            res["waves=%d U=%d" % (waves, U)] = round(rows * d * 4 / ms / 1e6, 1)
    # streaming read of the same bytes for comparison
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    slab.sum()
    e0.record()
    for _ in range(5):
        slab.sum()
    e1.record()
    torch.cuda.synchronize()
    res["torch.sum stream"] = round(slab.numel() * 4 * 5 / e0.elapsed_time(e1) / 1e6, 1)
    out["d=%d n=%d GB/s" % (d, n)] = res
    del slab
print(json.dumps(out, indent=1))
