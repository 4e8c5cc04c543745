"""gather probe with K2-like disturbances: idle time between chunks, and chunks whose rows are shared by all waves"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe", "libgather_probe.so"))
lib.gather_probe.restype = ctypes.c_float
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.gather_probe_config.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
n, d = 1000000, 384
slab = torch.randn(n, d, device="cuda:0")
sink = torch.zeros(1 << 20, device="cuda:0")
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
out = {}
for think in (0, 50, 100, 200):   # x64 clocks: 0, 1.3, 2.7, 5.3 us at 2.4 GHz
    for shared in (0, 10, 30, 60):
        assert lib.gather_probe_config(think, shared) == 0
        ms = min(lib.gather_probe(slab.data_ptr(), n, d, 3, 1024, 128, 16, sink.data_ptr(), stream) for _ in range(4))
        out["think=%d shared=%d%%" % (think, shared)] = round(1024 * 128 * 32 * d * 4 / ms / 1e6, 1)
print(json.dumps(out, indent=1))
