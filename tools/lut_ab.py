"""Time of the product quantizer's query-table kernel alone (k_pq_lut_*; product.go:255-263), for the library named by
SEMADB_AMD_LIB (default: the built one): 1 024 queries x 768 floats, K = 256, M = 8 and 192, euclidean and dot.  The table
is built inside sdb_pq_lut_distance (with 64 code rows its second kernel is noise); time = HIP events over 50 calls, and a
SHA-1 of the distances so that variants can be seen to agree.  Under `rocprofv3 --kernel-trace --stats` the kernel's own
average is in the trace."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semadb_amd import vectorstore as vs
out = {"lib": os.environ.get("SEMADB_AMD_LIB", "default")}
d, nq, K = 768, 1024, 256
g = torch.Generator(device="cuda:0").manual_seed(5)
q = torch.randn(nq, d, generator=g, device="cuda:0")
for metric in ("euclidean", "dot"):
    for M in (8, 192):
        pq = vs.ProductQuantizer(metric, vs.ProductQuantizerParameters(K, M, 10000), d)
        pq.set_codebook(np.random.default_rng(M).standard_normal((M, K, d // M)).astype(np.float32))
        codes = torch.randint(0, K, (64, M), dtype=torch.uint8, device="cuda:0", generator=g)
        for _ in range(5):
            o = pq.lut_distance(q, codes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            o = pq.lut_distance(q, codes)
        e1.record()
        torch.cuda.synchronize()
        out["%s_M%d" % (metric, M)] = {"us_per_call": round(e0.elapsed_time(e1) * 1000 / 50, 2),
                                      "sha1": hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]}
        pq.close()
print(json.dumps(out))
