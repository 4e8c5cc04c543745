"""Measurement only (library built with -DSDB_BACK_PROFILE): cycles spent by k_backedges waves by number of requests
per target, summed over a 1M x 384 build (stat slots 11..15)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, bench
from semadb_amd import _buf
from semadb_amd._lib import lib, check
class A: metric, search_size, degree_bound, alpha = "cosine", 75, 64, 1.2
base = bench.gen_rows(1000000, 384, 20250620, "latent:24", "cuda:0")
ix, bs = bench.build_index(A, base, 0)
out = np.zeros(16, dtype=np.uint64)
check(lib().sdb_index_build_stats(ix._h, _buf.np_ptr(out), out.size))
print("build_s", bs); print([int(x) for x in out])
