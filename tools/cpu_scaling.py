"""How does the oracle's batch search scale with host threads on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)), "omp max:", orc.max_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
rng = np.random.default_rng(1)
n, d = 200000, 384
base = rng.standard_normal((n, 24)).astype(np.float32) @ rng.standard_normal((24, d)).astype(np.float32)
base /= np.linalg.norm(base, axis=1, keepdims=True)
# random regular graph is enough to time the walk
R = 64
edges = rng.integers(2, n + 1, size=(n, R)).astype(np.uint64)
ids = np.arange(1, n + 1, dtype=np.uint64)
offsets = (np.arange(n + 1) * R).astype(np.uint64)
o = orc.Index(d, "cosine", 64, 75, 1.2, impl=orc.IMPL_AVX2)
assert o.load(ids, base, offsets, edges.reshape(-1)) == 0
q = base[rng.integers(0, n, 8192)] + 0.01 * rng.standard_normal((8192, d)).astype(np.float32)
q = q.astype(np.float32)
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    nq = min(8192, 256 * t)
    t0 = time.perf_counter()
    o.search_batch(q[:nq], 10, 75, n_threads=t)
    dt = time.perf_counter() - t0
    print("threads %3d: %8.1f QPS (%.2fs)" % (t, nq / dt, dt), flush=True)
