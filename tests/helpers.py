"""Shared fixtures for parity tests: seeded data, oracle-built graphs."""
import numpy as np


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def start_vector(rng, d):
    """setupStartNode vamana.go:99-110: uniform(-1,1)^d scaled by 1/float32(sqrt(float64(sum)))."""
    v = (rng.random(d, dtype=np.float32) * np.float32(2) - np.float32(1)).astype(np.float32)
    s = np.float32(0)
    for x in v:
        s = np.float32(s + x * x)
    return (v * np.float32(1 / np.float32(np.sqrt(np.float64(s))))).astype(np.float32)


def build_oracle_index(orc, base, metric, R=32, L=50, alpha=1.2, seed=20250622, first_id=2):
    rng = np.random.default_rng(seed)
    d = base.shape[1]
    ix = orc.Index(d, metric, R, L, alpha, impl=orc.IMPL_AVX2 if orc.has_avx2() else orc.IMPL_ASM)
    ix.set_start(start_vector(rng, d))
    for i in range(base.shape[0]):
        rc = ix.insert(first_id + i, base[i])
        assert rc == 0, rc
    return ix


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_graph(g, o):
    """device index g and oracle index o hold the same nodes, vectors and edge lists, in order"""
    o_ids, o_v, o_off, o_e = o.export()
    g_ids, g_v, g_off, g_e = g.export()
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(g_off, o_off), "degree sequence differs"
    assert np.array_equal(g_e, o_e), "edge lists differ"
    assert np.array_equal(bits(g_v), bits(o_v))


# ---- the oracle's restatement of the full-size build (tests/test_gpu_zz_c3_build.py::test_c3_build_equals_oracle_schedule)
# About 2.5 minutes of host-core time that touches no GPU: it runs on a thread of its own (the C library releases the
# GIL) under the suite's other tests, which mostly wait for the device, and is joined by the test that compares.
import threading

_ORACLE_BUILD = None


class OracleBuild(threading.Thread):
    def __init__(self, rows, d, R, L, base=None):
        super().__init__(daemon=True)
        self.rows, self.d, self.R, self.L, self.base = rows, d, R, L, base
        self.result, self.error = None, None

    def run(self):
        try:
            import os
            import sys
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            if root not in sys.path:
                sys.path.insert(0, root)
            import bench  # the synthetic data generator of the measured workload (SURVEY 8d seeds)
            from oracle import oracle
            base = self.base
            if base is None:  # the same rows the module's fixture builds on (seeded generator, on the device)
                base = bench.gen_rows(self.rows, self.d, 20250620, "latent:24", "cuda:0")
            rows = base[:self.rows].cpu().numpy()
            self.base = None
            o = oracle.Index(self.d, "cosine", self.R, self.L, 1.2, impl=oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM)
            o.set_start(np.asarray(bench.start_vector(self.d), dtype=np.float32))
            rc = o.insert_rounds(np.arange(2, self.rows + 2, dtype=np.uint64), rows)
            if rc != 0:
                raise RuntimeError("insert_rounds rc=%d" % rc)
            o_ids, _, o_off, o_e = o.export(with_vectors=False)
            self.result = (o_ids, o_off, o_e)
        except BaseException as e:  # reported by the test that joins
            self.error = e


def start_oracle_build(rows, d, R, L, base=None):
    global _ORACLE_BUILD
    if _ORACLE_BUILD is None or _ORACLE_BUILD.rows != rows:
        _ORACLE_BUILD = OracleBuild(rows, d, R, L, base)
        _ORACLE_BUILD.start()
    return _ORACLE_BUILD
