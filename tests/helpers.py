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
