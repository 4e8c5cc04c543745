"""BASELINE.json configs[0]: "vectorVamana 10k x 128 f32 cosine on CPU reference (distance/puredist.go +
asm path), no GPU -- plumbing".  The oracle runs the reference algorithm end to end on the CPU with both
arithmetic paths the reference has (distance_amd64.go:19-27 picks the assembly when AVX2+FMA are present,
puredist.go otherwise) and both reach the same neighbours; their distances differ in the last bits, which
is why the GPU path follows the assembly's summation order."""
import numpy as np

from tests.helpers import start_vector


def _latent_rows(rng, n, d, k=16):
    w = rng.standard_normal((k, d)).astype(np.float32)
    x = rng.standard_normal((n, k)).astype(np.float32) @ w + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def test_c1_cpu_reference_plumbing(oracle):
    rng = np.random.default_rng(20250620)
    n, d, nq, k = 10000, 128, 200, 10
    allrows = _latent_rows(rng, n + nq, d)
    base, queries = allrows[:n], allrows[n:]
    sv = start_vector(np.random.default_rng(20250622), d)
    impl_asm = oracle.IMPL_AVX2 if oracle.has_avx2() else oracle.IMPL_ASM
    # build once on the assembly path (R = 32, L = 50 keep the CPU suite short), search on both paths
    o = oracle.Index(d, "cosine", 32, 50, 1.2, impl=impl_asm)
    o.set_start(sv)
    for i in range(n):
        assert o.insert(i + 2, base[i]) == 0
    ids, vecs, off, edges = o.export()
    assert len(ids) == n + 1 and int(np.diff(off.astype(np.int64)).max()) <= 32
    p = oracle.Index(d, "cosine", 32, 50, 1.2, impl=oracle.IMPL_PURE)
    assert p.load(ids, vecs, off, edges) == 0
    a_ids, a_d, _, _, _, _ = o.search_batch(queries, k, 75)
    p_ids, p_d, _, _, _, _ = p.search_batch(queries, k, 75)
    truth = np.argsort(-(queries @ base.T), axis=1)[:, :k] + 2
    rec_a = np.mean([len(set(a_ids[i]) & set(truth[i])) for i in range(nq)]) / k
    rec_p = np.mean([len(set(p_ids[i]) & set(truth[i])) for i in range(nq)]) / k
    assert rec_a >= 0.95 and rec_p >= 0.95
    same = np.mean([np.array_equal(a_ids[i], p_ids[i]) for i in range(nq)])
    assert same >= 0.9  # same neighbours almost always...
    assert np.any(a_d.view(np.uint32) != p_d.view(np.uint32))  # ...but not the same bits
    assert np.allclose(a_d, p_d, rtol=1e-5, atol=1e-6)
