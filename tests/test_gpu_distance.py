"""K1 parity (GPU): batched distances are bit-identical to the oracle's AVX2-order arithmetic."""
import numpy as np
import pytest

from tests.helpers import bits

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 7, 31, 32, 33, 64, 96, 100, 128, 129, 384, 385, 768, 1536, 4096]


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d", DIMS)
def test_distance_batch_bit_exact(oracle, metric, d):
    from semadb_amd import distance
    rng = np.random.default_rng(d * 7 + len(metric))
    nq, nc = 5, 77
    q = (rng.standard_normal((nq, d)) * 3).astype(np.float32)
    c = (rng.standard_normal((nc, d)) * 3).astype(np.float32)
    got = distance.distance_batch(metric, q, c)
    want = oracle.distance_matrix(q, c, metric, oracle.IMPL_ASM)
    assert np.array_equal(bits(got), bits(want))


def test_reference_vector_table():
    # distance/distance_test.go:9-21 through the GPU seam (TestASMdotProduct distance_amd64_test.go:12-27)
    from semadb_amd import distance
    table = [([0, 0, 0], [0, 0, 0], 0, 0), ([1, 1], [1, 1], 2, 0), ([1, 2, 3], [4, 5, 6], 32, 27),
             ([-1, -2, -3], [-4, -5, -6], 32, 27), ([-1, 2, 3], [4, -5, 6], 4, 83)]
    dot = distance.GetFloatDistanceFn("dot")
    l2 = distance.GetFloatDistanceFn("euclidean")
    cos = distance.GetFloatDistanceFn("cosine")
    for x, y, want_dot, want_l2 in table:
        assert dot(x, y) == np.float32(-want_dot)       # dotProductDistance distance.go:19-21
        assert l2(x, y) == np.float32(want_l2)
        assert cos(x, y) == np.float32(1 - want_dot)    # cosineDistance distance.go:23-25


def test_unknown_metric_is_an_error():
    from semadb_amd import distance, SemaDBError
    with pytest.raises(SemaDBError):
        distance.GetFloatDistanceFn("manhattan")


def test_adversarial_values(oracle):
    """denormals, huge magnitudes, cancellation, signed zeros: still bit-identical."""
    from semadb_amd import distance
    rng = np.random.default_rng(5)
    d = 384
    q = rng.standard_normal((4, d)).astype(np.float32)
    c = rng.standard_normal((6, d)).astype(np.float32)
    q[0] *= np.float32(1e-38)      # denormal products
    c[0] *= np.float32(1e-3)
    q[1] *= np.float32(1e18)       # overflow to inf in L2
    c[1] *= np.float32(1e18)
    c[2] = q[2]                    # exact zero distance
    c[3] = -q[2]
    q[3, ::2] = 0.0
    c[4, 1::2] = -0.0
    for metric in ("euclidean", "cosine", "dot"):
        got = distance.distance_batch(metric, q, c)
        want = oracle.distance_matrix(q, c, metric, oracle.IMPL_ASM)
        assert np.array_equal(bits(got), bits(want)), metric


def test_device_memory_path(oracle):
    import torch
    from semadb_amd import distance
    rng = np.random.default_rng(9)
    q = rng.standard_normal((3, 384)).astype(np.float32)
    c = rng.standard_normal((130, 384)).astype(np.float32)
    got = distance.distance_batch("cosine", torch.from_numpy(q).cuda(), torch.from_numpy(c).cuda())
    torch.cuda.synchronize()
    want = oracle.distance_matrix(q, c, "cosine", oracle.IMPL_ASM)
    assert np.array_equal(bits(got.cpu().numpy()), bits(want))


# ---- the row-reuse kernels (csrc/distance_tile.hip): matrix cores for dot / cosine, packed FMAs for euclidean ----
TILE_DIMS = [32, 36, 64, 96, 100, 128, 160, 300, 384, 416, 608, 640, 768, 1024, 1056, 1536, 4096]


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
@pytest.mark.parametrize("d", TILE_DIMS)
def test_tiled_distance_bit_exact(oracle, metric, d):
    """more queries than one operand group, candidate counts that leave ragged tiles, rows with and without a tail
    chain (d % 32 = 4, 12), every register-block count of the matrix-core kernel, euclidean tiles of fewer than 64
    rows (d > 608), and the shapes past the tile kernels' reach, which fall back to the first kernel -- all
    bit-identical to the oracle"""
    from semadb_amd import distance
    rng = np.random.default_rng(d * 11 + len(metric))
    for nq, nc in [(37, 1003), (16, 64), (2, 130)]:
        q = (rng.standard_normal((nq, d)) * 2).astype(np.float32)
        c = (rng.standard_normal((nc, d)) * 2).astype(np.float32)
        got = distance.distance_batch(metric, q, c)
        want = oracle.distance_matrix(q, c, metric, oracle.IMPL_ASM)
        assert np.array_equal(bits(got), bits(want)), (nq, nc)


@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("d", [64, 384, 416, 1024])
def test_tiled_distance_query_blocks(oracle, metric, d):
    """the two matrix-core kernels meet at 256 queries: up to there the queries stay in registers, 64 per workgroup
    (one, two, four query blocks, a ragged last one) and the candidates stream in spans of 512 (one span, a ragged
    second and third); above, the candidate rows stay in registers and the query groups stream.  d = 416 has a tail
    (always the second kernel), d = 1 024 is the longest row of the first"""
    from semadb_amd import distance
    rng = np.random.default_rng(d * 13 + len(metric))
    for nq, nc in [(64, 512), (65, 600), (130, 1100), (256, 513), (257, 520), (300, 1030)]:
        q = (rng.standard_normal((nq, d)) * 2).astype(np.float32)
        c = (rng.standard_normal((nc, d)) * 2).astype(np.float32)
        got = distance.distance_batch(metric, q, c)
        want = oracle.distance_matrix(q, c, metric, oracle.IMPL_ASM)
        assert np.array_equal(bits(got), bits(want)), (nq, nc)


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
def test_tiled_distance_device_memory_and_special_values(oracle, metric):
    import torch
    from semadb_amd import distance
    rng = np.random.default_rng(77)
    d = 384
    q = rng.standard_normal((64, d)).astype(np.float32)
    c = rng.standard_normal((4096, d)).astype(np.float32)
    q[0] *= np.float32(1e-38)
    c[0] *= np.float32(1e-3)      # denormal products
    q[1] *= np.float32(1e18)
    c[1] *= np.float32(1e18)      # overflow
    c[2] = q[2]
    c[3] = -q[2]                  # cancellation
    q[3, ::2] = 0.0
    c[4, 1::2] = -0.0             # signed zeros
    want = oracle.distance_matrix(q, c, metric, oracle.IMPL_ASM)
    for nc in (4096, 4095):       # 16-byte stores, then the scalar store path (nc % 4 != 0)
        got = distance.distance_batch(metric, torch.from_numpy(q).cuda(), torch.from_numpy(c[:nc].copy()).cuda())
        torch.cuda.synchronize()
        g = got.cpu().numpy()
        w = want[:, :nc]
        same = bits(g) == bits(w)
        nan = np.isnan(g) & np.isnan(w)  # inf - inf: a NaN wherever the reference has one
        assert (same | nan).all(), nc
