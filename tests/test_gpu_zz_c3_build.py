"""C3 at BASELINE.json's size, edge for edge: the batched build of the 1M x 384 bench data against the oracle's
restatement of the round schedule (oracle/sdb_oracle.c insert_rounds; round_size = 1 is pinned to insert.go:16-68 by
tests/test_gpu_build.py).  All 1 000 000 rows by default; SDB_TEST_C3_ORACLE_ROWS shortens it for a quick run.

The oracle's side is ~2.5 minutes of host cores and touches no GPU: tests/conftest.py starts it on a thread of its own
when the session's tests have been collected, and this module sorts last, so that it runs under every other test and
is (nearly) through when it is joined here."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, L = 64, 75


def test_c3_build_equals_oracle_schedule(oracle):
    import torch
    from semadb_amd import vamana
    from tests import helpers
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench  # the synthetic data generator of the measured workload (SURVEY 8d seeds)
    rows = min(int(os.environ.get("SDB_TEST_C3_ORACLE_ROWS", 1_000_000)), int(os.environ.get("SDB_TEST_C2_ROWS", 1_000_000)))
    job = helpers.start_oracle_build(rows, 384, R, L)  # already running (conftest.py) unless this module runs alone
    base = bench.gen_rows(rows, 384, 20250620, "latent:24", "cuda:0")
    ix = vamana.NewIndexVamana("c3o", vamana.IndexVectorVamanaParameters(384, "cosine", L, R, 1.2), capacity=rows + 1)
    ix.set_start(bench.start_vector(384))
    ix.insert_batch(None, base)  # ids 2..rows+1, full-size rounds
    g_ids, _, g_off, g_e = ix.export(with_vectors=False)
    ix.close()
    del base
    torch.cuda.empty_cache()
    job.join()
    if job.error is not None:
        raise job.error
    o_ids, o_off, o_e = job.result
    assert len(g_ids) == rows + 1 and len(o_ids) == rows + 1  # the start node and every row, on both sides
    assert len(g_e) > 30 * rows and len(o_e) == len(g_e)
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(g_off, o_off), "degree sequence differs"
    assert np.array_equal(g_e, o_e), "edge lists differ"
