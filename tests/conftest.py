import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


def pytest_collection_finish(session):
    """The longest single piece of the GPU suite is host work: the oracle's restatement of the 1M-row build that
    tests/test_gpu_zz_c3_build.py::test_c3_build_equals_oracle_schedule (the module that sorts last) compares the device's graph with.  When that test
    is part of the run (and a GPU is there to generate the rows on), the oracle starts now, on its own thread."""
    if not any("test_c3_build_equals_oracle_schedule" in it.nodeid for it in session.items):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from tests import helpers
        rows = min(int(os.environ.get("SDB_TEST_C3_ORACLE_ROWS", 1_000_000)), int(os.environ.get("SDB_TEST_C2_ROWS", 1_000_000)))
        helpers.start_oracle_build(rows, 384, 64, 75)
    except Exception as e:  # the module's own fixture starts it then
        print("oracle build not started early: %r" % (e,))
