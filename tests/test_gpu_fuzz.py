"""Randomised differential trials (tools/fuzz_parity.py: device through the C ABI vs the oracle over random shapes,
metrics, parameters, data with ties and duplicates, build schedules and write sequences).  The numbered trials are
the ones that exposed defects when the soak was first run:
  seed 1 trial 0, 23   a delete leaves more stragglers than the start node's 64-entry row holds (prune.go:131-151:
                       the reference's start node has no bound) -- now an overflow list, exact;
  seed 1 trial 88      a store switched to the product quantizer kept per-row prune state (clean prefix, cached
                       distances) that only holds for the distance function it was made with.
A short fresh soak runs next to them."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import fuzz_parity
    return fuzz_parity


@pytest.mark.parametrize("seed,t", [(1, 0), (1, 23), (1, 88)])
def test_fuzz_regressions(oracle, monkeypatch, seed, t):
    monkeypatch.setenv("SDB_BIG_MIN", "512")  # trial() sets its own; monkeypatch restores the environment afterwards
    _fuzz().trial(np.random.default_rng([seed, t]), t)


def test_fuzz_short_soak(oracle, monkeypatch):
    monkeypatch.setenv("SDB_BIG_MIN", "512")
    fz = _fuzz()
    for t in range(40):
        fz.trial(np.random.default_rng([20251002, t]), t)
