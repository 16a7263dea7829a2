"""Host logic of bench.py that needs no GPU: how a timed region of K designs is split into batches."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plan_batches_covers_the_timed_region_without_waste():
    pb = _bench().plan_batches
    for k in range(1, 300):
        nb, bsz = pb(k)
        assert 1 <= nb <= 4 and 1 <= bsz <= 8
        j = nb * bsz
        if k <= 32:
            assert k <= j < k + nb       # one round, fewer than one spare design per batch
        else:
            assert j == 32
    assert pb(128, 16, 8) == (2, 8) and pb(8, 1, 1) == (1, 1) and pb(5, 32, 8) == (1, 5) and pb(64, 0, 4) == (8, 4)
