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


def test_lane_groups_for_a_radius_sweep():
    """BASELINE config 4: 256 radii on 2..10 cm; the lane batches of a rank hold radii of one simulation order each."""
    import numpy as np
    from emagls_amd.batch import lane_groups, shard_jobs, simulation_order
    radii = np.linspace(0.02, 0.10, 256)
    so = [simulation_order(4, 48000.0, r) for r in radii]
    assert so[0] == 9 and so[-1] == 44 and simulation_order(4, 48000.0, 0.001) == 4
    shards = shard_jobs([(s + 1) ** 2 for s in so], 8)
    assert sorted(j for s in shards for j in s) == list(range(256))
    loads = [sum((so[j] + 1) ** 2 for j in s) for s in shards]
    assert max(loads) / min(loads) < 1.02                       # longest-processing-time keeps the ranks within 2 %
    for s in shards:
        groups = lane_groups([so[j] for j in s])
        assert sorted(i for g in groups for i in g) == list(range(len(s)))
        for g in groups:
            assert 1 <= len(g) <= 8 and len({so[s[i]] for i in g}) == 1
    assert [len(g) for g in lane_groups(["a"] * 9 + ["b"] * 3)] == [5, 4, 3]
