"""Host logic of bench.py that needs no GPU: how a timed region of K designs is split into batches."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_schedule_processes_exactly_the_requested_designs():
    """The resident configuration (4 x 32: four chunks in flight) does not depend on --steps: a region of K designs is K // 32 full chunks and one
    partial chunk (the order in which emagls_jobs_run cuts a list of equal-shape jobs), never a design more."""
    b = _bench()
    sch = b.schedule
    assert (b.SLOTS, b.BSZ) == (4, 32)
    for k in range(1, 300):
        s = sch(k)
        assert sum(s) == k and all(x == 32 for x in s[:-1]) and 1 <= s[-1] <= 32
    assert sch(20) == [20] and sch(128) == [32] * 4 and sch(5, 8) == [5] and sch(20, 8) == [8, 8, 4] and sch(20, 16) == [16, 4]


def test_gpus_flag_spawns_fresh_ranks(tmp_path):
    """`bench.py --gpus N` without WORLD_SIZE starts N children with the torch.distributed environment (gloo here: the
    children rendezvous on 127.0.0.1 and all-reduce their ranks), before the parent imports torch or touches HIP."""
    import sys
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert int(os.environ['LOCAL_RANK']) == r and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "dist.init_process_group('gloo', rank=r, world_size=w)\n"
        "t = torch.tensor([float(r + 1)]); dist.all_reduce(t)\n"
        "open(sys.argv[1] + '.%d' % r, 'w').write(str(t.item()))\n"
        "dist.destroy_process_group()\n")
    b = _bench()
    assert "torch" not in b.__dict__                       # the module itself never imports torch at import time
    rc = b.spawn_ranks(2, [str(tmp_path / "out")], script=str(child), timeout=120)
    assert rc == 0
    assert [(tmp_path / ("out.%d" % r)).read_text() for r in range(2)] == ["3.0", "3.0"]
    # a failing rank fails the launch and takes the others down instead of leaving them in the rendezvous
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nsys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)\n")
    assert b.spawn_ranks(2, [], script=str(bad), timeout=30) == 7


def test_gpus_flag_fails_loudly_without_the_gpus():
    """`--gpus 2` on a box with fewer GPUs must not silently measure one GPU (or none)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)
    # WORLD_SIZE that contradicts --gpus is refused as well
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


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
    assert [len(g) for g in lane_groups(["a"] * 17 + ["b"] * 3, max_batch=16)] == [9, 8, 3]


def test_config4_split_is_class_aware():
    """BASELINE config 4 as named: 256 radii on 2..10 cm over 8 ranks.  Sharding single jobs (round 2) left every rank with
    23-29 lane batches of 1-2 designs; whole padded lane batches go to ranks now.  Round 5: the batches are cut at equal COST
    (batch_cost at the batch's size and highest order; high orders: fewer designs per batch, never more than 32), so that the two
    batches a rank gets weigh the same whichever they are: the most loaded rank carries 1 % more than the least loaded one
    (batches of equal size: 8 %); every batch holds designs of neighbouring simulation-order classes laid out for the highest."""
    import numpy as np
    from emagls_amd.batch import batch_cost, lane_groups, padded_lane_batches, shard_jobs, shard_lane_batches, simulation_order
    radii = np.linspace(0.02, 0.10, 256)
    so = [simulation_order(4, 48000.0, r, raw=True) for r in radii]
    for max_batch, per, spread, padding, classes in ((16, 2, 1.03, 1.08, 5), (8, 4, 1.03, 1.04, 3)):
        batches = padded_lane_batches(so, max_batch) if max_batch != 16 else padded_lane_batches(so)
        assert sorted(j for idx, _ in batches for j in idx) == list(range(256)) and len(batches) == 256 // max_batch
        costs = [batch_cost(len(idx), pad) for idx, pad in batches]
        assert max(costs) / min(costs) < 1.12                              # equal-cost batches (to one design)
        for idx, pad in batches:
            own = [so[j] for j in idx]
            assert pad == max(own) and max(own) - min(own) <= classes and 1 <= len(idx) <= 32   # neighbouring classes only
        per_rank, load = shard_lane_batches(batches, 8)
        assert sorted(j for bl in per_rank for idx, _ in bl for j in idx) == list(range(256))
        assert all(len(bl) == per for bl in per_rank)
        assert max(load) / min(load) < spread
        # equal-size batches for comparison: the round-4 split
        _, load_eq = shard_lane_batches(padded_lane_batches(so, max_batch, balance=False), 8)
        assert max(load_eq) / min(load_eq) >= max(load) / min(load)
        if max_batch == 16:
            assert max(load_eq) > max(load)      # (the job list ends earlier)
        # the cost of padding: the batches are laid out for 3 % (7 %) more SH channels than the designs own
        assert sum((pad + 1) ** 2 * len(idx) for idx, pad in batches) / sum((s + 1) ** 2 for s in so) < padding
        if max_batch == 16:
            load16 = max(load)
    assert load16 < 0.8 * max(load)      # (lane batches of 16 do the job list in three quarters of the time of batches of 8)
    # what the per-job split did (kept for reference: shard_jobs + lane_groups is still right for equal-shape jobs)
    old = shard_jobs([(s + 1) ** 2 for s in so], 8)
    sizes = [len(g) for s in old for g in lane_groups([so[j] for j in s])]
    assert max(sizes) <= 2
    # ragged job counts of few classes: never a batch of one next to full ones
    assert [len(idx) for idx, _ in padded_lane_batches([5] * 9 + [7] * 8, 8, balance=False)] == [6, 6, 5]
    assert [len(idx) for idx, _ in padded_lane_batches([5] * 9 + [7] * 8, balance=False)] == [9, 8]
    assert sum(len(idx) for idx, _ in padded_lane_batches([5] * 9 + [7] * 8, 8)) == 17 and min(len(idx) for idx, _ in padded_lane_batches([5] * 9 + [7] * 8, 8)) >= 4
    assert [len(idx) for idx, _ in padded_lane_batches([4] * 3)] == [3] and padded_lane_batches([]) == []
    assert batch_cost(8, 44) > batch_cost(8, 9) > batch_cost(1, 9)


@pytest.mark.gpu
def test_bench_with_two_ranks_on_the_gpus_that_exist():
    """The N > 1 control flow of bench.py with real GPU work: `python bench.py --gpus 2` (no launcher: the parent spawns the
    ranks) in the test mode EMAGLS_BENCH_SHARED_GPU=1 -- the ranks share the GPUs that exist, the collectives (barrier, gather of
    every rank's filters to rank 0, max over the ranks' times) run on gloo with host tensors.  One JSON line from rank 0, exit code
    0, the line marked invalid as a measurement when ranks had to share a GPU.  Small batches: two resident sweeps of four
    designs fit the chip side by side."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, EMAGLS_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--slots", "1", "--batch", "4",
                        "--no-cpu-baseline", "--no-secondary", "--no-sh-roofline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 4 and d["value"] > 0 and d["scaling"] == "weak"
    import torch
    assert ("INVALID_as_a_measurement" in d) == (torch.cuda.device_count() < 2)
