"""Host-side checks that need neither the GPU nor the library."""



def test_wave_fft_lds_layouts_are_conflict_free():
    """wave_fft.hpp's three LDS layouts against the lane groups of ds_write_b128 (8 contiguous lanes, 8 slots) and ds_read_b128 (four
    non-contiguous 16-lane groups, 16 slots) -- MI355X_MICROARCH.md, LDS table: the transpositions conflict-free, the mirrored reads of
    the natural-order layout at most 2-way (the simulation the header cites)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "experiments", "lds_layout_check.py")
    spec = importlib.util.spec_from_file_location("lds_layout_check", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    r = m.report()
    assert sorted(m.slot(k) for k in range(1024)) == list(range(1024))
    assert all(r[k] == 1 for k in ("transposition 1 write", "transposition 1 read", "transposition 2 write", "transposition 2 read", "natural-order write")), r
    assert r["natural-order read k"] <= 2 and r["natural-order read N-k"] <= 2, r


def test_c_shard_plan_equals_the_python_split():
    import numpy as np
    """emagls_jobs_shard (the split behind emagls_jobs_run_devices and the MEX 'jobs' command) against emagls_amd/batch.py on BASELINE
    config 4 -- 256 array radii over 8 ranks, padded lane batches of equal cost, whole batches by longest processing time -- and on a
    list of equal-shape jobs (round robin).  No GPU: the planner reads descriptors only."""
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    from emagls_amd.batch import padded_lane_batches, shard_lane_batches, simulation_order, shard_jobs
    h = np.zeros((8, 40))
    azi = np.zeros(40)
    radii = np.linspace(0.02, 0.10, 256)
    jl = JobList()
    for r in radii:
        jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, h, h, azi, azi, mic_radius=float(r), mic_azi=np.zeros(32), mic_zen=np.zeros(32), out_shape=(1024, 32, False))
    rank, pos, pad = jl.shard(8, 16)
    so = [simulation_order(4, 48000.0, float(r), raw=True) for r in radii]
    per_rank, load = shard_lane_batches(padded_lane_batches(so, 16), 8)
    for r, bl in enumerate(per_rank):
        share = [j for idx, _ in bl for j in idx]
        assert [j for j in range(256) if rank[j] == r and True] == sorted(share)
        assert [pos[j] for j in share] == list(range(len(share)))
        for idx, p in bl:
            assert all(pad[j] == p for j in idx)
    assert max(load) / min(load) < 1.02
    # equal shapes: one job per unit, longest processing time on unit costs = round robin
    jl2 = JobList()
    for _ in range(37):
        jl2.add(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, h, h, azi, azi, mic_radius=0.042, mic_azi=np.zeros(32), mic_zen=np.zeros(32), out_shape=(512, 25, True))
    rank2, pos2, pad2 = jl2.shard(4)
    shards = shard_jobs(np.ones(37), 4)
    for r, sjobs in enumerate(shards):
        assert [j for j in range(37) if rank2[j] == r] == sjobs
    assert all(p == 0 for p in pad2)


def test_xcd_run_index_is_a_bijection():
    """The XCD-aware workgroup -> (lane, tile) map of emagls_amd/csrc/common.hpp (xcd_run_index), restated: workgroup L of the dispatch
    order runs on XCD L % 8 and takes item x q + min(x, r) + L // 8 of the lane-major (lane, tile) list (W = 8 q + r items).  For any
    tile and lane count it must hit every item exactly once, and the items of one XCD must be a contiguous run."""
    for nt, lanes in ((28, 1), (28, 16), (28, 20), (28, 32), (528, 8), (21, 5), (96, 3), (1, 1), (7, 9)):
        W = nt * lanes
        q, r = divmod(W, 8)
        seen = {}
        for L in range(W):
            x, slot = L % 8, L // 8
            g = x * q + min(x, r) + slot
            assert 0 <= g < W
            seen.setdefault(x, []).append(g)
        items = sorted(g for v in seen.values() for g in v)
        assert items == list(range(W))
        for x, v in seen.items():
            assert v == list(range(v[0], v[0] + len(v)))   # (in slot order: a contiguous, ascending run)


def test_c_shard_plan_on_a_mixed_list():
    """emagls_jobs_shard on a list that mixes what a study mixes: an array-radius family (lane batches, laid out for their batch's order), LS and
    MagLS designs, a 64-capsule design and FromAtf subjects (units of their own).  Every job lands on exactly one rank at a position of its
    own, only the radius family is padded, and no rank stays empty."""
    import numpy as np
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    h = np.zeros((8, 40))
    azi = np.zeros(40)
    jl = JobList()
    fam = []
    for r in np.linspace(0.03, 0.06, 40):
        fam.append(jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 512, h, h, azi, azi, mic_radius=float(r), mic_azi=np.zeros(32), mic_zen=np.zeros(32), out_shape=(512, 32, False)))
    other = []
    for _ in range(6):
        other.append(jl.add(L.KIND_LS, "real", 4, 48000.0, 8, h, h, azi, azi, out_shape=(8, 25, False)))
        other.append(jl.add(L.KIND_MAGLS, "complex", 3, 48000.0, 64, h, h, azi, azi, out_shape=(64, 16, True)))
    other.append(jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 512, h, h, azi, azi, mic_radius=0.042, mic_azi=np.zeros(64), mic_zen=np.zeros(64), out_shape=(512, 64, False)))
    for _ in range(3):
        other.append(jl.add(L.KIND_FROM_ATF, "real", 0, 48000.0, 256, h, h, azi, azi, atf=1, atf_azi=azi, atf_zen=azi, nmics=8, f_trans=2000.0, out_shape=(256, 8, False)))
    world = 4
    rank, pos, pad = jl.shard(world, 16)
    n = len(jl)
    assert sorted(set(rank)) == list(range(world))
    for r in range(world):
        mine = [j for j in range(n) if rank[j] == r]
        assert sorted(pos[j] for j in mine) == list(range(len(mine)))
    assert all(pad[j] == 0 for j in other)
    assert any(pad[j] > 0 for j in fam)
