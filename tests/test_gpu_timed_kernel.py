"""The kernel `bench.py` times -- `sweep_reg_kernel` behind `emagls_jobs_run` -- at the size it is timed at: BASELINE config 3
(lib/getEMagLsFilters.m:32-142; em32, N = 4, complex SH, 2702 directions, 128-tap HRIRs, 512-tap filters), lists of 20 designs (the
driver's `--steps 20`: one chunk, stages forked onto three streams, the sweep's 220 workgroups dealt round over all XCDs) and of 32
designs (the chunks of the long runs: four designs per XCD, twelve waves per workgroup), every design on its OWN HRIR set, both
placements of each launch, against the oracle on the same inputs.  The 901-direction tests of tests/test_gpu_parity.py launch 4
workgroups per design; here it is 11 (20 designs) and 8 (32 designs)."""
import numpy as np
import pytest

from oracle import emagls_oracle as O  # checker only

pytestmark = pytest.mark.gpu

TOL = 2e-7      # (north_star's bound is 1e-6; the synthesising sweeps sit at 5e-9 ... 6e-8: DESIGN.md section 3)
CHECKED = (0, 11, 19, 31)   # designs compared with the oracle (the last one only exists in the 32-design list)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def sets(grids):
    """32 HRIR sets on the reference's 2702-point grid: the sets of bench.py's timed region (same generator, same seeds)."""
    from emagls_amd import synth
    return [synth.rigid_sphere_hrirs(grids["azi"], grids["zen"], seed=20250310 + j) for j in range(32)]


@pytest.fixture(scope="module")
def oracle_filters(grids, sets):
    out = {}
    for j in CHECKED:
        out[j] = O.getEMagLsFilters(sets[j][0], sets[j][1], grids["azi"], grids["zen"], 0.042, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 512, "complex")
    return out


def run_list(grids, sets, n, runs=1):
    """`n` config-3 designs through emagls_jobs_run as bench.py calls it (chunks of 32, four in flight); `runs` calls of the same list
    (eager run, hipGraph capture, replay): returns the filters of every call."""
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    jl = JobList()
    for j in range(n):
        jl.add(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, sets[j][0], sets[j][1], grids["azi"], grids["zen"], mic_radius=0.042, mic_azi=grids["mic_azi"],
               mic_zen=grids["mic_zen"], out_shape=(512, 25, True))
    res = []
    for _ in range(runs):
        jl.run(batch_size=32, in_flight=4)
        res.append([(a.copy(), b.copy()) for a, b in jl.results()])
    return res


@pytest.mark.parametrize("n", [20, 32])
def test_timed_kernel_at_full_size(grids, sets, oracle_filters, monkeypatch, n):
    import ctypes
    from emagls_amd import Plan, _lib as L
    lib = L.load()
    # the form such a chunk takes (decided per launch): 3 = the register-resident sweep
    p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, 128, 2702, 0.042, 32)
    p.set_hrir_grid(grids["azi"], grids["zen"])
    p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])    # (the antipodal pairs of the array decide the form: 17 units on the em32)
    form = ctypes.c_int(0)
    L.check(lib.emagls_plan_sweep_form_in_batch(p._h, n, ctypes.byref(form)))
    p.close()
    assert form.value == 3
    got = {}
    # placements: "2" the default (20 designs: spread over all XCDs, 32: four designs per XCD), "0" every design inside one XCD,
    # "1" spread whenever it fits (32 designs: 256 workgroups of twelve waves, one per CU)
    for spread in ("2", "0", "1"):
        monkeypatch.setenv("EMAGLS_REG_SPREAD", spread)
        L.check(lib.emagls_cache_clear())
        runs = run_list(grids, sets, n, runs=3 if spread == "2" else 1)
        for later in runs[1:]:   # the captured graphs replay what the eager run did
            assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(runs[0], later))
        got[spread] = runs[0]
    L.check(lib.emagls_cache_clear())
    # every placement against the oracle (a placement changes the waves per workgroup, hence the workgroups of a design and the
    # grouping of the partial sums: 20 designs run 11 workgroups of 8 waves spread over all XCDs, 9 of 10 waves inside one XCD)
    worst = 0.0
    for spread in got:
        for j in CHECKED:
            if j >= n:
                continue
            oL, oR = oracle_filters[j]
            e = max(rel(got[spread][j][0], oL), rel(got[spread][j][1], oR))
            nd, db, adb = O.assert_all_close_metrics(np.hstack(got[spread][j]), np.hstack([oL, oR]))
            if spread == "2":
                print(f"emagls_jobs_run, {n} config-3 designs at full size, design {j}: rel = {e:.3e}, normalised max abs diff = {nd:.3e}, max |dB| = {adb:.2e}")
            worst = max(worst, e)
    assert worst < TOL
    # between the placements: the same operand values, another order of the partial sums
    between = max(max(rel(a[0], b[0]), rel(a[1], b[1])) for other in ("0", "1") for a, b in zip(got["2"], got[other]))
    print(f"emagls_jobs_run, {n} designs: placements of the sweep launch against each other: worst rel = {between:.2e}")
    assert between < 1e-10
