"""Shape sweep: every entry point over ragged / small / large shapes, GPU result against the oracle.
Used by tests/test_gpu_shapes.py (a subset, FAST) and tools/fuzz_shapes.py (all cases)."""
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from oracle import emagls_oracle as O


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def mics(n, seed):
    azi, zen = synth.fibonacci_grid(n)
    rng = np.random.default_rng(seed)
    return np.mod(azi + 0.05 * rng.standard_normal(n), 2 * np.pi), np.clip(zen + 0.03 * rng.standard_normal(n), 0.05, np.pi - 0.05)


CASES = []
# kind, D, taps, len, fs, radius, nmics, order, basis
for i, (D, taps, ln, fs, r, M, N, basis) in enumerate([
        (450, 64, 64, 48000.0, 0.02, 4, 1, "real"),
        (450, 64, 128, 44100.0, 0.02, 9, 2, "complex"),
        (451, 100, 128, 48000.0, 0.03, 12, 2, "real"),
        (901, 128, 256, 48000.0, 0.042, 32, 4, "complex"),
        (901, 96, 128, 32000.0, 0.042, 25, 4, "real"),
        (1000, 128, 128, 48000.0, 0.042, 16, 3, "complex"),
        (1351, 128, 128, 48000.0, 0.05, 20, 3, "real"),
        (3001, 128, 128, 48000.0, 0.042, 32, 4, "real"),
        (3072, 64, 64, 48000.0, 0.042, 32, 4, "complex"),
        (3100, 64, 64, 48000.0, 0.042, 32, 4, "complex"),
        (700, 64, 64, 48000.0, 0.055, 32, 4, "real"),
        (400, 64, 64, 48000.0, 0.042, 32, 4, "real"),      # D == S
        (128, 64, 64, 48000.0, 0.005, 6, 1, "complex"),   # tiny array, simOrder = order
        (200, 64, 64, 16000.0, 0.042, 9, 2, "real"),
        (640, 64, 512, 48000.0, 0.042, 32, 4, "real"),     # len >> taps
        (640, 64, 64, 48000.0, 0.042, 5, 0, "real"),       # order 0
]):
    CASES.append(("emagls", D, taps, ln, fs, r, M, N, basis))
    CASES.append(("emagls2", D, taps, ln, fs, r, max(2, M - (i % 5)), N, basis))
for (D, taps, ln, fs, r, M, N, basis) in [
        (450, 64, 64, 48000.0, 0.02, 3, 1, "real"), (450, 64, 128, 48000.0, 0.03, 8, 2, "complex"),
        (901, 64, 128, 48000.0, 0.042, 13, 6, "real"), (901, 64, 64, 48000.0, 0.042, 31, 15, "complex"),
        (901, 64, 64, 48000.0, 0.042, 12, 0, "real"), (2702, 128, 128, 48000.0, 0.042, 9, 4, "complex")]:
    CASES.append(("emainch", D, taps, ln, fs, r, M, N, basis))
for (D, taps, ln, fs, N, basis) in [(30, 16, 32, 48000.0, 0, "real"), (64, 64, 64, 48000.0, 1, "complex"), (100, 33, 64, 44100.0, 2, "real"),
                                    (2702, 128, 512, 48000.0, 4, "complex"), (5000, 64, 128, 48000.0, 4, "real"), (25, 64, 128, 48000.0, 4, "real"),
                                    (1000, 128, 128, 96000.0, 3, "complex")]:
    CASES.append(("magls", D, taps, ln, fs, 0, 0, N, basis))
    CASES.append(("ls", D, taps, ln, fs, 0, 0, N, basis))
for (D, taps, ln, natf, M, ataps, ft) in [(450, 64, 64, 450, 1, 32, 1000.0), (450, 64, 128, 451, 3, 100, 2000.0), (901, 64, 128, 300, 8, 64, 1500.0),
                                          (901, 128, 256, 4096, 2, 300, 3000.0), (2702, 128, 128, 2702, 7, 64, 2000.0), (64, 32, 64, 5000, 8, 16, 500.0),
                                          (4096, 64, 64, 5000, 4, 64, 2000.0), (500, 64, 64, 37, 6, 64, 2000.0)]:
    CASES.append(("atf", D, taps, ln, 48000.0, natf, M, ataps, ft))

# fewer HRIR directions than simulated SH channels (D < (N_sim+1)^2, the reference's SVD takes them as they come): fine as long
# as the orders of the orthonormal route are covered (capi.hip, plan_routes); found by tools/fuzz_random.py
D_BELOW_S = len(CASES)
CASES.append(("emagls2", 590, 128, 342, 48000.0, 0.05603182265749108, 6, 4, "real"))     # simulation order 25: S = 676
CASES.append(("emagls", 486, 64, 248, 96000.0, 0.027496569985500114, 30, 4, "real"))     # simulation order 25
CASES.append(("emainch", 438, 128, 336, 32000.0, 0.06818923842651228, 13, 2, "complex"))  # simulation order 21: S = 484
CASES.append(("emagls", 300, 64, 128, 48000.0, 0.042, 32, 4, "real"))                    # config 3's array on 300 directions: S = 400

def run(case):
    kind = case[0]
    if kind == "atf":
        _, D, taps, ln, fs, natf, M, ataps, ft = case
        azi, zen = synth.fibonacci_grid(D)
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
        atf, aazi, azen = synth.glasses_atfs(natf=natf, nmics=M, taps=ataps, fs=fs)
        hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi + 0.01, azen])
        w = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, fs, ln, ft, verbose=False)
        o = O.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, fs, ln, ft)
        return max(rel(w[0], o[0]), rel(w[1], o[1]))
    _, D, taps, ln, fs, r, M, N, basis = case
    azi, zen = synth.fibonacci_grid(D)
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
    if kind == "ls":
        w, o = E.getLsFilters(hL, hR, azi, zen, N, basis), O.getLsFilters(hL, hR, azi, zen, N, basis)
    elif kind == "magls":
        w, o = E.getMagLsFilters(hL, hR, azi, zen, N, fs, ln, basis), O.getMagLsFilters(hL, hR, azi, zen, N, fs, ln, basis)
    elif kind == "emainch":
        ma = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2
        w = E.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
        o = O.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
    else:
        ma, mz = mics(M, D + M)
        fn, fo = (E.getEMagLsFilters, O.getEMagLsFilters) if kind == "emagls" else (E.getEMagLs2Filters, O.getEMagLs2Filters)
        w = fn(hL, hR, azi, zen, r, ma, mz, N, fs, ln, basis)
        o = fo(hL, hR, azi, zen, r, ma, mz, N, fs, ln, basis)
    assert w[0].shape == o[0].shape and w[0].dtype == o[0].dtype, (w[0].shape, o[0].shape, w[0].dtype, o[0].dtype)
    return max(rel(w[0], o[0]), rel(w[1], o[1]))



# cases of the sweep that run in the GPU test suite (each finishes in a few seconds including the oracle)
FAST = [0, 3, 4, 5, 16, 18, 22, 23, 24, 26, 30, 31, 32, 33, 34, 36, 38, 40, 42, 47, 48, 49, 52, 53, 57, 58, 59] + list(range(D_BELOW_S, D_BELOW_S + 4))
# case 35 (order-15 circular harmonics, 31 microphones) is kept as a documented ill-posed comparison: the lowest bins have
# cond(pwGrid) > 1/eps, where the clipped singular subspace -- hence the reference's own result -- is rounding noise
ILL_POSED = [35]
