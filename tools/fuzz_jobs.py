"""Random job lists on the GPU box: `n` seeded lists of 9-28 designs of one random shape through the library's scheduler
(emagls_amd.jobs.JobList -> emagls_jobs_run), i.e. through the register-resident sweep (launches of more than 8 designs) in the
layouts the launch picks -- designs inside one XCD or spread over all of them, 4-12 waves per workgroup --, against the one-shot
calls of the same designs (which take the slab form of the sweep) and, for one design per list, against the oracle.

    python tools/fuzz_jobs.py [n] [seed]

Shapes: eMagLS / eMagLS2 / EMAinCH, 100-3000 directions, 2-32 microphones (with and without antipodal pairs: up to 18 polynomial
units run the register-resident form, more the slab form), orders 0-4, filter lengths 32-512, sampling rates 16-96 kHz."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def mic_grid(rng, M, paired):
    """M microphones; `paired`: as many exact antipodal pairs as fit (the em32's kind of layout), the rest single capsules."""
    from emagls_amd import synth
    if not paired:
        azi, zen = synth.fibonacci_grid(M)
        return np.mod(azi + 0.05 * rng.standard_normal(M), 2 * np.pi), np.clip(zen + 0.03 * rng.standard_normal(M), 0.05, np.pi - 0.05)
    npair = M // 2
    # (the upper half of a Fibonacci grid of 2 npair points and its mirror image: spread over the sphere like the em32's pairs; a first
    # version squeezed the capsules towards the poles and drew designs whose pinv(Y_lo) is ill-conditioned -- two lists of seed 11
    # came out above the tolerance against the oracle AND against the single calls, profiles/r05_fuzz_jobs.md)
    fa, fz = synth.fibonacci_grid(2 * npair)
    up = np.argsort(fz)[:npair]
    azi, zen = fa[up] + 0.05 * rng.standard_normal(npair), np.clip(fz[up] + 0.02 * rng.standard_normal(npair), 0.05, np.pi / 2 - 0.02)
    a = np.concatenate([azi, azi + np.pi]); z = np.concatenate([zen, np.pi - zen])
    if M % 2:
        a = np.append(a, 0.3); z = np.append(z, np.pi / 2)
    return np.mod(a, 2 * np.pi), z


def draw(rng):
    kind = str(rng.choice(["emagls", "emagls2", "emainch"], p=[0.45, 0.35, 0.2]))
    N = int(rng.integers(0, 5))
    if kind == "emagls":
        M = int(rng.integers(max(2, (N + 1) ** 2), 33))
    elif kind == "emainch":
        M = int(rng.integers(max(2, 2 * N + 1), 19))
    else:
        M = int(rng.integers(4, 33))
    taps = int(rng.choice([16, 33, 64, 100, 128]))
    ln = int(2 * rng.integers(max(16, taps // 2), 257))
    paired = bool(rng.random() < 0.6)
    if kind == "emagls" and paired:
        # an array of antipodal pairs sees the even orders through the pairs' sums and the odd ones through their differences: pinv(Y_lo)
        # exists only with at least as many pairs as the larger of the two families has functions (order 4: 15 -- the em32 has exactly that)
        even, odd = sum(2 * n + 1 for n in range(0, N + 1, 2)), sum(2 * n + 1 for n in range(1, N + 1, 2))
        if M // 2 < max(even, odd):
            paired = False
    return dict(kind=kind, n=int(rng.integers(9, 29)), D=int(rng.integers(100, 3000)), taps=taps, ln=ln,
                fs=float(rng.choice([16000.0, 32000.0, 44100.0, 48000.0, 96000.0])), r=float(rng.uniform(0.01, 0.08)), M=M, N=N,
                basis=str(rng.choice(["real", "complex"])), paired=paired, spread=str(rng.choice(["0", "1", "2"])), seed=int(rng.integers(1 << 30)))


def run(c):
    import emagls_amd as E
    from emagls_amd import synth, _lib as L
    from emagls_amd.jobs import JobList
    from oracle import emagls_oracle as O
    import shape_cases as SC
    os.environ["EMAGLS_REG_SPREAD"] = c["spread"]
    rng = np.random.default_rng(c["seed"])
    azi, zen = synth.fibonacci_grid(c["D"])
    if c["kind"] == "emainch":
        ma, mz = np.linspace(0, 2 * np.pi, c["M"], endpoint=False) + 0.2, None
    else:
        ma, mz = mic_grid(rng, c["M"], c["paired"])
    sets = [synth.rigid_sphere_hrirs(azi, zen, fs=c["fs"], taps=c["taps"], centre_delay=c["taps"] / 4, seed=11 + j, head_radius=0.08 + 0.002 * j)
            for j in range(c["n"])]
    kid = {"emagls": L.KIND_EMAGLS, "emagls2": L.KIND_EMAGLS2, "emainch": L.KIND_EMA_CH}[c["kind"]]
    from emagls_amd.batch import _out_shape
    shape = _out_shape(dict(kind=kid, basis=c["basis"], order=c["N"], fs=c["fs"], length=c["ln"], hL=sets[0][0], mic_radius=c["r"], mic_azi=ma))
    jl = JobList()
    for hL, hR in sets:
        jl.add(kid, c["basis"], c["N"], c["fs"], c["ln"], hL, hR, azi, zen, mic_radius=c["r"], mic_azi=ma, mic_zen=mz, out_shape=shape)
    jl.run(batch_size=32, in_flight=2)
    res = jl.results()

    def single(j):
        hL, hR = sets[j]
        if c["kind"] == "emagls":
            return E.getEMagLsFilters(hL, hR, azi, zen, c["r"], ma, mz, c["N"], c["fs"], c["ln"], c["basis"])
        if c["kind"] == "emagls2":
            return E.getEMagLs2Filters(hL, hR, azi, zen, c["r"], ma, mz, c["N"], c["fs"], c["ln"], c["basis"])
        return E.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, c["r"], ma, c["N"], c["fs"], c["ln"], c["basis"])
    worst_single = 0.0
    for j in sorted({0, c["n"] // 2, c["n"] - 1}):
        w = single(j)
        assert w[0].shape == res[j][0].shape and w[0].dtype == res[j][0].dtype, (w[0].shape, res[j][0].shape, w[0].dtype, res[j][0].dtype)
        worst_single = max(worst_single, SC.rel(res[j][0], w[0]), SC.rel(res[j][1], w[1]))
    j = c["n"] - 1
    hL, hR = sets[j]
    if c["kind"] == "emagls":
        o = O.getEMagLsFilters(hL, hR, azi, zen, c["r"], ma, mz, c["N"], c["fs"], c["ln"], c["basis"])
    elif c["kind"] == "emagls2":
        o = O.getEMagLs2Filters(hL, hR, azi, zen, c["r"], ma, mz, c["N"], c["fs"], c["ln"], c["basis"])
    else:
        o = O.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, c["r"], ma, c["N"], c["fs"], c["ln"], c["basis"])
    return worst_single, max(SC.rel(res[j][0], o[0]), SC.rel(res[j][1], o[1]))


def main():
    from emagls_amd._lib import EmaglsError
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    tally = dict(ok=0, refused=0, above_tolerance=0, error=0)
    ws = wo = 0.0
    for i in range(n):
        c = draw(rng)
        t = time.time()
        try:
            a, b = run(c)
            good = a < 1e-6 and b < 1e-6
            tally["ok" if good else "above_tolerance"] += 1
            if good:
                ws, wo = max(ws, a), max(wo, b)
            print(f"case {i} {c} -> list vs single calls {a:.2e}, vs oracle {b:.2e}{'' if good else '  ABOVE 1e-6'} ({time.time() - t:.1f} s)", flush=True)
        except EmaglsError as e:
            if e.code in (1, 2):
                tally["refused"] += 1
                print(f"case {i} {c} -> refused: {str(e)[:160]}", flush=True)
            else:
                tally["error"] += 1
                print(f"case {i} {c} -> ERROR {e}", flush=True)
        except Exception as e:   # noqa: BLE001
            tally["error"] += 1
            print(f"case {i} {c} -> ERROR {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"summary: {tally} worst accepted: job list vs single calls {ws:.2e}, vs oracle {wo:.2e}")


if __name__ == "__main__":
    main()
