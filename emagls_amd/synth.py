"""Deterministic synthetic inputs (the reference's HRIR / ATF data sets are not shipped).

Used by bench.py, __graft_entry__.smoke() and the tests.  NumPy/SciPy only; this is input
generation, not part of the compute path, and it does not depend on oracle/.

* HRIRs: rigid-sphere head (radius a, ears at azimuth +-90 deg on the equator), plane-wave incidence,
  128 taps at 48 kHz, plus seeded Gaussian noise (measurement-noise stand-in).
* grids: the real 2702-point HRIR grid and the 32-point em32 grid travel as test fixtures; for runs
  without them `fibonacci_grid` gives a quasi-uniform grid of any size.
* ATFs: rigid-sphere scattering at M microphone positions on a glasses-like arc.
"""
from __future__ import annotations

import numpy as np
import scipy.special as sps

C_SOUND = 343.0

# em32 geometry, hard-coded in the reference's harness (verifyEMagLs.m:29-31) -- input data.
EM32_AZI_DEG = [0, 32, 0, 328, 0, 45, 69, 45, 0, 315, 291, 315, 91, 90, 90, 89, 180, 212, 180, 148, 180, 225,
                249, 225, 180, 135, 111, 135, 269, 270, 270, 271]
EM32_ZEN_DEG = [69, 90, 111, 90, 32, 55, 90, 125, 148, 125, 90, 55, 21, 58, 121, 159, 69, 90, 111, 90, 32, 55,
                90, 125, 148, 125, 90, 55, 21, 58, 122, 159]
EM32_RADIUS = 0.042


def em32_grid():
    return np.deg2rad(np.asarray(EM32_AZI_DEG, dtype=np.float64)), np.deg2rad(np.asarray(EM32_ZEN_DEG, dtype=np.float64))


def fibonacci_grid(n):
    """Quasi-uniform spherical Fibonacci lattice: (azimuth, zenith) in radians."""
    i = np.arange(n) + 0.5
    zen = np.arccos(1 - 2 * i / n)
    azi = np.mod(np.pi * (1 + 5 ** 0.5) * i, 2 * np.pi)
    return azi, zen


def _unit(azi, zen):
    return np.column_stack([np.sin(zen) * np.cos(azi), np.sin(zen) * np.sin(azi), np.cos(zen)])


def _sphere_surface_tf(ka, cos_theta, nmax):
    """Pressure on a rigid sphere for a unit plane wave arriving FROM the direction at angle theta
    to the observation point, phase-referenced to the sphere centre, e^{+i w t} convention:
    p = sum_n (2n+1) i^n [ -i / ((ka)^2 h_n^(2)'(ka)) ] P_n(cos theta)   (Wronskian form)."""
    out = np.zeros((ka.size, cos_theta.size), dtype=np.complex128)
    n = np.arange(nmax + 1)
    Pn = sps.eval_legendre(n[:, None], cos_theta[None, :])  # (nmax+1) x D
    for i, x in enumerate(ka):
        if x == 0:
            out[i] = 1.0
            continue
        dh = sps.spherical_jn(n, x, derivative=True) - 1j * sps.spherical_yn(n, x, derivative=True)
        with np.errstate(all="ignore"):
            c = (2 * n + 1) * (1j ** n) * (-1j) / (x * x * dh)
        c[~np.isfinite(c)] = 0
        out[i] = c @ Pn
    return out


_CLEAN_HRIRS = {}


def rigid_sphere_hrirs(azi, zen, fs=48000.0, taps=128, head_radius=0.0875, noise=1e-4, seed=20250310,
                       centre_delay=40.0):
    """(hL, hR), each [taps x D] float64 (MATLAB layout: samples down, directions across)."""
    azi = np.asarray(azi, dtype=np.float64)
    zen = np.asarray(zen, dtype=np.float64)
    u = _unit(azi, zen)
    ears = {"L": np.array([0.0, 1.0, 0.0]), "R": np.array([0.0, -1.0, 0.0])}  # +y = left
    P = taps // 2 + 1
    f = np.linspace(0, fs / 2, P)
    ka = 2 * np.pi * f / C_SOUND * head_radius
    nmax = int(np.ceil(ka.max() * 1.3)) + 12
    w = 2 * np.pi * f
    # smooth roll-off towards Nyquist so the truncated IR does not ring
    lp = 0.5 * (1 + np.cos(np.pi * np.clip((f - 0.7 * fs / 2) / (0.3 * fs / 2), 0, 1)))
    rng = np.random.default_rng(seed)
    # the noise-free responses depend on the geometry only: kept for the next call (a bench run builds ~70 HRIR sets that differ
    # in their noise seed)
    key = (azi.tobytes(), zen.tobytes(), float(fs), int(taps), float(head_radius), float(centre_delay))
    out = _CLEAN_HRIRS.get(key)
    if out is None:
        out = []
        for name in ("L", "R"):
            H = _sphere_surface_tf(ka, u @ ears[name], nmax)
            H = H * (lp * np.exp(-1j * w * centre_delay / fs))[:, None]
            H[-1] = H[-1].real
            h = np.fft.irfft(H, n=taps, axis=0)
            out.append(h)
        if len(_CLEAN_HRIRS) >= 8:
            _CLEAN_HRIRS.clear()
        _CLEAN_HRIRS[key] = out
    peak = max(np.abs(out[0]).max(), np.abs(out[1]).max())
    res = []
    for h in out:
        h = h / peak
        if noise:
            h = h + noise * rng.standard_normal(h.shape)
        res.append(np.ascontiguousarray(h))
    return res[0], res[1]


def glasses_atfs(natf=16384, nmics=8, fs=48000.0, taps=256, radius=0.09, noise=1e-4, seed=7):
    """ATF IRs [taps x nmics x natf] for mics on a glasses-like frontal arc of a rigid sphere,
    plus the ATF grid (azi, zen)."""
    azi, zen = fibonacci_grid(natf)
    u = _unit(azi, zen)
    mic_azi = np.deg2rad(np.linspace(-100, 100, nmics))
    mic_zen = np.deg2rad(90 - 10 * np.cos(np.linspace(-1, 1, nmics) * np.pi / 2))
    mu = _unit(mic_azi, mic_zen)
    P = taps // 2 + 1
    f = np.linspace(0, fs / 2, P)
    ka = 2 * np.pi * f / C_SOUND * radius
    nmax = int(np.ceil(ka.max() * 1.3)) + 12
    lp = 0.5 * (1 + np.cos(np.pi * np.clip((f - 0.7 * fs / 2) / (0.3 * fs / 2), 0, 1)))
    rng = np.random.default_rng(seed)
    atf = np.empty((taps, nmics, natf))
    for m in range(nmics):
        H = _sphere_surface_tf(ka, u @ mu[m], nmax)
        H = H * (lp * np.exp(-1j * 2 * np.pi * f * 40.0 / fs))[:, None]
        H[-1] = H[-1].real
        atf[:, m, :] = np.fft.irfft(H, n=taps, axis=0)
    atf /= np.abs(atf).max()
    if noise:
        atf += noise * rng.standard_normal(atf.shape)
    return atf, azi, zen
