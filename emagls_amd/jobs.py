"""Job lists on the library's own scheduler (emagls_jobs_run, include/emagls.h): the loop over array radii / HRIR sets / subjects
that a user of the reference writes around one of its functions (testEMagLs.m:75-95), handed to the library in one call.  The
library cuts the list into chunks of one shape, runs every chunk as a lane batch and keeps several chunks in flight from its own
threads; this module only packs the descriptors and the pointers."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class JobList:
    """Collects jobs and runs them.  Inputs are host arrays (kept alive until `run` returns) or device addresses (ints);
    outputs are written where `out` points: host arrays allocated here (`results()`), or the addresses given per job."""

    def __init__(self):
        self._jobs, self._keep, self._outs = [], [], []

    def _ptr(self, a):
        if a is None:
            return None
        if isinstance(a, int):        # a device (or any raw) address
            return C.c_void_p(a)
        a = np.asfortranarray(np.asarray(a, dtype=np.float64))
        self._keep.append(a)
        return C.c_void_p(a.ctypes.data)

    def add(self, kind, basis, order, fs, length, hL, hR, hrir_azi, hrir_zen=None, mic_radius=0.0, mic_azi=None, mic_zen=None,
            atf=None, atf_azi=None, atf_zen=None, f_trans=0.0, diffuseness=False, sim_order_pad=0, nsamp=None, ndirs=None, nmics=None,
            out=None, out_shape=None):
        """One design.  `out` = (address of wL, address of wR) to write the filters to memory of the caller's (device or host);
        otherwise host arrays of `out_shape` = (rows, cols, is_complex) are allocated (results())."""
        if nsamp is None:
            nsamp, ndirs = np.asarray(hL).shape
        if nmics is None:
            nmics = 0 if mic_azi is None else int(np.asarray(mic_azi).size)
        atf_taps = natf = 0
        if atf is not None and not isinstance(atf, int):
            atf_taps, nmics, natf = np.asarray(atf).shape
        desc = L.DesignDesc(int(kind), L.BASIS[basis], int(order), float(fs), int(length), int(nsamp), int(ndirs), float(mic_radius), int(nmics),
                            float(f_trans), int(atf_taps), int(natf), 0, 1 if diffuseness else 0, int(sim_order_pad))
        j = L.Job()
        j.desc = desc
        j.hL, j.hR = self._ptr(hL), self._ptr(hR)
        j.hrir_azi, j.hrir_zen = self._ptr(hrir_azi), self._ptr(hrir_zen)
        j.mic_azi, j.mic_zen = self._ptr(mic_azi), self._ptr(mic_zen)
        j.atf, j.atf_azi, j.atf_zen = self._ptr(atf), self._ptr(atf_azi), self._ptr(atf_zen)
        if out is not None:
            j.wL, j.wR = C.c_void_p(int(out[0])), C.c_void_p(int(out[1]))
            self._outs.append(None)
        else:
            rows, cols, cplx = out_shape
            dt = np.complex128 if cplx else np.float64
            wl, wr = np.zeros((rows, cols), dtype=dt, order="F"), np.zeros((rows, cols), dtype=dt, order="F")
            j.wL, j.wR = C.c_void_p(wl.ctypes.data), C.c_void_p(wr.ctypes.data)
            self._outs.append((wl, wr))
        self._jobs.append(j)
        return len(self._jobs) - 1

    def __len__(self):
        return len(self._jobs)

    def run(self, batch_size=0, in_flight=0, share_geometry=False, devices=None):
        """Runs every job added so far; returns when all filters are in place.  `devices`: HIP ordinals of the GPUs of this process to
        split the list over (emagls_jobs_run_devices: one host thread per device, host arrays in and out, no gather); None: the
        current device."""
        n = len(self._jobs)
        if n == 0:
            return
        arr = (L.Job * n)(*self._jobs)
        flags = L.JOBS_SHARE_GEOMETRY if share_geometry else 0
        if devices is None:
            L.check(L.load().emagls_jobs_run(arr, n, int(batch_size), int(in_flight), flags))
        else:
            dev = (C.c_int * len(devices))(*[int(d) for d in devices])
            L.check(L.load().emagls_jobs_run_devices(arr, n, dev, len(devices), int(batch_size), int(in_flight), flags))

    def shard(self, world, max_batch=16):
        """(rank of every job, its position in the rank's share, the simulation order its lane batch is laid out for) -- the split of
        emagls_jobs_shard (needs no GPU)."""
        n = len(self._jobs)
        arr = (L.Job * max(n, 1))(*self._jobs)
        rank, pos, pad = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))()
        L.check(L.load().emagls_jobs_shard(arr, n, int(world), int(max_batch), rank, pos, pad))
        return list(rank[:n]), list(pos[:n]), list(pad[:n])

    def results(self):
        """[(wL, wR), ...] of the jobs whose filters were allocated here (None for jobs with their own `out`)."""
        return list(self._outs)
