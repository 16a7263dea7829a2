"""Plan objects: inputs resident in HBM, repeated execution, stage timing, debug buffers."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class Plan:
    def __init__(self, kind, basis="real", order=4, fs=48000.0, length=512, nsamp=128, ndirs=0, mic_radius=0.0, nmics=0,
                 f_trans=0.0, atf_taps=0, natf=0, custom_basis=False, diffuseness=False, sim_order_pad=0):
        self._lib = L.load()
        self.desc = L.DesignDesc(kind, L.BASIS[basis], order, fs, length, nsamp, ndirs, mic_radius, nmics, f_trans,
                                 atf_taps, natf, 1 if custom_basis else 0, 1 if diffuseness else 0, int(sim_order_pad))
        self._cplx = basis == "complex"
        self._h = C.c_void_p()
        L.check(self._lib.emagls_plan_create(C.byref(self.desc), C.byref(self._h)))
        self._keep = []

    def close(self):
        if self._h:
            self._lib.emagls_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, a):
        a = np.asfortranarray(np.asarray(a, dtype=np.float64))
        self._keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    def set_hrir_grid(self, azi, zen):
        L.check(self._lib.emagls_plan_set_hrir_grid(self._h, self._p(azi), self._p(zen)))

    def set_mic_grid(self, azi, zen=None):
        """zen may be omitted for an equatorial array (KIND_EMA_CH): every microphone sits at pi/2."""
        L.check(self._lib.emagls_plan_set_mic_grid(self._h, self._p(azi), self._p(zen) if zen is not None else None))

    def set_basis(self, Y_hrir, Y_mic=None):
        """custom_basis plans: the SH matrices a custom shFunction returned ([ndirs x S], [nmics x S]) instead of the grids."""
        dt = np.complex128 if self._cplx else np.float64

        def ptr(a):
            a = np.asfortranarray(np.asarray(a, dtype=dt))
            self._keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        L.check(self._lib.emagls_plan_set_basis(self._h, ptr(Y_hrir), ptr(Y_mic) if Y_mic is not None else None))

    def set_hrirs(self, hL, hR):
        L.check(self._lib.emagls_plan_set_hrirs(self._h, self._p(hL), self._p(hR)))

    def set_atfs(self, atf, azi, zen):
        L.check(self._lib.emagls_plan_set_atfs(self._h, self._p(atf), self._p(azi), self._p(zen)))

    def execute(self):
        L.check(self._lib.emagls_plan_execute(self._h))

    def synchronize(self):
        L.check(self._lib.emagls_plan_synchronize(self._h))

    def info(self):
        i = L.PlanInfo()
        L.check(self._lib.emagls_plan_get_info(self._h, C.byref(i)))
        return i

    def get_filters(self):
        i = self.info()
        dt = np.complex128 if i.out_is_complex else np.float64
        wL = np.zeros((i.out_rows, i.out_cols), dtype=dt, order="F")
        wR = np.zeros((i.out_rows, i.out_cols), dtype=dt, order="F")
        L.check(self._lib.emagls_plan_get_filters(self._h, wL.ctypes.data_as(C.c_void_p), wR.ctypes.data_as(C.c_void_p)))
        return wL, wR

    def set_profiling(self, level):
        L.check(self._lib.emagls_plan_set_profiling(self._h, int(level)))

    def set_streams(self, n):
        L.check(self._lib.emagls_plan_set_streams(self._h, int(n)))

    def stage_times(self):
        n = self._lib.emagls_plan_num_stages(self._h)
        ms = (C.c_double * max(n, 1))()
        L.check(self._lib.emagls_plan_stage_times(self._h, ms, n))
        return [(self._lib.emagls_plan_stage_name(self._h, i).decode(), ms[i]) for i in range(1, n)]

    def sweep_kernel_time(self):
        tot = C.c_double(0.0)
        n = C.c_int(0)
        L.check(self._lib.emagls_plan_sweep_kernel_time(self._h, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    def debug(self, name, dtype, shape=None):
        nb = C.c_size_t(0)
        L.check(self._lib.emagls_plan_debug_buffer(self._h, name.encode(), None, C.byref(nb)))
        buf = np.empty(nb.value // np.dtype(dtype).itemsize, dtype=dtype)
        nb2 = C.c_size_t(buf.nbytes)
        L.check(self._lib.emagls_plan_debug_buffer(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), C.byref(nb2)))
        return buf.reshape(shape) if shape is not None else buf

    @property
    def stream(self):
        return self._lib.emagls_plan_stream(self._h)


class Batch:
    """Several plans of identical shape executed together, the sequential sweep as ONE resident launch for all of them:
    eMagLS / eMagLS2 / EMA designs (lane mode: one launch of every kernel for all designs), or FromAtf plans -- the HRTF subjects
    of one ATF set, whose ATF side is computed once (shares_atf_side)."""

    def __init__(self, plans):
        self._lib = L.load()
        self.plans = list(plans)
        arr = (C.c_void_p * len(self.plans))(*[p._h for p in self.plans])
        self._h = C.c_void_p()
        L.check(self._lib.emagls_batch_create(arr, len(self.plans), C.byref(self._h)))

    def execute(self):
        L.check(self._lib.emagls_batch_execute(self._h))

    def synchronize(self):
        L.check(self._lib.emagls_batch_synchronize(self._h))

    def get_filters(self):
        outs = []
        i = self.plans[0].info()
        dt = np.complex128 if i.out_is_complex else np.float64
        for p in self.plans:
            outs.append((np.zeros((i.out_rows, i.out_cols), dtype=dt, order="F"), np.zeros((i.out_rows, i.out_cols), dtype=dt, order="F")))
        n = len(self.plans)
        pl = (C.c_void_p * n)(*[o[0].ctypes.data for o in outs])
        pr = (C.c_void_p * n)(*[o[1].ctypes.data for o in outs])
        L.check(self._lib.emagls_batch_get_filters(self._h, pl, pr))
        return outs

    def set_profiling(self, level):
        L.check(self._lib.emagls_batch_set_profiling(self._h, int(level)))

    def lane_mode(self):
        """True when one launch of every kernel covers all designs of the batch (identical shapes), False in stream mode."""
        v = C.c_int(0)
        L.check(self._lib.emagls_batch_lane_mode(self._h, C.byref(v)))
        return bool(v.value)

    def set_stream(self, hip_stream):
        """Run on the caller's hipStream_t (an integer handle, e.g. torch.cuda.Stream().cuda_stream); the caller keeps it alive."""
        L.check(self._lib.emagls_batch_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def shares_atf_side(self):
        """FromAtf subjects: True when the last execute computed the ATF side once for all plans (same grids and ATF set)."""
        v = C.c_int(0)
        L.check(self._lib.emagls_batch_shares_atf_side(self._h, C.byref(v)))
        return bool(v.value)

    def share_geometry(self, enable=True):
        """HRIR sets on one geometry (same grids, array, orders): run the geometry stages once for the batch (opt-in; the grids
        are compared on the device, plans that differ run as independent designs)."""
        L.check(self._lib.emagls_batch_set_geometry_sharing(self._h, int(bool(enable))))

    def shares_geometry(self):
        """True when the last execute ran the geometry stages once for all plans."""
        v = C.c_int(0)
        L.check(self._lib.emagls_batch_shares_geometry(self._h, C.byref(v)))
        return bool(v.value)

    def set_streams(self, n):
        """Lane mode: fork the stages before the sweep onto n (1..4) streams (see emagls_batch_set_streams)."""
        L.check(self._lib.emagls_batch_set_streams(self._h, int(n)))

    def set_stage_order(self, order):
        """Order of the stages before the sweep of a batch of up to 8 designs (0 / 1 / 2, see emagls_batch_set_stage_order)."""
        L.check(self._lib.emagls_batch_set_stage_order(self._h, int(order)))

    def set_side_stream(self, hip_stream):
        """The stream of the second lane group of a batch of more than 8 designs (see emagls_batch_set_side_stream)."""
        L.check(self._lib.emagls_batch_set_side_stream(self._h, C.c_void_p(int(hip_stream))))

    def sweep_time_ms(self):
        ms = C.c_double(0.0)
        L.check(self._lib.emagls_batch_sweep_time(self._h, C.byref(ms)))
        return ms.value

    def get_filters_into(self, ptrs_l, ptrs_r):
        """Device (or host) destination addresses, one pair per plan: no host staging."""
        n = len(self.plans)
        pl = (C.c_void_p * n)(*ptrs_l)
        pr = (C.c_void_p * n)(*ptrs_r)
        L.check(self._lib.emagls_batch_get_filters(self._h, pl, pr))

    def close(self):
        if self._h:
            self._lib.emagls_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
