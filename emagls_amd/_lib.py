"""ctypes binding of include/emagls.h.  Loads emagls_amd/lib/libemagls.so; there is no fallback:
if the HIP library is missing or no GPU is present, calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMAGLS_LIB_PATH") or os.path.join(_HERE, "lib", "libemagls.so")   # (the override: A/B runs of two builds in one GPU session)

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NUMERIC = 0, 1, 2, 3, 4
BASIS = {"real": 0, "complex": 1}
KIND_LS, KIND_MAGLS, KIND_EMAGLS, KIND_EMAGLS2, KIND_FROM_ATF, KIND_EMA_CH, KIND_MAGLS_2D, KIND_EMA_SH = range(8)
RADIAL = {"tikhonov": 0, "softlimit": 1, "full": 2, "none": 3}

c_dp = C.POINTER(C.c_double)
c_i64 = C.c_int64


class DesignDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("basis", C.c_int), ("order", C.c_int), ("fs", C.c_double), ("len", c_i64),
                ("nsamp", c_i64), ("ndirs", c_i64), ("mic_radius", C.c_double), ("nmics", c_i64),
                ("f_trans", C.c_double), ("atf_taps", c_i64), ("natf", c_i64), ("custom_basis", C.c_int),
                ("diffuseness", C.c_int), ("sim_order_pad", C.c_int)]


class PlanInfo(C.Structure):
    _fields_ = [("nfft", C.c_int), ("num_pos_freqs", C.c_int), ("k_cut", C.c_int), ("sim_order", C.c_int),
                ("num_sh_sim", C.c_int), ("num_channels", C.c_int), ("out_is_complex", C.c_int),
                ("out_rows", c_i64), ("out_cols", c_i64), ("grp_delay_l", C.c_double), ("grp_delay_r", C.c_double),
                ("mean_grid_dev_deg", C.c_double), ("num_sweep_launches", C.c_int), ("device_bytes", c_i64),
                ("gram_from", C.c_int), ("hh_end", C.c_int), ("hh_orders", C.c_int), ("g_first", C.c_int),
                ("sim_order_own", C.c_int), ("sweep_form", C.c_int), ("sweep_units", C.c_int)]


class Job(C.Structure):
    """emagls_job: one design of a job list (emagls_jobs_run)."""
    _fields_ = [("desc", DesignDesc), ("hL", C.c_void_p), ("hR", C.c_void_p), ("hrir_azi", C.c_void_p), ("hrir_zen", C.c_void_p),
                ("mic_azi", C.c_void_p), ("mic_zen", C.c_void_p), ("atf", C.c_void_p), ("atf_azi", C.c_void_p), ("atf_zen", C.c_void_p),
                ("wL", C.c_void_p), ("wR", C.c_void_p)]


JOBS_SHARE_GEOMETRY = 1

# name -> (restype, argtypes); every symbol include/emagls.h declares
SYMBOLS = {
    "emagls_last_error": (C.c_char_p, []),
    "emagls_version": (C.c_int, []),
    "emagls_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "emagls_set_device": (C.c_int, [C.c_int]),
    "emagls_cache_clear": (C.c_int, []),
    "emagls_cache_release_designs": (C.c_int, []),
    "emagls_fp64_peak_tflops": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "emagls_fp64_peak_tflops_ex": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "emagls_self_test": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "emagls_sh_basis": (C.c_int, [C.c_int, c_i64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "emagls_sh_basis_device": (C.c_int, [C.c_int, c_i64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_modal_bn": (C.c_int, [C.c_int, c_i64, C.c_void_p, C.c_void_p]),
    "emagls_get_ls_filters": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "emagls_get_magls_filters": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_double, c_i64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    "emagls_get_emagls2_filters": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                             C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int,
                                             C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters_ema_in_ch": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                                      C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters_ema_in_sh": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                                      C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters_from_atf": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                                     C.c_double, c_i64, C.c_double, C.c_void_p, C.c_void_p,
                                                     C.POINTER(C.c_double)]),
    "emagls_simulation_order": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double]),
    "emagls_get_ls_filters_with_basis": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_magls_filters_with_basis": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_int, C.c_double, c_i64, C.c_int,
                                                      C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters_with_basis": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_double, C.c_void_p, c_i64, C.c_int,
                                                       C.c_double, c_i64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_emagls2_filters_with_basis": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_double, C.c_void_p, c_i64, C.c_int,
                                                        C.c_double, c_i64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_plan_set_basis": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_binaural_decode": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_void_p]),
    "emagls_binaural_decode_complex": (C.c_int, [C.c_void_p, C.c_int, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int, c_i64, C.c_int,
                                                 C.c_void_p, C.c_void_p]),
    "emagls_binaural_decode_device": (C.c_int, [C.c_void_p, C.c_int, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int, c_i64, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "emagls_get_magls_filters_dc": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_double, c_i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_get_emagls_filters_dc": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                               C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    "emagls_get_emagls2_filters_dc": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_double,
                                                C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p]),
    "emagls_get_magls_filters_2d": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_int, C.c_double, c_i64, C.c_int,
                                              C.c_void_p, C.c_void_p]),
    "emagls_get_radial_filter": (C.c_int, [C.c_int, C.c_double, C.c_double, c_i64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "emagls_apply_radial_filter_rows": (c_i64, [c_i64, c_i64, C.c_int]),
    "emagls_apply_radial_filter": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_double, C.c_double, c_i64, C.c_int, C.c_int, C.c_double,
                                             C.c_double, C.c_void_p]),
    "emagls_sh_encode": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "emagls_ch_basis": (C.c_int, [C.c_int, c_i64, C.c_void_p, C.c_int, C.c_void_p]),
    "emagls_get_smair_matrix": (C.c_int, [C.c_int, C.c_double, c_i64, C.c_int, C.c_double, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_int)]),
    "emagls_eq_filter_nfft": (c_i64, [c_i64]),
    "emagls_get_magls_spherical_head_filter": (C.c_int, [C.c_double, C.c_int, C.c_double, c_i64, C.c_void_p, C.c_void_p]),
    "emagls_get_magls_array_diffuse_filter": (C.c_int, [C.c_double, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double, c_i64, C.c_int,
                                                        C.c_void_p, C.c_void_p]),
    "emagls_plan_create": (C.c_int, [C.POINTER(DesignDesc), C.POINTER(C.c_void_p)]),
    "emagls_plan_destroy": (C.c_int, [C.c_void_p]),
    "emagls_plan_set_hrir_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_plan_set_mic_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_plan_set_hrirs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_plan_set_atfs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_plan_execute": (C.c_int, [C.c_void_p]),
    "emagls_plan_synchronize": (C.c_int, [C.c_void_p]),
    "emagls_plan_get_filters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emagls_plan_get_info": (C.c_int, [C.c_void_p, C.POINTER(PlanInfo)]),
    "emagls_plan_sweep_form_in_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "emagls_plan_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_plan_set_streams": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_plan_num_stages": (C.c_int, [C.c_void_p]),
    "emagls_plan_stage_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "emagls_plan_stage_times": (C.c_int, [C.c_void_p, c_dp, C.c_int]),
    "emagls_plan_sweep_kernel_time": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int)]),
    "emagls_plan_debug_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "emagls_plan_stream": (C.c_void_p, [C.c_void_p]),
    "emagls_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "emagls_batch_execute": (C.c_int, [C.c_void_p]),
    "emagls_batch_synchronize": (C.c_int, [C.c_void_p]),
    "emagls_batch_get_filters": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "emagls_batch_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_batch_lane_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "emagls_batch_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "emagls_set_batch_max": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "emagls_batch_shares_atf_side": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "emagls_design_hrir_sets": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "emagls_from_atf_hrir_sets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_double)]),
    "emagls_jobs_run": (C.c_int, [C.POINTER(Job), C.c_int64, C.c_int, C.c_int, C.c_int]),
    "emagls_design_out_shape": (C.c_int, [C.POINTER(DesignDesc), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(C.c_int)]),
    "emagls_jobs_shard": (C.c_int, [C.POINTER(Job), C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "emagls_jobs_run_devices": (C.c_int, [C.POINTER(Job), C.c_int64, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]),
    "emagls_jobs_set_profiling": (C.c_int, [C.c_int]),
    "emagls_jobs_sweep_times": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "emagls_batch_set_geometry_sharing": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_batch_shares_geometry": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "emagls_batch_set_side_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "emagls_batch_set_streams": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_batch_set_stage_order": (C.c_int, [C.c_void_p, C.c_int]),
    "emagls_batch_sweep_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "emagls_batch_destroy": (C.c_int, [C.c_void_p]),
}


class EmaglsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("emagls error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP library not built: %s is missing (run `python -m emagls_amd.build`)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise EmaglsError(rc, load().emagls_last_error().decode("utf-8", "replace"))
