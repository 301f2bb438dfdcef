"""ctypes loader for libaerobulk_amd.so (the C ABI of include/aerobulk_amd.h).

The library is built in-tree by `python -m aerobulk_amd.build` (hipcc, gfx950).  There is no
CPU fallback: if the shared object is missing this module raises, and every compute entry
point returns AB_ERR_HIP when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AEROBULK_AMD_LIB") or os.path.join(PKG, "libaerobulk_amd.so")  # env override: A/B experiments only

vp = C.c_void_p
dp = C.POINTER(C.c_double)


class InitReport(C.Structure):
    _fields_ = [("n_cells", C.c_long), ("n_masked", C.c_long), ("hum_type", C.c_int), ("bad_field", C.c_int),
                ("bad_min", C.c_double), ("bad_max", C.c_double), ("bad_mean", C.c_double)]


class Diag(C.Structure):
    """ab_diag: 16 optional output pointers, in this order."""
    NAMES = ("Cd", "Ch", "Ce", "t_zu", "q_zu", "Ubzu", "CdN", "ChN", "CeN", "z0", "u_star", "L", "UN10", "dT_cs", "dT_wl", "Hz_wl")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


class TurbFields(C.Structure):
    """ab_turb_fields: the mandatory arguments of a TURB_* routine."""
    NAMES = ("T_s", "theta_zt", "q_s", "q_zt", "U_zu", "Qsw", "rad_lw", "slp", "Cd", "Ch", "Ce", "t_zu", "q_zu", "Ubzu")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


class IceFields(C.Structure):
    """ab_ice_fields: arguments of a TURB_ICE_* routine."""
    NAMES = ("Ts_i", "theta_zt", "qs_i", "q_zt", "U_zu", "frice", "Cd", "Ch", "Ce", "t_zu", "q_zu", "Ub",
             "CdN", "ChN", "CeN", "z0", "u_star", "L", "UN10", "CdN_frm")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


class ShardArrays(C.Structure):
    """ab_shard_arrays: one shard's device-resident rows (8 inputs, 6 outputs)."""
    NAMES = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw", "ql", "qh", "tau_x", "tau_y", "evap", "t_s")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


class FluxArrays(C.Structure):
    """ab_flux_arrays: whole-grid destinations of ab_session_gather."""
    NAMES = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


# every symbol include/aerobulk_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ab_algo_from_string": (C.c_int, [C.c_char_p, C.c_int]),
    "ab_algo_name": (C.c_char_p, [C.c_int]),
    "ab_strerror": (C.c_char_p, [C.c_int]),
    "ab_last_error": (C.c_char_p, []),
    "ab_device_count": (C.c_int, []),
    "ab_session_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ab_session_destroy": (C.c_int, [vp]),
    "ab_session_create_sharded": (C.c_int, [C.POINTER(vp), C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "ab_session_create_sharded_rows": (C.c_int, [C.POINTER(vp), C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                                 C.POINTER(C.c_long)]),
    "ab_session_shard_count": (C.c_int, [vp]),
    "ab_session_shard_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int)]),
    "ab_session_init": (C.c_int, [vp] + [vp] * 8 + [C.c_int, vp, C.POINTER(InitReport)]),
    "ab_session_init_stats": (C.c_int, [vp] + [vp] * 8 + [C.c_int, vp, dp]),
    "ab_session_init_apply": (C.c_int, [vp, dp, C.c_int, C.POINTER(InitReport)]),
    "ab_session_set_humidity": (C.c_int, [vp, C.c_int]),
    "ab_session_set_diagnostics": (C.c_int, [vp, C.POINTER(Diag), C.c_int]),
    "ab_session_turb": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(TurbFields), C.c_int, vp]),
    "ab_turb": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, dp] + [dp] * 8 + [dp] * 6
                + [C.POINTER(Diag), C.c_long, C.c_long]),
    "ab_turb_get_wl_state": (C.c_int, [C.c_int, dp, dp, dp, dp, C.c_long]),
    "ab_turb_neutral_10m": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, vp]),
    "ab_turb_ice": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(IceFields), C.c_long, C.c_int, C.c_int, vp]),
    "ab_turb_ice_easy": (C.c_int, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(IceFields), C.c_long,
                                   C.c_int, C.c_int, vp]),
    "ab_ice_algo_from_string": (C.c_int, [C.c_char_p]),
    "ab_session_compute": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_int] + [vp] * 8 + [vp] * 6 + [C.c_int, vp]),
    "ab_session_check": (C.c_int, [vp]),
    "ab_session_compute_shards": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(ShardArrays), C.POINTER(vp)]),
    "ab_session_gather": (C.c_int, [vp, C.c_int, C.POINTER(ShardArrays), C.POINTER(FluxArrays), C.POINTER(vp), C.c_int]),
    "ab_session_set_regroup": (C.c_int, [vp, C.c_int]),
    "ab_session_set_solar_time": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    "ab_session_get_wl_state": (C.c_int, [vp, dp]),
    "ab_session_last_kernel_ms": (C.c_double, [vp]),
    "ab_synth_fields_device": (C.c_int, [vp] * 8 + [C.c_long, C.c_long, C.c_long, C.c_int, vp]),
    "ab_test_math": (C.c_int, [C.c_int, dp, dp, dp, C.c_long]),
    "ab_calibrate": (C.c_int, [C.c_int, C.c_int, vp, dp, dp]),
    "ab_phymbl": (C.c_int, [C.c_int, C.c_long, C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_int, dp, C.c_int, C.c_int, vp, dp]),
    "ab_model": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_double] + [dp] * 6 + [dp] * 5
                 + [C.c_int, C.c_int, dp, dp, dp, C.c_long, C.c_long, C.POINTER(InitReport)]),
    "aerobulk_cxx_skin": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, dp, dp] + [dp] * 6 + [dp] * 5
                          + [C.POINTER(C.c_int), C.POINTER(C.c_bool), dp, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "aerobulk_cxx_no_skin": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, dp, dp] + [dp] * 6 + [dp] * 5
                             + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load the HIP engine; raises RuntimeError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP engine is not built (run `python -m aerobulk_amd.build`). "
                "aerobulk_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)  # AttributeError if the ABI is incomplete
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib
