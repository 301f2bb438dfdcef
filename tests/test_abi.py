"""CPU: the C-ABI library loads and exports every symbol include/aerobulk_amd.h declares; error paths that
need no GPU behave like the reference's (message text), and the engine refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from aerobulk_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from aerobulk_amd import build
        build.build_engine()
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "aerobulk_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ab_[a-z0-9_]+|aerobulk_cxx_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from aerobulk_amd import _lib
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aerobulk_amd.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in aerobulk_amd/_lib.py"


def test_cxx_api_symbols_exported(lib):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", os.path.join(ROOT, "aerobulk_amd", "libaerobulk_amd.so")], text=True)
    for sym in ("aerobulk::model(", "aerobulk::algorithm_to_string", "aerobulk::check_sizes("):
        assert sym in out, sym


def test_algorithm_names_round_trip(lib):
    # SELECT CASE(TRIM(calgo)), mod_aerobulk_compute.f90:129-176 ; aerobulk.cpp:22-49
    for i, n in enumerate(["coare3p0", "coare3p6", "ncar", "ecmwf", "andreas"], 1):
        assert lib.ab_algo_from_string(n.encode(), -1) == i
        assert lib.ab_algo_from_string((n + "   ").encode(), len(n) + 3) == i  # Fortran blank padding
        assert lib.ab_algo_name(i) == n.encode()
    assert lib.ab_algo_from_string(b"coare", -1) == 0
    assert lib.ab_algo_from_string(b"COARE3P6", -1) == 0  # the reference is case-sensitive
    assert lib.ab_algo_name(0) == b"other"


def test_argument_errors_need_no_gpu(lib):
    h = C.c_void_p()
    assert lib.ab_session_create(C.byref(h), 9, 10, 1, 1, 0, 0, -1) == 1            # AB_ERR_ALGO
    assert lib.ab_session_create(C.byref(h), 3, 10, 1, 1, 1, 0, -1) == 2            # skin with ncar: AB_ERR_SKIN_ALGO
    assert b"COARE*" in lib.ab_last_error()
    assert lib.ab_session_create(C.byref(h), 2, 0, 1, 1, 0, 0, -1) == 10            # AB_ERR_ARG
    assert lib.ab_strerror(8).startswith(b"wind stress")


def test_sharded_session_arguments_need_no_gpu(lib):
    h = C.c_void_p()
    devs = (C.c_int * 3)(0, 0, 0)
    assert lib.ab_session_create_sharded(C.byref(h), 2, 10, 2, 1, 0, 0, devs, 3) == 10     # more shards than rows: AB_ERR_ARG
    assert lib.ab_session_create_sharded(C.byref(h), 2, 10, 8, 1, 0, 0, None, 2) == 10     # no device list
    assert lib.ab_session_shard_count(None) == 0
    rows = (C.c_long * 3)(4, 3, 2)
    assert lib.ab_session_create_sharded_rows(C.byref(h), 2, 10, 8, 1, 0, 0, devs, 3, rows) == 10   # 9 rows for a grid of 8
    assert b"9 rows" in lib.ab_last_error()
    assert lib.ab_session_create_sharded_rows(C.byref(h), 2, 10, 8, 1, 0, 0, devs, 3, None) == 10
    if lib.ab_device_count() == 0:   # no GPU: the shards cannot be created, and the failure is loud
        assert lib.ab_session_create_sharded(C.byref(h), 2, 10, 8, 1, 0, 0, devs, 3) == 9
        assert lib.ab_session_create(C.byref(h), 2, 10, 8, 1, 0, 0, -2) == 9                # AB_DEVICE_ALL
        assert not h.value


def test_helper_entry_argument_errors_need_no_gpu(lib):
    """ab_phymbl (the mod_phymbl helpers, include/aerobulk_amd.h) rejects malformed calls with AB_ERR_ARG before it looks for a device."""
    import numpy as np
    x = np.full(8, 290.0)
    y = np.zeros(8)
    pin = (C.c_void_p * 3)(x.ctypes.data, x.ctypes.data, None)
    pout = (C.c_void_p * 2)(y.ctypes.data, None)
    par = (C.c_double * 2)(2.0, 0.0)
    f = lib.ab_phymbl
    ARG = 10
    assert f(0, 8, pin, 2, pout, 1, par, 0, 0, None, None) == ARG and b"unknown function" in lib.ab_last_error()
    assert f(99, 8, pin, 2, pout, 1, par, 0, 0, None, None) == ARG
    assert f(3, 0, pin, 2, pout, 1, par, 0, 0, None, None) == 0 and not y.any()        # virt_temp on no cells: a no-op, like the reference on zero-size arrays
    assert f(3, -1, pin, 2, pout, 1, par, 0, 0, None, None) == ARG
    assert f(3, 8, pin, 2, pout, 1, par, 0, 1, None, None) in (ARG, 9)                 # AB_MEM_DEVICE with host pointers: refused (no device: AB_ERR_HIP)
    assert f(3, 8, None, 2, pout, 1, par, 0, 0, None, None) == ARG
    assert f(3, 8, pin, 1, pout, 1, par, 0, 0, None, None) == ARG                      # virt_temp needs two arrays
    assert f(3, 8, pin, 2, pout, 1, par, 0, 7, None, None) == ARG and b"bad mem" in lib.ab_last_error()
    pin_hole = (C.c_void_p * 3)(x.ctypes.data, None, None)
    assert f(3, 8, pin_hole, 2, pout, 1, par, 0, 0, None, None) == ARG and b"required input" in lib.ab_last_error()
    pout_none = (C.c_void_p * 2)(None, None)
    assert f(3, 8, pin, 2, pout_none, 1, par, 0, 0, None, None) == ARG and b"no output" in lib.ab_last_error()
    if lib.ab_device_count() == 0:      # a well-formed call without a GPU: AB_ERR_HIP, the arrays untouched
        assert f(3, 8, pin, 2, pout, 1, par, 0, 0, None, None) == 9 and not y.any()


def test_calibrate_rejects_unknown_workloads_and_needs_a_gpu(lib):
    ms, rate = C.c_double(-1.), C.c_double(-1.)
    assert lib.ab_calibrate(7, 0, None, C.byref(ms), C.byref(rate)) == 10
    if lib.ab_device_count() == 0:
        assert lib.ab_calibrate(0, 0, None, C.byref(ms), C.byref(rate)) == 9           # AB_ERR_HIP: nothing runs on the host


def test_no_cpu_fallback(lib):
    """Without a visible GPU the product path must fail loudly (AB_ERR_HIP), never compute on the host."""
    if lib.ab_device_count() > 0:
        pytest.skip("a GPU is visible here")
    import numpy as np
    import aerobulk_amd as ab
    with pytest.raises(ab.AerobulkError) as e:
        ab.Session("coare3p6", 16)
    assert e.value.status == 9
    x = np.full(4, 290.0)
    with pytest.raises(ab.AerobulkError) as e:
        ab.aerobulk_model(1, 1, "ncar", 2.0, 10.0, x, x, x * 0 + 0.01, x * 0 + 5, x * 0, x * 0 + 1e5)
    assert e.value.status == 9


def test_model_protocol_errors(lib):
    import numpy as np
    import aerobulk_amd as ab
    x = np.full(4, 290.0)
    args = (x, x, x * 0 + 0.01, x * 0 + 5, x * 0, x * 0 + 1e5)
    with pytest.raises(ab.AerobulkError) as e:      # jt < 1, mod_aerobulk.f90:244
        ab.aerobulk_model(0, 1, "ncar", 2.0, 10.0, *args)
    assert e.value.status == 4
    with pytest.raises(ab.AerobulkError) as e:      # skin scheme for NCAR, mod_aerobulk.f90:69-70
        ab.aerobulk_model(1, 1, "ncar", 2.0, 10.0, *args, l_use_skin=True, rad_sw=x, rad_lw=x)
    assert e.value.status == 2
    with pytest.raises(ab.AerobulkError) as e:      # skin without radiation, mod_aerobulk.f90:72
        ab.aerobulk_model(1, 1, "coare3p6", 2.0, 10.0, *args, l_use_skin=True)
    assert e.value.status == 3
    with pytest.raises(ab.AerobulkError) as e:      # unknown algorithm, mod_aerobulk_compute.f90:173-175
        ab.aerobulk_model(1, 1, "coare", 2.0, 10.0, *args)
    assert e.value.status == 1
