"""Inputs of the skin-module tests (tests/test_skin_modules.py, tools/gen_skin_golden.py): the 13 columns aerobulk_amd/fortran/skin_driver.f90
reads, in its order, and the sequences of that driver restated as `ab_phymbl` calls (functions 57-60, include/aerobulk_amd.h)."""
import numpy as np

from phymbl_cases import _qsat_approx, read_records, write_input  # noqa: F401

COLUMNS = ["Qsw", "Qns", "us", "SST", "Qlat", "Tau", "lon", "ustk", "th", "qa", "U", "rlw", "slp"]
HOURS = (8, 10, 12, 14, 18, 23, 28)              # skin_driver.f90: ihours, rsun
RSUN = (0.4, 0.8, 1.0, 0.8, 0.05, 0.0, 0.0)
HWL_MAX, RD0 = 20.0, 3.0
CS_COARE, CS_ECMWF, WL_COARE, WL_ECMWF = 57, 58, 59, 60


def make_columns(n=256, seed=20261003):
    g = np.random.default_rng(seed)
    u = lambda lo, hi: g.uniform(lo, hi, n)
    c = {}
    c["Qsw"] = u(0.0, 950.0)
    c["Qsw"][::11] = 0.0
    c["Qns"] = u(-420.0, 40.0)
    c["us"] = 10.0 ** u(-2.6, -0.1)
    c["us"][::17] = 2.0e-5                        # below the 1e-4 floor of u* in the water
    c["SST"] = u(271.5, 305.0)
    c["Qlat"] = u(-320.0, 25.0)
    c["Tau"] = 10.0 ** u(-3.6, -0.1)              # some below the 0.002 N/m2 floor of the momentum integral
    c["lon"] = u(-180.0, 360.0)
    c["lon"][::5] = u(-20.0, 20.0)[::5]           # near Greenwich: the driver's 28 h (= 4h30 UTC) is a dawn there
    c["ustk"] = u(0.0, 0.25)
    c["ustk"][::13] = 0.0
    c["th"] = c["SST"] + u(-6.0, 4.0)
    c["slp"] = u(98000.0, 103000.0)
    c["qa"] = u(0.55, 0.95) * _qsat_approx(c["th"], c["slp"])
    c["U"] = u(0.3, 20.0)
    c["rlw"] = u(250.0, 450.0)
    return np.stack([c[k] for k in COLUMNS])


def col(cols, name):
    return cols[COLUMNS.index(name)]
