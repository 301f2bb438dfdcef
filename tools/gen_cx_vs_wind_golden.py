#!/usr/bin/env python3
"""Golden data on the grid of the reference's own sweep driver src/tests/test_cx_vs_wind.f90: 1201 winds (steps of wind_max / 1200 = 1/24 m/s: quarter
steps below 5 m/s, double steps above 30, which ends the table at 40 m/s, :96-105) x 75 air-sea differences of virtual potential temperature (:114-120) x 7 relative
humidities (:19), SST = 22 C, zt = 2 m, zu = 10 m, nb_iter = 20 (:77), no skin scheme — its densest exercise of the Charnock ramps, the
LKB bins and NCAR's 33 m/s threshold.  The (theta, q) couples come from the driver's fixed point FIND_COUPLES (:456-516) restated here;
the 630 525 cells go through the UNMODIFIED reference's TURB_* in one call per algorithm (oracle/_ref/ref_series_driver.x = our
turb_series_driver.f90 linked with the reference modules) at full precision (the driver itself prints f16.8).

Stored (tests/golden/cx_vs_wind.npz, data only): the inputs, and per algorithm the driver's product — the mean over the seven
humidities, accumulated in its order (:262-277) — of Cd, Ch, Ce, z0, u* for ALL winds and every fourth temperature difference
(19 of 75, 0.0 among them).  Build container only.

    python tools/gen_cx_vs_wind_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

VRH = np.array([70., 75., 80., 85., 90., 95., 100.])
T_DVT = np.array([-15., -13., -12., -11., -10., -9.5, -9., -8.5, -8., -7.5, -7., -6.5,
                  -6., -5.5, -5., -4.5, -4., -3.5, -3., -2.5, -2., -1.75, -1.5, -1.25,
                  -1., -0.75, -0.5, -0.4, -0.35, -0.3, -0.25, -0.2, -0.15, -0.12, -0.1, -0.07, -0.05,
                  -0.01, 0.0, 0.01, 0.05, 0.07, 0.1, 0.12, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.5, 0.75,
                  1., 1.25, 1.5, 1.75, 2., 2.5, 3., 3.5, 4., 4.5, 5., 5.5,
                  6., 6.5, 7., 7.5, 8., 8.5, 9., 9.5, 10., 11., 12.])
N_W, WIND_MAX, PATM, RT0, RCTV0 = 1201, 50., 101000., 273.15, 461.495 / 287.05 - 1.
ALGOS = ("coare3p0", "coare3p6", "ncar", "ecmwf", "andreas")
KEEP = np.arange(2, 75, 4)                       # 19 of the 75 differences; T_DVT[38] = 0.0 is one of them
FIELDS = {"Cd": 0, "Ch": 1, "Ce": 2, "z0": 9, "us": 10}      # planes of the series driver's output


def winds():
    dw0 = WIND_MAX / (N_W - 1)
    w = np.zeros(N_W)
    for j in range(1, N_W):
        dw = 0.25 * dw0
        if w[j - 1] >= 5.:
            dw = dw0
        if w[j - 1] >= 30.:
            dw = 2. * dw0
        w[j] = w[j - 1] + dw
    return w


def find_couples(L, sstk, dvt):
    """(theta, q) for every humidity of VRH such that the virtual temperature is that of the sea surface + dvt."""
    sstv = sstk * (1. + RCTV0 * L.abo_q_sat(sstk, PATM))
    out = np.empty((2, VRH.size))
    for jh, rh in enumerate(VRH):
        tv = sstv + dvt
        ta, diff = tv, 10.
        while diff > 1.e-7:
            told = ta
            qa = L.abo_q_air_rh(rh, ta, PATM)
            ta = tv / (1. + RCTV0 * qa)
            diff = abs(ta - told)
        out[:, jh] = (ta, qa)
    return out


def make_inputs(sst_c=22.):
    L = po.lib()
    sstk = sst_c + RT0
    w = winds()
    cpl = np.stack([find_couples(L, sstk, d) for d in T_DVT])            # [75, 2, 7]
    qsat_sst = 0.98 * L.abo_q_sat(sstk, PATM)
    return sstk, qsat_sst, w, cpl


def cells(sstk, qsat_sst, w, cpl):
    """The driver's loop nest (dT, wind, humidity) flattened, humidity fastest: the 8 planes of the series driver's input."""
    ndt, nw, nh = T_DVT.size, w.size, VRH.size
    n = ndt * nw * nh
    tht = np.broadcast_to(cpl[:, 0][:, None, :], (ndt, nw, nh)).reshape(n)
    q = np.broadcast_to(cpl[:, 1][:, None, :], (ndt, nw, nh)).reshape(n)
    ww = np.broadcast_to(w[None, :, None], (ndt, nw, nh)).reshape(n)
    one = np.ones(n)
    return np.stack([sstk * one, tht, qsat_sst * one, q, ww, 0. * one, 0. * one, PATM * one])


def rh_mean(x):
    """mX = mX + X/nrh over jh = 1..7, in that order (test_cx_vs_wind.f90:262-277)."""
    x = x.reshape(T_DVT.size, N_W, VRH.size)
    m = np.zeros(x.shape[:2])
    for jh in range(VRH.size):
        m = m + x[:, :, jh] / VRH.size
    return m


def main():
    if not po.have_reference():
        sys.exit("oracle/_ref is not built: make -C oracle all (needs /root/reference)")
    sstk, qsat_sst, w, cpl = make_inputs()
    rec = cells(sstk, qsat_sst, w, cpl)
    n = rec.shape[1]
    out = {"sstk": sstk, "qsat_sst": qsat_sst, "winds": w, "couples": cpl, "t_dvt": T_DVT, "vrh": VRH, "keep": KEEP, "nb_iter": 20}
    for algo in ALGOS:
        o = po.run_reference_series(algo, 0, 0, 20, 2.0, 10.0, np.zeros(n), np.array([43200.]), rec[None])[0]     # [18, n]
        for name, plane in FIELDS.items():
            out[f"{algo}_{name}"] = rh_mean(o[plane])[KEEP]
        print(algo, "Cd x1000 at dT=0, 10 m/s:", 1000. * out[f"{algo}_Cd"][list(KEEP).index(38), np.argmin(abs(w - 10.))])
    path = os.path.join(ROOT, "tests", "golden", "cx_vs_wind.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB,", n, "cells per algorithm")


if __name__ == "__main__":
    main()
