#!/usr/bin/env python
"""TURB_* called directly on random cells with random longitudes and a record sequence that crosses dawn (the warm layer's 4 h - 6.5 h
local-solar-time reset, mod_skin_coare.f90:146-163), cool skin and warm layer switched separately: ab_session_turb against the C
restatement (oracle.pyoracle.oracle_turb_series).  Run on the GPU box.

    python tools/turb_series_probe.py [n_cells] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aerobulk_amd as ab  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_fuzz import _fields  # noqa: E402

NAMES = ("Cd", "Ch", "Ce", "t_zu", "q_zu", "Ubzu", "CdN", "ChN", "CeN", "z0", "u_star", "L", "UN10", "dT_cs", "dT_wl", "Hz_wl", "T_s", "q_s")
OPT = NAMES[6:16]
WELL = ("Cd", "t_zu", "q_zu", "Ubzu", "CdN", "z0", "u_star", "UN10", "dT_cs", "dT_wl", "Hz_wl", "T_s", "q_s")   # not divided by dq, dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    r = np.random.default_rng(seed)
    f = _fields(seed, n)
    keep = np.hypot(f["u_zu"], f["v_zu"]) < 25.0
    f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
    n = f["sst"].size
    lon = r.uniform(-180.0, 360.0, n)
    lon[::11] = np.round(lon[::11] / 15.0) * 15.0           # whole hours of solar time
    isec = np.array([3600 * 1, 3600 * 3 + 1800, 3600 * 5, 3600 * 6 + 1799, 3600 * 9, 3600 * 14, 3600 * 21, 86399, 0, 3600 * 4])
    nt = isec.size
    qs0 = 0.98 * 0.622 * 611.2 * np.exp(17.62 * (f["sst"] - 273.15) / (f["sst"] - 30.03)) / f["slp"]
    recs = np.empty((nt, 8, n))
    for jt in range(nt):
        sw = f["rad_sw"] * max(0.0, np.sin(np.pi * (isec[jt] / 86400.0)))      # some diurnal cycle
        wind = np.hypot(f["u_zu"], f["v_zu"]) * (1.0 + 0.1 * np.sin(jt + np.arange(n)))
        recs[jt] = [f["sst"] + 0.02 * jt, f["t_zt"], qs0, f["hum_zt"], wind, sw, f["rad_lw"], f["slp"]]
    worst_all = 0.0
    for algo in ("coare3p6", "coare3p0", "ecmwf"):
        for cs, wl in ((True, True), (True, False), (False, True), (False, False)):
            for zt, zu, niter in ((2.0, 10.0, 5), (15.0, 10.0, 4)):
                ref = po.oracle_turb_series(algo, cs, wl, niter, zt, zu, lon, isec, recs)
                skin = cs or wl
                got = np.zeros_like(ref)
                with ab.Session(algo, n, 1, nt, False) as s:
                    d = s.set_diagnostics(OPT)
                    for jt in range(nt):
                        rr = recs[jt]
                        s.set_solar_time(int(isec[jt]), lon)
                        T_s, q_s = rr[0].copy(), rr[2].copy()
                        o = s.turb(jt + 1, zt, zu, T_s, rr[1].copy(), q_s, rr[3].copy(), rr[4].copy(), cs, wl, Qsw=rr[5].copy() if skin else None,
                                   rad_lw=rr[6].copy() if skin else None, slp=rr[7].copy() if skin else None, nb_iter=niter)
                        for i, k in enumerate(NAMES[:6]):
                            got[jt, i] = o[k]
                        for i, k in enumerate(OPT):
                            got[jt, 6 + i] = d[k]
                        if not cs:
                            got[jt, 13] = 0.0
                        if not wl:
                            got[jt, 14:16] = 0.0
                        got[jt, 16], got[jt, 17] = T_s, q_s
                line, worst = [], 0.0
                for k in WELL:
                    i = NAMES.index(k)
                    g, rf = got[:, i], ref[:, i]
                    if not np.any(rf):
                        assert not np.any(g), (algo, cs, wl, k)
                        continue
                    e = np.abs(g - rf) / np.maximum(np.abs(rf), 1e-6 * np.abs(rf).max())
                    worst = max(worst, float(e.max()))
                    if e.max() > 1e-10:
                        jt, c = np.unravel_index(np.argmax(e), e.shape)
                        line.append(f"{k} {e.max():.1e} (record {jt} cell {c}: ref {rf[jt, c]:.6e} got {g[jt, c]:.6e}; n>1e-10: {(e > 1e-10).sum()})")
                worst_all = max(worst_all, worst)
                print(f"{algo} cs={int(cs)} wl={int(wl)} zt={zt} zu={zu} n={niter}: worst {worst:.1e} " + "; ".join(line), flush=True)
    print(f"cells {n}, records {nt}, worst of all {worst_all:.2e}")


if __name__ == "__main__":
    main()
