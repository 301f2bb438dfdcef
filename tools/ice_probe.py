#!/usr/bin/env python
"""TURB_ICE_* on random cells (ice temperature 230-273 K, air within +-15 K of it, humidity 20-100 % of saturation over ice, wind 0-30
m/s, ice fraction 0-1) against the C restatement (oracle.pyoracle.oracle_turb_ice), several heights and pass counts.  GPU box.

    python tools/ice_probe.py [n_cells] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aerobulk_amd as ab  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    r = np.random.default_rng(seed)
    ts = r.uniform(230.0, 273.15, n)
    th = ts + r.uniform(-15.0, 15.0, n)
    th[::101] = ts[::101]
    esi = 611.15 * np.exp(22.452 * (ts - 273.16) / (ts - 0.61))
    qs = 0.622 * esi / (100000.0 - 0.378 * esi)
    f = dict(Ts_i=ts, theta_zt=th, qs_i=qs, q_zt=qs * r.uniform(0.2, 1.3, n), U_zu=r.uniform(0.0, 1.0, n) ** 2 * 30.0, frice=r.uniform(0.0, 1.0, n))
    f["U_zu"][::57] = 0.0
    f["frice"][::13] = 1.0
    f["frice"][5::13] = 0.0
    worst_all = 0.0
    for algo in ("nemo", "an05", "lu12", "lg15", "lg15_io", "easy"):
        for zt, zu, niter in ((2.0, 10.0, 5), (10.0, 10.0, 8), (20.0, 8.0, 3), (2.0, 10.0, 1)):
            ref = po.oracle_turb_ice(algo, niter, zt, zu, f)
            opt = [k for k in ref if k not in ("Cd", "Ch", "Ce", "t_zu", "q_zu", "Ub", "Ubzu")]
            o = ab.turb_ice(algo, zt, zu, f["Ts_i"], f["theta_zt"], f["qs_i"], f["q_zt"], f["U_zu"],
                            frice=f["frice"] if algo in ("lu12", "lg15", "lg15_io") else None, nb_iter=niter, optional=tuple(opt),
                            cxn=(1.5e-3, 1.3e-3, 1.4e-3) if algo == "easy" else None)
            line, worst = [], 0.0
            for k in ref:
                g, rf = np.asarray(o["Ub" if k == "Ubzu" else k]), ref[k]
                if k == "L":
                    g, rf = 1.0 / g, 1.0 / rf
                fin = np.isfinite(rf)
                assert np.array_equal(np.isfinite(g), fin), (algo, k, "finite pattern differs")
                e = np.abs(g[fin] - rf[fin]) / np.maximum(np.abs(rf[fin]), 1e-6 * np.abs(rf[fin]).max())
                worst = max(worst, float(e.max()))
                if e.max() > 1e-10:
                    line.append(f"{k} {e.max():.1e} (n>1e-10: {(e > 1e-10).sum()})")
            worst_all = max(worst_all, worst)
            print(f"{algo} zt={zt} zu={zu} n={niter}: worst {worst:.1e} " + "; ".join(line), flush=True)
    print(f"cells {n}, worst of all {worst_all:.2e}")


if __name__ == "__main__":
    main()
