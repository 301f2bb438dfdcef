#!/usr/bin/env python
"""Kernel-only timing sweep (device-resident synthetic fields): ms per launch and per-iteration cost.

    python tools/kbench.py [--grid 4320x3600] [--algos coare3p6,...] [--iters 0,4,8] [--precision f64]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="4320x3600")
    ap.add_argument("--algos", default="coare3p6,coare3p0,ecmwf,ncar,andreas")
    ap.add_argument("--iters", default="0,5,8")
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--regroup", type=int, default=1)
    a = ap.parse_args()
    ni, nj = (int(x) for x in a.grid.split("x"))
    f = ab.synth_fields_device(ni, nj, precision=a.precision)
    iters = [int(x) for x in a.iters.split(",")]
    with ab.Session("coare3p6", ni, nj, 1, False, precision=a.precision) as s:   # clocks ramp up during the first ~100 ms of work
        for _ in range(60):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
        s.last_kernel_ms()
    for algo in a.algos.split(","):
        for skin in ((False, True) if algo in ("coare3p0", "coare3p6", "ecmwf") else (False,)):
            with ab.Session(algo, ni, nj, 1, skin, precision=a.precision) as s:
                s.set_regroup(a.regroup)
                ms = []
                for it in iters:
                    best = 1e9
                    for _ in range(a.reps):
                        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=it, rad_sw=f["rad_sw"] if skin else None,
                                  rad_lw=f["rad_lw"] if skin else None, check=False)
                        best = min(best, s.last_kernel_ms())
                    ms.append(best)
                per = (ms[-1] - ms[0]) / max(iters[-1] - iters[0], 1)
                cells = ni * nj
                # nb_iter = 5 is measured, not extrapolated: WL_COARE only runs on the iterations that divide nb_iter
                n5 = ms[iters.index(5)] if 5 in iters else ms[0] + 5 * per
                print(f"{algo:9s} skin={int(skin)} " + " ".join(f"n{it}={m:8.3f}ms" for it, m in zip(iters, ms))
                      + f"  per-iter={per:7.3f}ms  n5={n5:8.3f}ms -> {cells / n5 / 1e3:8.1f} Mcell/s", flush=True)


if __name__ == "__main__":
    main()
