#!/usr/bin/env python
"""Error of the three fp32-array session kinds against fp64 (run on the GPU box): AB_F32_MIXED (fp32 arrays, fp64 anchors and
differences, fp32 transcendentals: config 5's mode), AB_F32 (fp32 arithmetic throughout) and AB_F32_STORAGE (fp32 arrays, fp64
arithmetic), for the three algorithms with skin schemes; every cell of the grid, and the share of cells beyond 1e-4.
(Against the ORACLE instead of the fp64 HIP path: tests/test_gpu_mixed.py, on 360x180 and on a 4.5 M-cell subsample of 12960x10800.)

The reference is the fp64 path fed with the SAME fp32-rounded inputs (converted exactly): the error measured is the one the fp32
session adds, not the rounding of the caller's data.  Metric per flux x: |x32 - x64| / max(|x64|, 1 W/m2-equivalent floor)
i.e. relative where the flux exceeds the floor, absolute (in units of the floor) below — SURVEY §8d's proposal for config 5
(<= 1e-4 relative where |flux| > 1 W/m2, absolute 1e-4 below).  Floors: QL, QH 1 W/m2; tau 1e-3 N/m2; E 4e-7 kg/m2/s (1 W/m2 / Lv);
T_s: absolute error in K.  Prints a histogram of quantiles per field.

    python tools/fp32_error.py [360x180] [4320x3600]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
FLOOR = {"QL": 1.0, "QH": 1.0, "Tau_x": 1e-3, "Tau_y": 1e-3, "Evap": 4e-7, "T_s": None}
QS = (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)


def errors(algo, ni, nj, niter=5):
    import torch
    f32 = ab.synth_fields_device(ni, nj, precision="f32")
    f64 = {k: v.double() for k, v in f32.items()}          # the same numbers, exactly
    out = {}
    with ab.Session(algo, ni, nj, 1, True) as s:
        ref = s.compute(1, 2.0, 10.0, *[f64[k] for k in IN6], Niter=niter, rad_sw=f64["rad_sw"], rad_lw=f64["rad_lw"])
    for prec in ("f32_mixed", "f32", "f32_storage"):
        with ab.Session(algo, ni, nj, 1, True, precision=prec) as s:
            got = s.compute(1, 2.0, 10.0, *[f32[k] for k in IN6], Niter=niter, rad_sw=f32["rad_sw"], rad_lw=f32["rad_lw"])
        rows = {}
        for k, fl in FLOOR.items():
            d = (got[k].double() - ref[k]).abs()
            e = d if fl is None else d / ref[k].abs().clamp_min(fl)
            rows[k] = [float(torch.quantile(e[:: max(1, e.numel() // 4_000_000)], q)) if q < 1.0 else float(e.max()) for q in QS]
            rows[k].append(float((e > 1e-4).sum()) / e.numel())
        out[prec] = rows
    return out


def main():
    grids = [g for g in sys.argv[1:]] or ["360x180", "4320x3600"]
    for g in grids:
        ni, nj = (int(x) for x in g.split("x"))
        for algo in ("coare3p6", "coare3p0", "ecmwf"):
            res = errors(algo, ni, nj)
            for prec, rows in res.items():
                print(f"{algo} + skin, {g}, nb_iter=5, {prec}: error vs fp64 on the same (fp32-rounded) inputs; quantiles " + " ".join(f"p{q * 100:g}" for q in QS) + " | share of cells beyond 1e-4")
                for k, v in rows.items():
                    unit = "K (absolute)" if FLOOR[k] is None else f"relative, floor {FLOOR[k]:g}"
                    print(f"   {k:6s} " + " ".join(f"{x:9.2e}" for x in v[:-1]) + f" | {v[-1]:8.2e}   [{unit}]")


if __name__ == "__main__":
    main()
