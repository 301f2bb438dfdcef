#!/usr/bin/env python
"""Kernel time per record of a short series (warm-layer state carried between records), regrouping on / off.

    python tools/series_probe.py [--grid 4320x3600] [--algos ecmwf,coare3p6] [--nt 4]
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
    ap.add_argument("--algos", default="ecmwf,coare3p6")
    ap.add_argument("--nt", type=int, default=4)
    a = ap.parse_args()
    ni, nj = (int(x) for x in a.grid.split("x"))
    f = ab.synth_fields_device(ni, nj)
    with ab.Session("coare3p6", ni, nj, 1, False) as s:      # clock ramp
        for _ in range(60):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
    for algo in a.algos.split(","):
        for regroup in (1, 0):
            best = [1e9] * a.nt
            for _ in range(3):
                with ab.Session(algo, ni, nj, a.nt, True) as s:
                    s.set_regroup(regroup)
                    for jt in range(1, a.nt + 1):
                        s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], check=False)
                        best[jt - 1] = min(best[jt - 1], s.last_kernel_ms())
            print(f"{algo:9s} regroup={regroup} " + " ".join(f"jt{j + 1}={m:7.3f}ms" for j, m in enumerate(best)), flush=True)


if __name__ == "__main__":
    main()
