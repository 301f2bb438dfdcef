#!/usr/bin/env python
"""Error of the fp32-array arithmetic modes against the fp64 ORACLE, without a GPU: the product's physics header compiled for the host
(tests/physics_host.cpp, hardware seeds emulated at their accuracy) in the arithmetic of AB_F32 (`f32`) and AB_F32_MIXED (`mixed`:
fp64 anchors SST / theta / T_s / q / q_s and their differences, fp32 elsewhere), fed like the oracle with the fp32-ROUNDED synthetic
fields.  Metric = tools/fp32_error.py: |x - ref| / max(|ref|, floor), floors 1 W/m2 and its equivalents (tau 1e-3 N/m2, E 4e-7),
T_s absolute in K; quantiles and the share of cells beyond 1e-4.  A design tool (which quantities must be anchors) and a CPU-side
regression of the mixed mode; the numbers that count are measured on the GPU (tools/fp32_error.py, profiles/r3_fp32_error.txt).

    python tools/mixed_error_host.py [360x180 ...] [--algos ecmwf,coare3p6] [--modes mixed,f32] [--niter 5] [--nt 1]
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
ALGOS = {"coare3p0": 1, "coare3p6": 2, "ncar": 3, "ecmwf": 4, "andreas": 5}
FLOOR = {"ql": 1.0, "qh": 1.0, "tau_x": 1e-3, "tau_y": 1e-3, "evap": 4e-7, "t_s": None}
ORDER = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
QS = (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)


def build(tmp):
    exe = os.path.join(tmp, "physics_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=fast", "-march=x86-64-v3", "-DAB_PSI_LDS_TABLES=1", *os.environ.get("AB_HOST_FLAGS", "").split(), "-o", exe,
                           os.path.join(ROOT, "tests", "physics_host.cpp")])
    return exe


def run_host(exe, tmp, algo, skin, niter, nt, zt, zu, f, mode):
    n = f["sst"].size
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("<5iq2d", ALGOS[algo], int(skin), niter, nt, 0, n, zt, zu))
        for k in IN8:
            np.ascontiguousarray(f[k], dtype=np.float64).tofile(fh)
    subprocess.check_call([exe, fin, fout, mode])
    return np.fromfile(fout).reshape(nt, 6, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("grids", nargs="*", default=["360x180"])
    ap.add_argument("--algos", default="ecmwf,coare3p6,coare3p0")
    ap.add_argument("--modes", default="mixed,f32")
    ap.add_argument("--niter", type=int, default=5)
    ap.add_argument("--nt", type=int, default=1)
    ap.add_argument("--noskin", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        for g in a.grids:
            ni, nj = (int(x) for x in g.split("x"))
            f = po.synth_fields(ni, nj)
            f64 = {k: f[k].astype(np.float32).astype(np.float64) for k in IN8}
            for algo in a.algos.split(","):
                skin = (not a.noskin) and algo in ("ecmwf", "coare3p6", "coare3p0")
                osess = po.OracleSession(algo, ni * nj, a.nt, skin)
                refs = []
                for jt in range(1, a.nt + 1):
                    o = osess.compute(jt, 2.0, 10.0, a.niter, *[f64[k] for k in IN8[:6]], rad_sw=f64["rad_sw"] if skin else None,
                                      rad_lw=f64["rad_lw"] if skin else None)
                    refs.append({k: o[k].copy() for k in ORDER if k in o})
                for mode in a.modes.split(","):
                    got = run_host(exe, tmp, algo, skin, a.niter, a.nt, 2.0, 10.0, f64, mode)
                    for jt in range(1, a.nt + 1):
                        print(f"{algo} skin={int(skin)} {g} nb_iter={a.niter} jt={jt} {mode} (host build) vs fp64 oracle on the fp32-rounded inputs; "
                              + " ".join(f"p{q * 100:g}" for q in QS) + " | share > 1e-4")
                        for i, k in enumerate(ORDER):
                            if k not in refs[jt - 1]:
                                continue
                            r = refs[jt - 1][k]
                            x = got[jt - 1, i]
                            if mode != "f64":
                                x = x.astype(np.float32).astype(np.float64)      # the session stores fp32
                            d = np.abs(x - r)
                            e = d if FLOOR[k] is None else d / np.maximum(np.abs(r), FLOOR[k])
                            qs = [float(np.quantile(e, q)) for q in QS]
                            print(f"   {k:6s} " + " ".join(f"{v:9.2e}" for v in qs) + f" | {float((e > 1e-4).mean()):.2e}"
                                  + ("   [K]" if FLOOR[k] is None else ""))


if __name__ == "__main__":
    main()
