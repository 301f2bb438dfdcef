#!/usr/bin/env python
"""What the REFERENCE's arithmetic asks for, per cell: calls of exp / log / log10 / pow / sqrt / atan made by the oracle (the line-by-line
C restatement of the reference, oracle/ab_oracle.c) on the benchmark fields, counted by compiling it with counting macros — and what those
calls would cost at the best per-function price measured on the MI355X (issue slots of the engine's own fp64 functions,
aerobulk_amd/csrc/ab_fastmath.hpp; one slot = one fp64 FMA of a wave = 4 cycles).  That product is the floor of any implementation that
evaluates the reference's formulas function by function; next to it the kernel's measured issue slots (tools/isa_profile.py).

    python tools/oracle_opcount.py [--grid 432x360] [--configs coare3p6:1:5,...]  [-o profiles/r4_opcount_floor.txt]     (CPU only)
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FUNCS = ("exp", "log", "log10", "pow", "sqrt", "atan")
# best measured price of one call, in issue slots (ab_fastmath.hpp: qlog 3.5 + 13; qexp 15; pow_pos = log + exp; qsqrt_pos: v_rsq_f64 (4) + 6;
# qatan_ge1: a division (v_rcp_f64 4 + 4) + degree-10 polynomial in t^2 + 4; profiles/r2_instr_rates.txt, r3_instr_rates_more.txt)
PRICE = {"exp": 15.0, "log": 16.5, "log10": 17.5, "pow": 31.5, "sqrt": 10.0, "atan": 25.0}
HDR = r"""
#include <math.h>
extern long ab_cnt[8];
#define exp(x)   (ab_cnt[0]++, exp(x))
#define log(x)   (ab_cnt[1]++, log(x))
#define log10(x) (ab_cnt[2]++, log10(x))
#define pow(x,y) (ab_cnt[3]++, pow(x,y))
#define sqrt(x)  (ab_cnt[4]++, sqrt(x))
#define atan(x)  (ab_cnt[5]++, atan(x))
"""


def build(td):
    hdr = os.path.join(td, "cnt.h")
    open(hdr, "w").write(HDR)
    cnt = os.path.join(td, "cnt.c")
    open(cnt, "w").write("long ab_cnt[8];\n")
    so = os.path.join(td, "liboracle_cnt.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-include", hdr, "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "oracle", "ab_oracle.c"), cnt, "-o", so, "-lm"])
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="432x360")
    ap.add_argument("--configs", default="coare3p6:1:5,coare3p6:0:8,coare3p0:1:5,ecmwf:1:5,ecmwf:0:5,andreas:0:5,ncar:0:5")
    ap.add_argument("--measured", default="coare3p6:1:5=4715", help="measured issue slots per cell of the kernels (tools/isa_profile.py), cfg=slots,...")
    ap.add_argument("-o", "--out", default=None)
    a = ap.parse_args()
    ni, nj = (int(x) for x in a.grid.split("x"))
    from oracle import pyoracle as po
    measured = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.measured.split(",") if kv)
    with tempfile.TemporaryDirectory() as td:
        so = build(td)
        po.ORACLE_SO = so
        po._libs.clear()
        L = po.lib()
        cnt = (C.c_long * 8).in_dll(L, "ab_cnt")
        # every (4320/ni)-th column and (3600/nj)-th row of the benchmark grid: the same mix of stable / unstable / day / night cells
        f = po.synth_fields(4320, 3600)
        sel = (np.arange(nj)[:, None] * (3600 // nj) * 4320 + np.arange(ni)[None, :] * (4320 // ni)).ravel()
        f = {k: np.ascontiguousarray(v[sel]) for k, v in f.items()}
        n = sel.size
        rows = []
        for cfg in a.configs.split(","):
            algo, skin, niter = cfg.split(":")
            skin, niter = skin == "1", int(niter)
            for i in range(8):
                cnt[i] = 0
            s = po.OracleSession(algo, n, 1, skin)
            o = s.compute(1, 2.0, 10.0, niter, *[f[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")],
                          rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
            assert o["rc"] == 0
            per = {fn: cnt[i] / n for i, fn in enumerate(FUNCS)}
            floor = sum(per[fn] * PRICE[fn] for fn in FUNCS)
            rows.append((cfg, per, floor))
    L_ = []
    p = L_.append
    p("tools/oracle_opcount.py: libm calls per cell of the oracle (= the reference's formulas, line by line) on every "
      f"{4320 // ni}th x {3600 // nj}th cell of the 4320x3600 benchmark fields ({n} cells), zt = 2, zu = 10,")
    p("priced at the engine's best per-function cost on the MI355X (issue slots; one slot = one fp64 FMA of a wave = 4 cycles): "
      + ", ".join(f"{fn} {PRICE[fn]:g}" for fn in FUNCS) + ".")
    p("Divisions, the fp64 multiply-adds around the calls, MAX / MIN / SIGN and the bulk formula are NOT in this floor (they add ~35 % in the kernel).")
    p("")
    p(f"{'configuration':18s} " + " ".join(f"{fn:>7s}" for fn in FUNCS) + "   calls   function-by-function floor [slots/cell]   kernel measured [slots/cell]")
    for cfg, per, floor in rows:
        m = measured.get(cfg)
        p(f"{cfg:18s} " + " ".join(f"{per[fn]:7.1f}" for fn in FUNCS) + f"  {sum(per.values()):6.1f}   {floor:10.0f}"
          + (f"                                 {m:8.0f}  ({m / floor:.2f} of the floor)" if m else ""))
    txt = "\n".join(L_) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
