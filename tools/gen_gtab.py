#!/usr/bin/env python
"""tools/gen_gtab.py — piecewise tables read through the VECTOR-MEMORY path (aerobulk_amd/csrc/ab_gtables.hpp).

The flux kernels are bound by VALU issue with HBM at 8 % and the L1 / L2 idle; LDS is full.  A table laid out
INTERVAL-MAJOR — the eight coefficients of an interval in 64 contiguous bytes — costs one address computation and four
global_load_dwordx4 that hit L1: measured equal to the same table in LDS (profiles/r3_notes.md: e_sat table, +-0.5 %), where the
coefficient-major layout of the LDS tables (eight scattered 8-byte loads, a 64-bit address each) cost +2...+6 %.  So the long tables
LDS has no room for live in global memory:

  kGPsiCoareM / H   COARE's WHOLE unstable profile functions (mod_common_coare.f90:235-252, :326-342)
                        psi = (1 - f) psi_Kansas + f psi_convective,  f = zeta^2 / (1 + zeta^2)
                    as functions of s = LOG(|1 - 15 zeta|), 107 intervals of 1/16 on [0, 6.6875) (zeta >= -50).  One log and two
                    table evaluations on a shared index replace three logs, a table, a degree-22 polynomial, a second table and the
                    blend with its division.  (In s the blend's poles zeta = +-i sit 1.5 from the real axis and the branch points of
                    the convective arguments pi: intervals of 1/16 put the degree-7 interpolation error at the rounding of the
                    coefficients; the LDS of a block has no room for 2 x 6.7 KB, the L1 does.)
  kGPsikM / H       Kansas / Paulson psi_m and psi_h in s = LOG(|1 - 16 zeta|) (ECMWF, NCAR, ANDREAS), 28 intervals on [0, 6.6875)
  kLPsiCoareM / H   the same blended psi_m / psi_h of COARE as functions of y = |1 - 15 zeta| itself, 1 <= y < 1024 (zeta > -68), indexed by the
                    BITS of y: interval = exponent and top three mantissa bits (8 intervals per binade, 10 binades = 80 intervals), local
                    variable from the remaining mantissa bits — no logarithm, no floor: a shift, a subtraction, a mask and one FMA.  Degree 9,
                    interval-major rows of ten doubles (80 B).  The logarithm that the tables in s need (one per evaluation, eleven per
                    unstable cell of the headline kernel: ~20 issue slots each) is gone.  6.4 KB each: in LDS where a whole CU shares one
                    copy (flux_kernel_cu), else through L1 — the same numbers and the same polynomial either way.
  kLWlAbs           WL_COARE's absorbed fraction 1 - sum_i c_i a_i (1 - exp(-H/a_i))/H of the solar flux in a warm layer of depth H
                    (mod_skin_coare.f90:167-168, 205-207; three exponentials and a division, up to ten times per cell), on 0.0625 <= H < 32
                    (the scheme clamps H to [0.1, 20]), indexed by the bits of H the same way: 72 intervals, degree 9, 1.0e-16 relative.
  kGCsG             the cool skin's absorption profile g(u) = (1 - exp(-u))/u, u = delta/8e-4 (mod_phymbl.f90:2030-2044 via
                    CS_COARE / CS_ECMWF: zfr = c0 + 11 delta - 6.6e-5/delta (1 - exp(-delta/8e-4))), 64 intervals of 1/8 on [0, 8)

Each interval: degree-7 interpolant at the Chebyshev nodes (60-digit arithmetic), monomial coefficients in the local variable
u in [-1, 1), c0 first.  The reference's literals are taken as the doubles they are at run time.  Prints the header.
"""
import sys

import mpmath as mp

mp.mp.dps = 60
DEG = 7
RPI = mp.mpf(float("3.141592653589793"))      # rpi of the reference (mod_const.f90:39)
SMAX = mp.mpf("6.6875")


def D(x):
    return mp.mpf(float(x))


def psic(y):
    """COARE's convective profile at y = |1 - a zeta| (a = 10.15 for psi_m, 34.15 for psi_h)"""
    c = mp.power(y, D(0.3333))
    S3 = D(1.7320508)
    return mp.mpf("1.5") * mp.log((1 + c + c * c) / 3) - S3 * mp.atan((1 + 2 * c) / S3) + D(1.813799447)


def psik_m(s):
    x = mp.exp(s / 4)
    return 2 * mp.log((1 + x) / 2) + mp.log((1 + x * x) / 2) - 2 * mp.atan(x) + RPI / 2


def psik_h(s):
    return 2 * mp.log((1 + mp.exp(s / 2)) / 2)


def coare(which):
    a_c = D(10.15) if which == "m" else D(34.15)
    pk = psik_m if which == "m" else psik_h

    def f(s):
        zeta = -(mp.exp(s) - 1) / 15
        zf = zeta * zeta
        zf = zf / (1 + zf)
        return (1 - zf) * pk(s) + zf * psic(abs(1 - a_c * zeta))
    return f


def psim_andreas_stable(s):
    """psi_m of ANDREAS on the stable side (Grachev et al. 2007; mod_blk_andreas.f90:321-350) as a function of s = LOG(1 + zeta)"""
    am, bm = D(5.), D(5.) / D(6.5)
    B = mp.cbrt((1 - bm) / bm)
    x = mp.exp(s / 3)
    return (-3 * am / bm * (x - 1)
            + am * B / (2 * bm) * (mp.log((x + B) ** 2 / (x * x - x * B + B * B)) - mp.log((1 + B) ** 2 / (1 - B + B * B))
                                   + 2 * mp.sqrt(3) * (mp.atan((2 * x - B) / (mp.sqrt(3) * B)) - mp.atan((2 - B) / (mp.sqrt(3) * B)))))


def g_cs(u):
    if u == 0:
        return mp.mpf(1)
    return -mp.expm1(-u) / u


def wl_abs(H):
    """absorbed fraction of the solar flux in a warm layer of depth H (WL_COARE, mod_skin_coare.f90:167-168, 205-207), the products of
    the literals taken as the doubles the kernel's closed form uses"""
    a1, a2, a3 = D(0.014), D(0.357), D(12.82)
    c1, c2, c3 = D(0.28 * 0.014), D(0.27 * 0.357), D(0.45 * 12.82)
    return 1 - (c1 * (-mp.expm1(-H / a1)) + c2 * (-mp.expm1(-H / a2)) + c3 * (-mp.expm1(-H / a3))) / H


def local_fit(f, a, b):
    n = DEG + 1
    us = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, u in enumerate(us):
        for j in range(n):
            A[i, j] = u ** j
        y[i] = f((a + b) / 2 + (b - a) / 2 * u)
    c = mp.lu_solve(A, y)
    return [float(c[j]) for j in range(n)]


def table(f, x0, x1, nint):
    rows, worst = [], mp.mpf(0)
    for i in range(nint):
        a, b = x0 + (x1 - x0) * mp.mpf(i) / nint, x0 + (x1 - x0) * mp.mpf(i + 1) / nint
        c = local_fit(f, a, b)
        rows.append(c)
        for k in range(33):
            u = mp.mpf(-1) + mp.mpf(2) * k / 32
            p = mp.mpf(0)
            for cc in reversed(c):
                p = p * u + mp.mpf(cc)
            worst = max(worst, abs(p - f((a + b) / 2 + (b - a) / 2 * u)))
    return rows, float(worst)


def emit(out, name, rows, err, what):
    out.append(f"// {what}: {len(rows)} intervals, max |table - function| = {err:.2e}")
    out.append(f"AB_TAB double {name}[{8 * len(rows)}] = {{")
    flat = [c for r in rows for c in r]
    for j in range(0, len(flat), 4):
        out.append("    " + ", ".join(repr(v) for v in flat[j:j + 4]) + ("," if j + 4 < len(flat) else "};"))


def main():
    out = ["// ab_gtables.hpp — GENERATED by tools/gen_gtab.py: piecewise tables read through the vector-memory path (interval-major: the eight",
           "// degree-7 coefficients of an interval in 64 contiguous bytes, c0 first).  See that file for what each table is and why it lives",
           "// in global memory rather than LDS.  Included by ab_physics.hpp.",
           "#pragma once", "namespace ab {",
           f"constexpr double kGPsiSMax = {float(SMAX)!r};", "constexpr int kGPsiCoareN = 107, kGPsikN = 28, kGCsGN = 64;", "constexpr double kGCsGMax = 8.;"]
    for which, name in (("m", "kGPsiCoareM"), ("h", "kGPsiCoareH")):
        rows, err = table(coare(which), mp.mpf(0), SMAX, 107)
        emit(out, name, rows, err, f"COARE unstable psi_{which} (Kansas / convective blend) in s = LOG(|1 - 15 zeta|)")
        print(name, err, file=sys.stderr)
    for f, name, what in ((psik_m, "kGPsikM", "psi_m"), (psik_h, "kGPsikH", "psi_h")):
        rows, err = table(f, mp.mpf(0), SMAX, 28)
        emit(out, name, rows, err, f"Kansas / Paulson {what} in s = LOG(|1 - 16 zeta|)")
        print(name, err, file=sys.stderr)
    rows, err = table(g_cs, mp.mpf(0), mp.mpf(8), 64)
    emit(out, "kGCsG", rows, err, "cool-skin absorption profile g(u) = (1 - exp(-u))/u")
    print("kGCsG", err, file=sys.stderr)
    # For LDS (tiled COARE + skin kernels: twenty evaluations per cell inside the cool skin's fixed-point chain, where the L1 round trip of
    # the interval-major table showed): not g itself but what the chain wants from it, the part of the absorbed fraction that depends on
    # the layer alone,  T(u) = 11 * 8e-4 u - (6.6e-5 / 8e-4) g(u),  zfr = MAX(c0 + T(delta / 8e-4), 0.01)  (CS_COARE mod_skin_coare.f90:85,
    # delta_skin_layer / the absorption profile of Fairall et al. 1996): a product, an FMA and a subtraction less per evaluation than
    # c0 + 11 delta - 0.0825 g.  Degree 7 on 56 intervals of [0, 7) (3 584 B, coefficient-major; the room the blended psi_h took until
    # round 4, when COARE's psi tables went to L1 with bit indexing): two FMAs less per evaluation than the degree-9 table on 16 intervals.
    def t_cs(u):
        return D(11.) * D(8.E-4) * u - (D(6.6E-5) / D(8.E-4)) * g_cs(u)
    rows, err = table(t_cs, mp.mpf(0), mp.mpf(7), 56)
    print("kCsTTab", err, file=sys.stderr)
    out.append(f"// T(u) = 8.8e-3 u - 0.0825 g(u) for LDS, degree 7, coefficient-major [k * 56 + i], 56 intervals on [0, 7): max |table - function| = {err:.2e}")
    out.append("constexpr int kCsGTabN = 56, kCsGTabDeg = 7;")
    out.append("constexpr double kCsGTabMax = 7.;")
    out.append("AB_TAB double kCsGTab[448] = {")
    flat = [rows[i][k] for k in range(8) for i in range(56)]
    for jj in range(0, len(flat), 4):
        out.append("    " + ", ".join(repr(v) for v in flat[jj:jj + 4]) + ("," if jj + 4 < len(flat) else "};"))
    global DEG
    # COARE's blended psi_m / psi_h once more, for LDS: the flux kernels WITHOUT the skin schemes have 5.4 KB of LDS left at five blocks
    # per CU, and their iteration is short enough that three table lookups through L1 (twelve 1 KB gathers per wave and iteration) keep the
    # texture addresser 72 % busy and the VALU waiting (77 % busy).  Degree 9 on 32 intervals of [0, 6.6875): 2 x 2 560 B, the error at
    # the rounding of the coefficients like the 107-interval degree-7 tables; coefficient-major [k * 32 + i]
    DEG = 9
    out.append("constexpr int kPsiCoareLdsN = 32, kPsiCoareLdsDeg = 9;")
    for which, name in (("m", "kPsiCoareLdsM"), ("h", "kPsiCoareLdsH")):
        rows, err = table(coare(which), mp.mpf(0), SMAX, 32)
        print(name, err, file=sys.stderr)
        out.append(f"// COARE unstable psi_{which} for LDS, degree 9, coefficient-major [k * 32 + i], 32 intervals on [0, 6.6875): max |table - function| = {err:.2e}")
        out.append(f"AB_TAB double {name}[320] = {{")
        flat = [rows[i][k] for k in range(10) for i in range(32)]
        for j in range(0, len(flat), 4):
            out.append("    " + ", ".join(repr(v) for v in flat[j:j + 4]) + ("," if j + 4 < len(flat) else "};"))
    # ANDREAS, stable side: psi_m (a cube root, a log, an atan and a division in closed form) is analytic in s = LOG(1 + zeta), its
    # singularities at Im s = +-pi; zeta <= 15 (the reference's cap): s <= LOG(16) = 2.7726, the table covers [0, 2.8) so that the capped
    # argument stays inside whatever the last bit of its logarithm
    out.append("constexpr int kPsiAndStabN = 10;")
    out.append("constexpr double kPsiAndStabSMax = 2.8;")
    rows, err = table(psim_andreas_stable, mp.mpf(0), mp.mpf("2.8"), 10)
    print("kPsiAndStabM", err, file=sys.stderr)
    out.append(f"// ANDREAS stable psi_m in s = LOG(1 + zeta) for LDS, degree 9, coefficient-major [k * 10 + i], 10 intervals on [0, 2.8): max |table - function| = {err:.2e}")
    out.append("AB_TAB double kPsiAndStabM[100] = {")
    flat = [rows[i][k] for k in range(10) for i in range(10)]
    for j in range(0, len(flat), 4):
        out.append("    " + ", ".join(repr(v) for v in flat[j:j + 4]) + ("," if j + 4 < len(flat) else "};"))
    DEG = 7
    # ---- tables indexed by the bits of their argument: degree 9 on 8 intervals per binade, interval-major rows of ten doubles (80 B: five
    # 16-byte reads on one address; from LDS in flux_kernel_cu, where a whole CU shares one copy, through L1 elsewhere)
    DEG = 9

    def bits_table(f, e0, e1, nsub, lo=None, hi=None, rel=False):
        rows, worst = [], mp.mpf(0)
        for e in range(e0, e1):
            for k in range(nsub):
                a, b = mp.mpf(2) ** e * (1 + mp.mpf(k) / nsub), mp.mpf(2) ** e * (1 + mp.mpf(k + 1) / nsub)
                c = local_fit(f, a, b)
                rows.append(c)
                if (lo is None or b > lo) and (hi is None or a < hi):
                    for j in range(17):
                        u = mp.mpf(-1) + mp.mpf(2) * j / 16
                        pp = mp.mpf(0)
                        for cc in reversed(c):
                            pp = pp * u + mp.mpf(cc)
                        x = (a + b) / 2 + (b - a) / 2 * u
                        worst = max(worst, abs(pp - f(x)) / (abs(f(x)) if rel else 1))
        return rows, float(worst)

    def emit10(name, rows, err, what):
        out.append(f"// {what}: {len(rows)} intervals x 10 coefficients (degree 9, interval-major), max error {err:.2e}")
        out.append(f"AB_TAB double {name}[{10 * len(rows)}] = {{")
        flat = [c for r in rows for c in r]
        for jj in range(0, len(flat), 4):
            out.append("    " + ", ".join(repr(v) for v in flat[jj:jj + 4]) + ("," if jj + 4 < len(flat) else "};"))

    out.append("constexpr int kLPsiCoareN = 80, kLWlAbsN = 72;    // LDS copies: 8 intervals per binade; y in [1, 1024), H in [2^-4, 2^5)")
    for which, name in (("m", "kLPsiCoareM"), ("h", "kLPsiCoareH")):
        fs = coare(which)
        rows, err = bits_table(lambda y, fs=fs: fs(mp.log(y)), 0, 10, 8)
        print(name, err, file=sys.stderr)
        emit10(name, rows, err, f"COARE unstable psi_{which} vs y = |1 - 15 zeta| for LDS")
    rows, err = bits_table(wl_abs, -4, 5, 8, lo=mp.mpf("0.1"), hi=mp.mpf(20), rel=True)
    print("kLWlAbs", err, file=sys.stderr)
    emit10("kLWlAbs", rows, err, "WL_COARE absorbed fraction vs depth H for LDS (relative error on [0.1, 20])")
    if "--psik-bits" in sys.argv:      # experiment of round 5 (profiles/r5_notes.md section 8): the Kansas pair indexed by the bits of y = |1 - 16 zeta|
        out.append("constexpr int kLPsikN = 80;")
        for nm, fsel in (("kLPsikM", psik_m), ("kLPsikH", psik_h)):
            rows, err = bits_table(lambda y, fsel=fsel: fsel(mp.log(y)), 0, 10, 8)
            print(nm, err, file=sys.stderr)
            emit10(nm, rows, err, "Kansas " + nm + " vs y = |1 - 16 zeta| (bit-indexed)")
    DEG = 7
    out.append("}  // namespace ab")
    print("\n".join(out))


if __name__ == "__main__":
    main()
