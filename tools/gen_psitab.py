#!/usr/bin/env python
"""tools/gen_psitab.py — piecewise tables of two unstable profile functions in their log variables (ab_physics.hpp, kPsiTabM / kPsiTabC).

    F_M(s) = psi_m of Kansas / Paulson as a function of s = LOG(y), y = |1 - a zeta|   (tools/gen_poly.py section 6),  0 <= s < 6.6875
    F_C(L) = COARE's convective psi as a function of L = LOG(y)                          (tools/gen_poly.py section 7),  0 <= L < 7.4453125
Both are analytic with the nearest singularities at +-2 pi i.  The range is cut into 32 equal intervals; on each the function is
replaced by its degree-7 Chebyshev interpolant (60-digit arithmetic), stored as monomial coefficients in the local variable
u in [-1, 1).  Half-width 0.105 / 0.116 against a distance of 2 pi: the interpolation error is below 1e-16 absolute.
Layout: coefficient-major, tab[k * 32 + i] = coefficient of u^k on interval i: the 64 lanes of a wave read one k at a time, lanes
on different intervals hit different LDS banks (32 intervals x 8 B = the 64 banks), lanes on the same interval the same address.
Prints the C initialisers and the measured error.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
NINT, DEG = 32, 7
RPI = mp.mpf(float("3.141592653589793"))      # rpi of the reference (mod_const.f90:39), as the double it is
S3 = mp.mpf("1.7320508")
# table ranges: just beyond LOG(1 + 16*50) = 6.68586 and LOG(1 + 34.15*50) = 7.44337 (the callers clamp zeta at -50: a clamped cell
# must stay inside), binary-friendly
SMAX = mp.mpf("6.6875")
LMAX = mp.mpf("7.4453125")


def psik_m(s):
    return 2 * mp.log((1 + mp.exp(s / 4)) / 2) + mp.log((1 + mp.exp(s / 2)) / 2) - 2 * mp.atan(mp.exp(s / 4)) + RPI / 2


def psik_h(s):
    return 2 * mp.log((1 + mp.exp(s / 2)) / 2)


def psic_L(L):
    c = mp.exp(mp.mpf("0.3333") * L)
    return mp.mpf("1.5") * mp.log((1 + c + c * c) / 3) - S3 * mp.atan((1 + 2 * c) / S3) + mp.mpf("1.813799447")


def local_fit(f, a, b, deg=None):
    """monomial coefficients in u = (2 x - a - b)/(b - a) of the degree-deg interpolant at the Chebyshev nodes of [a, b]"""
    n = (DEG if deg is None else deg) + 1
    us = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, u in enumerate(us):
        for j in range(n):
            A[i, j] = u ** j
        y[i] = f((a + b) / 2 + (b - a) / 2 * u)
    c = mp.lu_solve(A, y)
    return [float(c[j]) for j in range(n)]


def table(f, xmax, deg=None):
    rows, worst = [], mp.mpf(0)
    for i in range(NINT):
        a, b = xmax * i / NINT, xmax * (i + 1) / NINT
        c = local_fit(f, a, b, deg)
        rows.append(c)
        for k in range(41):
            u = mp.mpf(-1) + mp.mpf(2) * k / 40
            p = mp.mpf(0)
            for cc in reversed(c):
                p = p * u + mp.mpf(cc)
            worst = max(worst, abs(p - f((a + b) / 2 + (b - a) / 2 * u)))
    return rows, float(worst)


def emit(name, rows, err, what):
    print(f"// {what}: max |table - function| = {err:.2e}")
    print(f"AB_TAB double {name}[{(DEG + 1) * NINT}] = {{")
    flat = [rows[i][k] for k in range(DEG + 1) for i in range(NINT)]
    for j in range(0, len(flat), 4):
        print("    " + ", ".join(repr(v) for v in flat[j:j + 4]) + ("," if j + 4 < len(flat) else "};"))


def main():
    print(f"// SMAX = {float(SMAX)!r}, LMAX = {float(LMAX)!r}, {NINT} intervals, degree {DEG}")
    rows, err = table(psik_m, SMAX)
    emit("kPsiTabM", rows, err, "psi_m (Kansas / Paulson) in s = LOG(y)")
    rows, err = table(psic_L, LMAX)
    emit("kPsiTabC", rows, err, "COARE convective psi in L = LOG(y)")
    # fp32 kernels: degree 3 on the same intervals (5e-9 of interpolation error), float entries: psi_m, psi_h, convective psi
    flat, errs = [], []
    for f, xmax in ((psik_m, SMAX), (psik_h, SMAX), (psic_L, LMAX)):
        rows, err = table(f, xmax, 3)
        errs.append(err)
        flat += [rows[i][k] for k in range(4) for i in range(NINT)]
    print("// fp32: psi_m, psi_h (Kansas / Paulson), convective psi; 32 intervals, degree 3, [function][coefficient][interval]; "
          "max |table - function| = " + " / ".join(f"{e:.1e}" for e in errs))
    print("AB_TAB float kPsiTab32[384] = {")
    for j in range(0, len(flat), 6):
        print("    " + ", ".join(f"{np.float32(v):.9g}f" for v in flat[j:j + 6]) + ("," if j + 6 < len(flat) else "};"))


if __name__ == "__main__":
    main()
