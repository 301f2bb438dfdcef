#!/usr/bin/env python
"""tools/gen_psitab.py — piecewise tables of the unstable profile functions in their log variables (ab_physics.hpp, kPsiTabM / kPsiTabH / kPsiTabC).

    F_M(s) = psi_m of Kansas / Paulson as a function of s = LOG(y), y = |1 - a zeta|   (tools/gen_poly.py section 6),  0 <= s < 6.6875
    F_H(s) = psi_h of Kansas / Paulson, same variable, same intervals (evaluated together with F_M at one table position)
    F_C(L) = COARE's convective psi as a function of L = LOG(y)                          (tools/gen_poly.py section 7),  0 <= L < 7.4453125
    E(T)   = e_sat_sclr of the reference (Goff 1957; mod_phymbl.f90:792-798) in Pa,                                          265 <= T < 312 K
The psi functions are analytic with the nearest singularities at +-2 pi i.  Each range is cut into equal intervals (28 / 28 / 24);
on each the function is replaced by its degree-7 Chebyshev interpolant (60-digit arithmetic), stored as monomial coefficients in
the local variable u in [-1, 1).  Half-width 0.12 / 0.13 against a distance of 2 pi: the interpolation error is below 1e-16
absolute; e_sat: 8.8e-17 relative (the rounding of the coefficients).
Layout: coefficient-major, tab[k * nint + i] = coefficient of u^k on interval i: the 64 lanes of a wave read one k at a time, lanes
on different intervals hit different LDS banks (8 B per interval, at most 32 intervals = the 64 banks), lanes on the same interval
the same address.  The fp32 tables (psi_m, psi_h, convective psi): 32 intervals, degree 3.
Prints the C initialisers and the measured error.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
DEG = 7
NINT = 32                 # fp32 tables
NINT_PSI, NINT_ESAT = 28, 24   # fp64 tables: intervals per function
RPI = mp.mpf(float("3.141592653589793"))      # rpi of the reference (mod_const.f90:39), as the double it is
# table ranges: just beyond LOG(1 + 16*50) = 6.68586 and LOG(1 + 34.15*50) = 7.44337 (the callers clamp zeta at -50: a clamped cell
# must stay inside), binary-friendly
SMAX = mp.mpf("6.6875")
LMAX = mp.mpf("7.4453125")


def D(x):
    """a literal of the reference as the double it is at run time (rt0 = 273.15 is 2.3e-14 below 273.15: seven ulp of e_sat)"""
    return mp.mpf(float(x))


def psik_m(s):
    return 2 * mp.log((1 + mp.exp(s / 4)) / 2) + mp.log((1 + mp.exp(s / 2)) / 2) - 2 * mp.atan(mp.exp(s / 4)) + RPI / 2


def psik_h(s):
    return 2 * mp.log((1 + mp.exp(s / 2)) / 2)


def psic_L(L):
    S3 = D(1.7320508)          # the reference's literals as doubles
    c = mp.exp(D(0.3333) * L)
    return mp.mpf("1.5") * mp.log((1 + c + c * c) / 3) - S3 * mp.atan((1 + 2 * c) / S3) + D(1.813799447)


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


T0 = D(273.15)


def e_sat(T):
    """e_sat_sclr of the reference (src/mod_phymbl.f90:792-798), Goff 1957, in Pa, with its literals as doubles"""
    z, x = T0 / T, T / T0
    A = (D(10.79574) * (1 - z) - D(5.028) * mp.log10(x) + D(1.50475) * mp.mpf(10) ** -4 * (1 - mp.power(10, D(-8.2969) * (x - 1)))
         + D(0.42873) * mp.mpf(10) ** -3 * (mp.power(10, D(4.76955) * (1 - z)) - 1) + D(0.78614))
    return 100 * mp.power(10, A)


def table(f, xmax, deg=None, nint=None, x0=0, rel=False):
    nint = NINT if nint is None else nint
    rows, worst = [], mp.mpf(0)
    for i in range(nint):
        a, b = x0 + (xmax - x0) * mp.mpf(i) / nint, x0 + (xmax - x0) * mp.mpf(i + 1) / nint
        c = local_fit(f, a, b, deg)
        rows.append(c)
        for k in range(41):
            u = mp.mpf(-1) + mp.mpf(2) * k / 40
            p = mp.mpf(0)
            for cc in reversed(c):
                p = p * u + mp.mpf(cc)
            fx = f((a + b) / 2 + (b - a) / 2 * u)
            worst = max(worst, abs(p - fx) / (abs(fx) if rel else 1))
    return rows, float(worst)


def emit(name, rows, err, what):
    nint = len(rows)
    print(f"// {what}: {nint} intervals, max |table - function| = {err:.2e}")
    print(f"AB_TAB double {name}[{(DEG + 1) * nint}] = {{")
    flat = [rows[i][k] for k in range(DEG + 1) for i in range(nint)]
    for j in range(0, len(flat), 4):
        print("    " + ", ".join(repr(v) for v in flat[j:j + 4]) + ("," if j + 4 < len(flat) else "};"))


def main():
    print(f"// SMAX = {float(SMAX)!r}, LMAX = {float(LMAX)!r}, degree {DEG}")
    rows, err = table(psik_m, SMAX, nint=NINT_PSI)
    emit("kPsiTabM", rows, err, "psi_m (Kansas / Paulson) in s = LOG(y)")
    rows, err = table(psik_h, SMAX, nint=NINT_PSI)
    emit("kPsiTabH", rows, err, "psi_h (Kansas / Paulson) in s = LOG(y)")
    rows, err = table(psic_L, LMAX, nint=NINT_PSI)
    emit("kPsiTabC", rows, err, "COARE convective psi in L = LOG(y)")
    rows, err = table(e_sat, mp.mpf(312), nint=NINT_ESAT, x0=mp.mpf(265), rel=True)
    emit("kEsatTab", rows, err, "e_sat(T) [Pa] on 265 K <= T < 312 K, RELATIVE error")
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
