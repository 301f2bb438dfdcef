#!/usr/bin/env python
"""tools/gen_psi_small.py — the ECMWF stability functions at TINY arguments (psi at z0/L, z0t/L, z0q/L, mod_blk_ecmwf.f90:276,290-292).

turb_ecmwf evaluates psi_m / psi_h at zeta = z0/L ~ 1e-6 .. 1e-4 three times per iteration.  There the closed forms
(mod_blk_ecmwf.f90:462-475, 519-531) are a sum of O(1) terms that cancel to O(zeta): two square roots, a log and an atan (with
its division) for a number a short polynomial gives exactly.  Each of the four functions (m/h x unstable/stable) is analytic on
its side of 0; this script fits them on |zeta| <= ZMAX in 60-digit arithmetic and prints the tables for ab_physics.hpp.
"""
import mpmath as mp

mp.mp.dps = 60
ZMAX = mp.mpf("1e-3")
PI_REF = mp.mpf(float("3.141592653589793"))     # rpi of the reference (mod_const.f90:39) as the double it is; psi_m adds 0.5*rpi
C = mp.mpf(5) / mp.mpf("0.35")


def psim_u(z):      # z < 0
    x2 = mp.sqrt(abs(1 - 16 * z)); x = mp.sqrt(x2)
    return mp.log((1 + x) ** 2 * (1 + x2) / 8) - 2 * mp.atan(x) + PI_REF / 2


def psih_u(z):
    x2 = mp.sqrt(abs(1 - 16 * z))
    return 2 * mp.log((1 + x2) / 2)


def psim_s(z):      # z >= 0
    return -mp.mpf(2) / 3 * (z - C) * mp.exp(-mp.mpf("0.35") * z) - z - mp.mpf(2) / 3 * C


def psih_s(z):
    return -mp.mpf(2) / 3 * (z - C) * mp.exp(-mp.mpf("0.35") * z) - abs(1 + mp.mpf(2) / 3 * z) ** mp.mpf("1.5") - mp.mpf(2) / 3 * C + 1


def cheb_fit(f, a, b, deg):
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    A = mp.matrix(n, n); y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [float(c[j]) for j in range(n)]


def max_abs_err(f, coef, a, b, npts=2001):
    worst = mp.mpf(0)
    for k in range(npts):
        x = a + (b - a) * mp.mpf(k) / (npts - 1)
        p = mp.mpf(0)
        for c in reversed(coef):
            p = p * x + mp.mpf(c)
        worst = max(worst, abs(p - f(x)))
    return float(worst)


# the variable is t = zeta / ZMAX in [-1, 0] or [0, 1] (coefficients of comparable size, Horner stable)
for name, f, lo, hi in (("kPsiMU", psim_u, -1, 0), ("kPsiHU", psih_u, -1, 0), ("kPsiMS", psim_s, 0, 1), ("kPsiHS", psih_s, 0, 1)):
    g = lambda t: f(t * ZMAX)
    for deg in (5, 6, 7, 8, 9, 10):
        c = cheb_fit(g, mp.mpf(lo), mp.mpf(hi), deg)
        e = max_abs_err(g, c, mp.mpf(lo), mp.mpf(hi))
        print(f"// {name} deg {deg}: max abs err {e:.2e}")
        if e < 1.2e-18:
            print(f"AB_TAB double {name}[fm::ab_pad4({deg + 1})] = {{" + ", ".join(repr(x) for x in c) + "};")
            break
