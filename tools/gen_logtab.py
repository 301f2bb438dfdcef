#!/usr/bin/env python
"""tools/gen_logtab.py — table and polynomial of the division-free fp64 log of aerobulk_amd/csrc/ab_fastmath.hpp (qlog).

    x = 2^n m,  m in [1/sqrt2, sqrt2);  i = top 6 bits of (high word of m - 0x3fe6a09e), 0..63;  r = m * invc[i] - 1,  |r| <= 0.00797
    log x = n ln2 + logc[i] + r + r^2 Q(r)
The 64 bins are equal steps of the HIGH WORD of m (2^14 units: 2^-7 wide below 1, 2^-6 above), so that the index is a shift and a
mask of the integer the range reduction already holds.  invc[i] = a double within 2048 ulp of 1/centre of bin i, picked so that
logc[i] = -log(invc[i]) (of that very double: the identity log m = log(m invc) - log(invc) is exact) is itself within 0.002 ulp of
a double (Gal's accurate tables); the bin that holds 1.0 (i = 37) has invc = 1 and logc = 0 exactly, so that arguments next to 1
lose nothing.
Prints the C initialisers and the measured error of the rounded polynomial (60-digit arithmetic).
"""
import struct

import mpmath as mp
import numpy as np

mp.mp.dps = 60
OFF = 0x3FE6A09E        # high word of the first bin's lower edge (just below 1/sqrt2)
NBIN, STEP = 64, 1 << 14


def hi2d(hi):
    return struct.unpack("<d", struct.pack("<Q", hi << 32))[0]


def table():
    """[(invc, logc, a, b)] for the 64 bins [a, b) of m"""
    rows = []
    for i in range(NBIN):
        a, b = hi2d(OFF + i * STEP), hi2d(OFF + (i + 1) * STEP)
        if a <= 1.0 < b:
            rows.append((1.0, 0.0, a, b))
            continue
        # Gal's accurate tables: among the doubles within 2048 ulp of 1/centre take the one whose -log is closest to a double, so
        # that logc carries no rounding error of its own (< 0.002 ulp) where it cancels against r next to the bin that holds 1
        c0 = float(1 / ((mp.mpf(a) + mp.mpf(b)) / 2))
        best = None
        for d in range(-2048, 2049):
            cand = float(c0 + d * np.spacing(c0))
            lg = -mp.log(mp.mpf(cand))
            err = abs(lg - mp.mpf(float(lg))) / mp.mpf(float(np.spacing(abs(float(lg)))))
            if best is None or err < best[0]:
                best = (err, cand, float(lg))
        rows.append((best[1], best[2], a, b))
    return rows


def cheb_fit(f, a, b, deg):
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [float(c[j]) for j in range(n)]


def main():
    rows = table()
    rmax = max(max(abs(mp.mpf(a) * invc - 1), abs(mp.mpf(b) * invc - 1)) for invc, _, a, b in rows) * (1 + mp.mpf(10) ** -12)
    q = lambda r: (mp.log1p(r) - r) / r ** 2 if abs(r) > mp.mpf(10) ** -20 else -mp.mpf(1) / 2 + r / 3
    for deg in (4, 5, 6):
        c = cheb_fit(q, -rmax, rmax, deg)
        worst = mp.mpf(0)
        for i in range(4001):
            r = -rmax + 2 * rmax * i / 4000
            if r == 0:
                continue
            p = mp.mpf(0)
            for cc in reversed(c):
                p = p * r + mp.mpf(cc)
            worst = max(worst, abs((r + r * r * p) / mp.log1p(r) - 1))
        print(f"// Q degree {deg}: max relative error of r + r^2 Q(r) vs log1p(r) on |r| <= {float(rmax):.5f}: {float(worst):.2e}")
        print("{ " + ", ".join(repr(x) for x in c) + " }")
    print(f"// invc, logc for the {NBIN} bins")
    txt = [f"{invc!r}, {logc!r}" for invc, logc, _, _ in rows]
    for i in range(0, len(txt), 2):
        print("    " + ", ".join(txt[i:i + 2]) + ",")


if __name__ == "__main__":
    main()
