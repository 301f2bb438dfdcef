#!/usr/bin/env python
"""tools/gen_logtab.py — table and polynomial of the division-free fp64 log of aerobulk_amd/csrc/ab_fastmath.hpp (qlog).

    x = 2^n m,  m in [1/sqrt2, sqrt2);  k = rint(64 m) in [45, 91];  r = m * invc[k] - 1,  |r| <= 0.0112
    log x = n ln2 + logc[k] + r + r^2 Q(r)
invc[k] = double(64/k);  logc[k] = -log(invc[k]) for the ROUNDED invc (so that the identity is exact; k = 64: 1 and 0 exactly).
Prints the C initialisers and the measured error of the rounded polynomial (60-digit arithmetic).
"""
import mpmath as mp

mp.mp.dps = 60
K0, K1 = 45, 91


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
    rmax = mp.mpf(1) / 128 / (mp.mpf(K0) / 64) * (1 + mp.mpf(10) ** -12)
    q = lambda r: (mp.log1p(r) - r) / r ** 2 if abs(r) > mp.mpf(10) ** -20 else -mp.mpf(1) / 2 + r / 3
    for deg in (5, 6, 7):
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
    print(f"// invc, logc for k = {K0}..{K1}")
    rows = []
    for k in range(K0, K1 + 1):
        invc = float(mp.mpf(64) / k)
        logc = float(-mp.log(mp.mpf(invc)))
        rows.append(f"{invc!r}, {logc!r}")
    for i in range(0, len(rows), 2):
        print("    " + ", ".join(rows[i:i + 2]) + ",")


if __name__ == "__main__":
    main()
