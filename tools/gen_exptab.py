#!/usr/bin/env python
"""tools/gen_exptab.py — table and polynomial of the table-driven fp64 exp of aerobulk_amd/csrc/ab_fastmath.hpp (qexp, qexp10).

    x = (N k' + j) ln2/N + r,  |r| <= ln2/(2N):   exp x = 2^k' T[j] (1 + r + r^2 P(r)),   T[j] = 2^(j/N)
Prints T, the split of ln2/N and log10(2)/N into a short head (products with k exact) and a tail, and P with its measured error.
"""
import sys

import mpmath as mp

mp.mp.dps = 60


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


def head(x, bits):
    """x rounded to `bits` significant bits (as a double)"""
    m, e = mp.frexp(x)
    return float(mp.ldexp(mp.nint(mp.ldexp(m, bits)), e - bits))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rmax = mp.log(2) / (2 * N) * (1 + mp.mpf(10) ** -9)
    f = lambda r: (mp.expm1(r) - r) / r ** 2 if abs(r) > mp.mpf(10) ** -20 else mp.mpf(1) / 2 + r / 6
    for deg in (2, 3, 4, 5):
        c = cheb_fit(f, -rmax, rmax, deg)
        worst = mp.mpf(0)
        for i in range(2001):
            r = -rmax + 2 * rmax * i / 2000
            p = mp.mpf(0)
            for cc in reversed(c):
                p = p * r + mp.mpf(cc)
            worst = max(worst, abs((1 + r + r * r * p) / mp.exp(r) - 1))
        print(f"// N = {N}: P degree {deg}: max relative error of 1 + r + r^2 P(r) vs exp(r): {float(worst):.2e}")
        print("{ " + ", ".join(repr(x) for x in c) + " }")
    for name, v in (("ln2/N", mp.log(2) / N), ("log10(2)/N", mp.log10(2) / N)):
        h = head(v, 30)
        print(f"// {name}: head (30 bits) {h!r}, tail {float(v - mp.mpf(h))!r};  N/{name.split('/')[0]} = {float(1 / v)!r}")
    print(f"// T[j] = 2^(j/{N})")
    t = [repr(float(mp.mpf(2) ** (mp.mpf(j) / N))) for j in range(N)]
    for i in range(0, N, 4):
        print("    " + ", ".join(t[i:i + 4]) + ",")


if __name__ == "__main__":
    main()
