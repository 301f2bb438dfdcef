#!/usr/bin/env python
"""tools/gen_poly.py — near-minimax polynomial coefficients for the device math in aerobulk_amd/csrc/ab_math.hpp.

Chebyshev-node interpolation in 50-digit arithmetic (mpmath), coefficients rounded to double, maximum error of the
ROUNDED polynomial re-measured on a dense grid.  Output: C initialiser lists to paste into ab_math.hpp.
"""
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


def max_err(f, coef, a, b, npts=4001, rel=True):
    worst = mp.mpf(0)
    for k in range(npts):
        x = a + (b - a) * mp.mpf(k) / (npts - 1)
        p = mp.mpf(0)
        for c in reversed(coef):
            p = p * x + mp.mpf(c)
        fx = f(x)
        e = abs(p - fx)
        if rel and fx != 0:
            e = e / abs(fx)
        worst = max(worst, e)
    return float(worst)


def show(name, coef):
    print(f"// {name}")
    print("{ " + ", ".join(f"{c!r}" for c in coef) + " }")


ln2 = mp.log(2)
# 1) exp(r) = 1 + r + r^2 * P(r),  |r| <= ln2/2
fe = lambda r: (mp.exp(r) - 1 - r) / r ** 2 if abs(r) > mp.mpf(10) ** -15 else mp.mpf(1) / 2 + r / 6 + r * r / 24
for deg in (7, 8, 9, 10):
    c = cheb_fit(fe, -ln2 / 2, ln2 / 2, deg)
    # error of the full exp
    err = max_err(lambda r: mp.exp(r), [1.0, 1.0] + c, -ln2 / 2, ln2 / 2)
    print("exp  deg", deg, "rel err of full exp", err)
    show(f"EXP_P{deg}: exp(r) = 1 + r + r^2*P(r)", c)

# 2) log(m) = 2 atanh(s), s=(m-1)/(m+1), |s| <= 0.17157288 ; 2 atanh(s) = 2s + s^3 * P(s^2)
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
umax = smax ** 2
fl = lambda u: (2 * mp.atanh(mp.sqrt(u)) / mp.sqrt(u) - 2) / u if u != 0 else mp.mpf(2) / 3
for deg in (5, 6, 7):
    c = cheb_fit(fl, mp.mpf(0), umax, deg)
    # relative error of 2atanh(s)/s: (2 + u P(u)) vs exact
    err = max_err(lambda u: 2 * mp.atanh(mp.sqrt(u)) / mp.sqrt(u) if u != 0 else mp.mpf(2), [2.0] + c, mp.mpf(0), umax)
    print("log  deg", deg, "rel err of 2atanh(s)/s", err)
    show(f"LOG_P{deg}: 2atanh(s) = 2s + s^3*P(s^2)", c)

# 3) atan(t) = t + t^3 * P(t^2), |t| <= tan(pi/8)
tmax = mp.tan(mp.pi / 8)
fa = lambda u: (mp.atan(mp.sqrt(u)) / mp.sqrt(u) - 1) / u if u != 0 else -mp.mpf(1) / 3
for deg in (8, 9, 10):
    c = cheb_fit(fa, mp.mpf(0), tmax ** 2, deg)
    err = max_err(lambda u: mp.atan(mp.sqrt(u)) / mp.sqrt(u) if u != 0 else mp.mpf(1), [1.0] + c, mp.mpf(0), tmax ** 2)
    print("atan deg", deg, "rel err of atan(t)/t", err)
    show(f"ATAN_P{deg}: atan(t) = t + t^3*P(t^2)", c)

print("ln2_hi/lo etc.")
for name, v in (("ln2", ln2), ("log2e", 1 / ln2), ("ln10", mp.log(10)), ("log2_10", mp.log(10) / ln2), ("log10_2", ln2 / mp.log(10)),
                ("log10e", 1 / mp.log(10)), ("pi_2", mp.pi / 2), ("pi_4", mp.pi / 4)):
    hi = float(v)
    lo = float(v - mp.mpf(hi))
    print(f"{name}: hi={hi!r} lo={lo!r}")
# split ln2 with trailing zero bits so that k*ln2_hi is exact for |k| < 2^11
import struct
def trunc_bits(x, nbits):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    b &= ~((1 << nbits) - 1)
    return struct.unpack("<d", struct.pack("<Q", b))[0]
for name, v in (("ln2", ln2), ("log10_2", ln2 / mp.log(10))):
    hi = trunc_bits(float(v), 21)
    lo = float(v - mp.mpf(hi))
    print(f"{name} (hi with 21 trailing zero bits): hi={hi!r} lo={lo!r}")

# 4) Goff (1957) exponent A(T) of e_sat = 100*10^A(T) (reference: src/mod_phymbl.f90:792-798) on T in [265, 312] K,
#    x = (T - 288.5)/23.5 in [-1, 1]: polynomial surrogate of the SAME analytic function (3 exp10 + log10 + division -> 14 FMAs)
def D(x):
    """a literal of the reference as the double it is at run time (rt0 = 273.15 is 2.3e-14 below 273.15, which is seven ulp of e_sat:
    the round-1 fit used the decimal values and sat 1.3e-15 below the reference, profiles/r2_notes.md)"""
    return mp.mpf(float(x))


T0 = D(273.15)


def goff_A(T):
    z, x = T0 / T, T / T0
    return (D(10.79574) * (1 - z) - D(5.028) * mp.log10(x)
            + D(1.50475) * mp.mpf(10) ** -4 * (1 - mp.power(10, D(-8.2969) * (x - 1)))
            + D(0.42873) * mp.mpf(10) ** -3 * (mp.power(10, D(4.76955) * (1 - z)) - 1) + D(0.78614))


for deg in (13, 14, 15):
    c = cheb_fit(lambda x: goff_A(mp.mpf("288.5") + mp.mpf("23.5") * x), mp.mpf(-1), mp.mpf(1), deg)
    err = max_err(lambda x: goff_A(mp.mpf("288.5") + mp.mpf("23.5") * x), c, mp.mpf(-1), mp.mpf(1), rel=False)
    print("goff deg", deg, "abs err of A(T)", err)
    show(f"GOFF_A{deg}: A(T), x=(T-288.5)/23.5", c)

# 5) COARE convective psi (reference: src/mod_common_coare.f90:240-243,330-333), with the reference's own truncated
#    literals 1.7320508 and 1.813799447:
#      psi_c(c) = 1.5 ln((1+c+c^2)/3) - 1.7320508 atan((1+2c)/1.7320508) + 1.813799447,  c = y^.3333 >= 1
#    = 3 ln c + G(w), w = 1/c in (0,1]:  G(w) = 1.5 ln((w^2+w+1)/3) - S (pi/2 - atan(S w/(w+2))) + 1.813799447
S3 = D(1.7320508)          # the reference's literals as doubles (0.15 ulp of psi_c at most; D() above)
G = lambda w: mp.mpf("1.5") * mp.log((w * w + w + 1) / 3) - S3 * (mp.pi / 2 - mp.atan(S3 * w / (w + 2))) + D(1.813799447)
for deg in (9, 18, 20, 22, 24):     # 9: the fp32 path (3.4e-8 absolute)
    c = cheb_fit(lambda x: G((x + 1) / 2), mp.mpf(-1), mp.mpf(1), deg)
    err = max_err(lambda x: G((x + 1) / 2), c, mp.mpf(-1), mp.mpf(1), rel=False)
    print("psic G deg", deg, "abs err", err)
    if deg in (9, 20, 22):
        show(f"PSIC_G{deg}: G(w), x = 2w-1", c)

# 6) Round 2 — the Kansas/Paulson unstable profile functions as polynomials in s = ln(y), y = |1 - a zeta| >= 1
#    (a = 15: mod_common_coare.f90:235-238,326-328; a = 16: mod_blk_ecmwf.f90:462-467,519-523, mod_blk_ncar.f90:350-362,
#    mod_blk_andreas.f90:351-358,402-408), x = y**.25 = exp(s/4):
#      psi_m = 2 ln((1+x)/2) + ln((1+x^2)/2) - 2 atan(x) + 0.5 rpi ,   psi_h = 2 ln((1+x^2)/2)
#    Analytic in s with the nearest singularities at s = +-2 pi i: one log + 22 FMAs each instead of two square roots, a log
#    and an atan with its division.  Fitted on s in [0, SMAX], SMAX = ln(1 + 16*50) (zeta >= -50 in every caller's loop);
#    variable t = 2 s / SMAX - 1.
RPI = mp.mpf(float("3.141592653589793"))      # rpi of the reference (mod_const.f90:39), as the double it is
SMAX = mp.log(801)
psik_m = lambda s: 2 * mp.log((1 + mp.exp(s / 4)) / 2) + mp.log((1 + mp.exp(s / 2)) / 2) - 2 * mp.atan(mp.exp(s / 4)) + RPI / 2
psik_h = lambda s: 2 * mp.log((1 + mp.exp(s / 2)) / 2)
print("SMAX =", repr(float(SMAX)))
for name, f in (("PSIK_M", psik_m), ("PSIK_H", psik_h)):
    for deg in (20, 22, 24):
        c = cheb_fit(lambda t: f((t + 1) / 2 * SMAX), mp.mpf(-1), mp.mpf(1), deg)
        err = max_err(lambda t: f((t + 1) / 2 * SMAX), c, mp.mpf(-1), mp.mpf(1), rel=False)
        print(name, "deg", deg, "abs err", err)
        if deg == 22:
            show(f"{name}{deg}: t = 2 s / SMAX - 1", c)

# 7) Round 2 — COARE convective psi (section 5) as a polynomial in ITS OWN log L = ln(y), y = |1 - a zeta| (a = 10.15 / 34.15):
#    psi_c = 1.5 ln((1+c+c^2)/3) - 1.7320508 atan((1+2c)/1.7320508) + 1.813799447, c = exp(.3333 L): analytic in L, nearest
#    singularities at L = +-2 pi i.  One log + 28 FMAs instead of a log, an exponential and the degree-20 G(w).
#    L in [0, LMAX], LMAX = ln(1 + 34.15*50); t = 2 L / LMAX - 1.
LMAX = mp.log(1 + mp.mpf("34.15") * 50)
def psic_L(L):
    c = mp.exp(D(0.3333) * L)
    return mp.mpf("1.5") * mp.log((1 + c + c * c) / 3) - S3 * mp.atan((1 + 2 * c) / S3) + D(1.813799447)
print("LMAX =", repr(float(LMAX)))
for deg in (24, 26, 28, 30):
    c = cheb_fit(lambda t: psic_L((t + 1) / 2 * LMAX), mp.mpf(-1), mp.mpf(1), deg)
    err = max_err(lambda t: psic_L((t + 1) / 2 * LMAX), c, mp.mpf(-1), mp.mpf(1), rel=False)
    print("PSIC_L deg", deg, "abs err", err)
    if deg in (24, 26):
        show(f"PSIC_L{deg}: t = 2 L / LMAX - 1", c)
