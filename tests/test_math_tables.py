"""The LDS tables of the fp64 log / exp (aerobulk_amd/csrc/ab_fastmath.hpp) are what tools/gen_logtab.py / gen_exptab.py define:
invc[i] within 2048 ulp of 1/centre of bin i for the 64 equal steps of m's high word from 0x3fe6a09e, logc[i] = -log(invc[i]) of that
very value and within 0.002 ulp of exact, the bin that holds 1.0 has exactly (1, 0); T[j] = 2^(j/64)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

mp = pytest.importorskip("mpmath")
HDR = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_fastmath.hpp")).read()


def _table(name):
    m = re.search(rf"AB_TAB double {name}\[[^\]]*\] = \{{(.*?)\}};", HDR, re.S)
    assert m, name
    return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])


def test_log_table_is_exactly_what_the_identity_needs():
    mp.mp.dps = 50
    import struct
    t = _table("kLogTab").reshape(-1, 2)
    assert t.shape == (64, 2)
    edge = lambda i: struct.unpack("<d", struct.pack("<Q", (0x3FE6A09E + i * (1 << 14)) << 32))[0]
    assert edge(0) < 0.5 ** 0.5 and edge(64) == 2 * edge(0)      # the bins cover one binade starting just below 1/sqrt2
    worst = 0.0
    for i, (invc, logc) in enumerate(t):
        a, b = edge(i), edge(i + 1)
        if a <= 1.0 < b:
            assert (invc, logc) == (1.0, 0.0)
        else:
            c0 = float(1 / ((mp.mpf(a) + mp.mpf(b)) / 2))
            assert abs(invc - c0) <= 2048 * np.spacing(c0)
        lg = -mp.log(mp.mpf(invc))
        assert logc == float(lg)                               # of THAT reciprocal: log m = log(m invc) - log(invc) exactly
        assert abs(lg - mp.mpf(logc)) <= mp.mpf(0.002) * mp.mpf(float(np.spacing(abs(logc)))) or logc == 0.0   # accurate table
        worst = max(worst, abs(a * invc - 1), abs(b * invc - 1))
    assert worst < 0.00797                                     # the range the polynomial Q was fitted on


def test_exp_table_and_reduction_constants():
    mp.mp.dps = 50
    t = _table("kExpTab")
    assert t.size == 64
    assert all(t[j] == float(mp.mpf(2) ** (mp.mpf(j) / 64)) for j in range(64))
    # the 30-bit heads of ln2/64 and log10(2)/64 times any |k| < 2^22 are exact in double; head + tail reproduce the constant
    for head, tail, val in ((0.010830424696905538, -6.563929801064195e-13, mp.log(2) / 64),
                            (0.004703593680460472, 1.7892345153159123e-12, mp.log10(2) / 64)):
        assert f"{head!r}" in HDR and f"{tail!r}" in HDR
        m, _ = np.frexp(head)
        assert (m * 2.0 ** 30) == int(m * 2.0 ** 30)
        assert abs(mp.mpf(head) + mp.mpf(tail) - val) < mp.mpf(10) ** -28


def test_lds_constant_table_repeats_the_coefficients_it_stands_for():
    """fm::kConstTab (constants fetched from LDS instead of being copied into VGPRs) holds c[N-2] of the polynomial tables (c[N-1]
    where the top coefficient is alone in its chunk of eight) and four literals of the cube / fourth root refinements."""
    phys = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_physics.hpp")).read()

    def tab(src, name):
        m = re.search(rf"AB_TAB double {name}\[[^\]]*\] = \{{(.*?)\}};", src, re.S)
        assert m, name
        return [float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]

    names = re.search(r"enum ab_const \{(.*?)\};", HDR, re.S).group(1)
    names = [n.split("=")[0].strip() for n in names.replace("\n", " ").split(",")]
    names = [n for n in names if n and n not in ("kC_N", "kC_None")]
    c = tab(HDR, "kConstTab")
    assert len(c) == len(names) == 12
    def macro(name):     # coefficient lists kept as macros (instruction literals)
        m = re.search(rf"#define {name} ([^/\n]*)", HDR)
        assert m, name
        return [float(x) for x in m.group(1).split(",")]

    assert len(macro("AB_LOGQ")) == 6 and len(macro("AB_EXPQ")) == 4
    assert "horner_lit6<kC_LogQ4>(r, AB_LOGQ)" in HDR and "horner_lit4<kC_ExpQ2>(r, AB_EXPQ)" in HDR
    want = {"kC_LogQ4": macro("AB_LOGQ")[4], "kC_ExpQ2": macro("AB_EXPQ")[2], "kC_AtanP9": tab(HDR, "kAtanP")[9],
            "kC_PsikM21": tab(phys, "kPsikM")[21], "kC_PsikH21": tab(phys, "kPsikH")[21], "kC_PsicL24": tab(phys, "kPsicL")[24],
            "kC_PsicG19": tab(phys, "kPsicG")[19], "kC_Goff13": tab(phys, "kGoffA")[13], "kC_Third": 0.3333333333333333,
            "kC_Quarter": 0.25, "kC_TwoNinths": 0.2222222222222222, "kC_5_32": 0.15625}
    for n, v in zip(names, c):
        assert v == want[n], n
    # (N, entry) as used: the entry is c[N-2], or c[N-1] when (N-1) % 8 == 0
    for n, (N, t) in {"kC_AtanP9": (11, "kAtanP")}.items():
        assert f"horner_coefs<{N}, {n}>({t}," in HDR
    for n, (N, t) in {"kC_PsikM21": (23, "kPsikM"), "kC_PsikH21": (23, "kPsikH"), "kC_PsicL24": (25, "kPsicL"), "kC_PsicG19": (21, "kPsicG")}.items():
        assert f"horner_tab<{N}, fm::{n}>({t}," in phys
        idx = int(re.search(r"(\d+)$", n).group(1))
        assert idx == (N - 1 if (N - 1) % 8 == 0 else N - 2)
    assert "horner_coefs<15, fm::kC_Goff13>(kGoffA," in phys


def test_psi_lds_tables_are_what_the_generator_defines_and_accurate():
    """kPsiTabM / kPsiTabH / kPsiTabC / kEsatTab (ab_physics.hpp): psi_m, psi_h (Kansas) in s = LOG(y) on [0, 6.6875) and COARE's convective psi in
    L = LOG(y) on [0, 7.4453125), 28 intervals each; e_sat(T) on [265, 312) K, 24 intervals; degree 7, coefficient-major —
    tools/gen_psitab.py; evaluated as the kernels do."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_psitab", os.path.join(ROOT, "tools", "gen_psitab.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    phys = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_physics.hpp")).read()
    rng = np.random.default_rng(5)
    for name, f, x0, x1, nint, rel in (("kPsiTabM", g.psik_m, 0.0, 6.6875, 28, False), ("kPsiTabH", g.psik_h, 0.0, 6.6875, 28, False),
                                       ("kPsiTabC", g.psic_L, 0.0, 7.4453125, 28, False), ("kEsatTab", g.e_sat, 265.0, 312.0, 24, True)):
        m = re.search(rf"AB_TAB double {name}\[{8 * nint}\] = \{{(.*?)\}};", phys, re.S)
        assert m, name
        t = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]).reshape(8, nint)
        rows, err = g.table(f, g.mp.mpf(x1), nint=nint, x0=g.mp.mpf(x0), rel=rel)
        assert err < 4.2e-16      # (psi_h reaches 6.6: 4.1e-16 is half an ulp there)
        np.testing.assert_array_equal(t, np.array(rows).T)
        worst = 0.0
        for x in np.concatenate([rng.uniform(x0, x1, 300), [x0, np.nextafter(x1, 0)]]):
            xn = (x - x0) * (nint / (x1 - x0))
            i = int(np.floor(xn))
            u = (xn - np.floor(xn)) * 2.0 - 1.0
            p = t[7, i]
            for k in range(6, -1, -1):
                p = p * u + t[k, i]
            fx = float(f(g.mp.mpf(float(x))))
            worst = max(worst, abs(p - fx) / (abs(fx) if rel else 1.0))
        assert worst < 3e-15, (name, worst)      # psi values up to 6.6: a few ulp; e_sat: relative
    assert f"constexpr int kTabNint[4] = {{28, 28, 24, 28}}, kTabOff[3] = {{0, 224, 448}}, kTabTotal = 640;" in phys


def test_fp32_psi_tables():
    """kPsiTab32: psi_m, psi_h (Kansas / Paulson) and the convective psi, 32 intervals x degree 3, float entries (tools/gen_psitab.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_psitab", os.path.join(ROOT, "tools", "gen_psitab.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    phys = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_physics.hpp")).read()
    m = re.search(r"AB_TAB float kPsiTab32\[384\] = \{(.*?)\};", phys, re.S)
    assert m
    t = np.array([np.float32(x.strip().rstrip("f")) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]).reshape(3, 4, 32)
    rng = np.random.default_rng(6)
    for w, (f, xmax) in enumerate(((g.psik_m, 6.6875), (g.psik_h, 6.6875), (g.psic_L, 7.4453125))):
        rows, err = g.table(f, g.mp.mpf(xmax), 3)
        assert err < 2e-8
        np.testing.assert_array_equal(t[w], np.array(rows, dtype=np.float32).T)
        worst = 0.0
        for x in rng.uniform(0, xmax, 200):
            x32 = np.float32(x) * np.float32(32.0 / xmax)
            i = int(np.floor(x32))
            u = np.float32((x32 - np.floor(x32)) * np.float32(2.0) - np.float32(1.0))
            p = t[w, 3, i]
            for k in range(2, -1, -1):
                p = np.float32(p * u + t[w, k, i])
            worst = max(worst, abs(float(p) - float(f(g.mp.mpf(float(np.float32(x)))))))
        assert worst < 3e-6, (w, worst)     # fp32: values up to 6.6 (ulp 4.8e-7), the argument itself rounded to float


def test_lkb_rows_in_the_log_domain_are_the_references_table():
    """kLkbLog (ab_physics.hpp): {ln a_t, b_t - 1, ln a_q, b_q - 1} of the eight rows of z0tq_LKB (mod_phymbl.f90:1658-1667), the literals
    taken as the doubles they are; the bin edges and clamps of z0tq_lkb_log are the logarithms of the reference's."""
    import mpmath as mp
    mp.mp.dps = 40
    phys = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_physics.hpp")).read()
    m = re.search(r"AB_TAB double kLkbLog\[32\] = \{(.*?)\};", phys, re.S)
    got = [float(x) for x in m.group(1).replace("\n", " ").split(",")]
    rows = [(0.177, 0., 0.292, 0.), (1.376, 0.929, 1.808, 0.826), (1.026, -0.599, 1.393, -0.528), (1.625, -1.018, 1.956, -0.870),
            (4.661, -1.475, 4.994, -1.297), (34.904, -2.067, 30.709, -1.845), (1667.19, -2.907, 1448.68, -2.682), (5.88e5, -3.935, 2.98e5, -3.616)]
    want = []
    for at, bt, aq, bq in rows:
        want += [float(mp.log(mp.mpf(at))), float(mp.mpf(bt) - 1), float(mp.log(mp.mpf(aq))), float(mp.mpf(bq) - 1)]
    assert got == want
    body = phys[phys.index("void z0tq_lkb_log("):phys.index("lnz0q = lq;")]
    for x in (0.11, 0.825, 3.0, 10.0, 30.0, 100., 300., 1000., 1e-9, 0.05):
        assert repr(float(mp.log(mp.mpf(x)))) in body, x
    assert repr(float(mp.log(mp.mpf(0.0025)))) in phys           # ln z0_sea_max in turb_andreas


def test_table_positions_stay_inside_their_tables_at_the_last_admissible_argument():
    """Every piecewise table is indexed by (int)(x * scale) behind a test x < bound.  Scales that are not powers of two are rounded: the
    position of the LAST admissible argument (the predecessor of the bound) must still truncate to the last interval — the product is
    monotone in x, so this one value decides for all.  Bounds and scales as the kernels write them (ab_physics.hpp)."""
    phys = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_physics.hpp")).read()
    gt = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_gtables.hpp")).read()
    n_lds = int(re.search(r"kPsiCoareLdsN = (\d+)", gt).group(1))
    n_csg = int(re.search(r"kCsGTabN = (\d+)", gt).group(1))
    csg_max = float(re.search(r"kCsGTabMax = ([\d.]+);", gt).group(1))
    assert "constexpr double kGPsiSMax = 6.6875;" in gt and "kGPsiCoareN = 107" in gt and "kGCsGN = 64" in gt
    assert "kPsiTabSMax = 6.6875, kPsiTabLMax = 7.4453125, kEsatTabT0 = 265., kEsatTabT1 = 312." in phys
    assert "if (sl <= R(6.68586094706836))" in phys                      # the Kansas LDS pair's own bound, inside [0, 6.6875)

    def last64(bound, num, den, n, off=0.0):
        x = np.nextafter(np.float64(bound), np.float64(0))
        xn = (x - np.float64(off)) * (np.float64(num) / np.float64(den))
        return int(xn) == n - 1 and xn < n

    def last32(bound, num, den, n):
        x = np.nextafter(np.float32(bound), np.float32(0))
        xn = np.float32(x * np.float32(np.float64(num) / np.float64(den)))
        return int(xn) == n - 1 and xn < n

    assert last64(312., 24., 47., 24, off=265.)          # e_sat
    assert last64(6.6875, n_lds, 6.6875, n_lds)          # COARE's blended psi in LDS
    assert last64(6.6875, 107, 6.6875, 107)              # ... and through L1
    assert last64(np.nextafter(6.68586094706836, 7), 28, 6.6875, 28)   # Kansas pair (<= bound: the bound itself is admissible)
    assert last64(7.4453125, 28, 7.4453125, 28)          # convective psi (math_test_kernel)
    assert (n_csg, csg_max) == (56, 7.0) and last64(csg_max, n_csg, csg_max, n_csg)        # cool skin T(u) = 8.8e-3 u - 0.0825 g(u), LDS
    assert last64(8., 64, 8., 64)                        # ... L1
    assert last32(6.6875, 32, 6.6875, 32) and last32(7.4453125, 32, 7.4453125, 32)   # fp32 tables


def _bits_table(name, n):
    gt = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_gtables.hpp")).read()
    assert "constexpr int kLPsiCoareN = 80, kLWlAbsN = 72;" in gt
    m = re.search(rf"AB_TAB double {name}\[{10 * n}\] = \{{(.*?)\}};", gt, re.S)
    return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]).reshape(n, 10)


def _bits_eval(tab, x, e0):
    """as bits_pos8<E0>() and lds10_at() of ab_physics.hpp: interval from the exponent and the top three mantissa bits, local variable
    from the other 49, Horner from the top coefficient"""
    import struct
    bits = struct.unpack("<Q", struct.pack("<d", x))[0]
    hi, lo = bits >> 32, bits & 0xFFFFFFFF
    idx = (hi >> 17) - ((1023 + e0) << 3)
    assert 0 <= idx < tab.shape[0], (x, idx)
    v = struct.unpack("<d", struct.pack("<Q", (((hi & 0x1FFFF) | 0x3FF00000) << 32) | lo))[0]
    u = v * 16.0 - 17.0
    assert -1.0 <= u < 1.0
    p = 0.0
    for c in tab[idx][::-1]:
        p = p * u + c
    return p


def test_warm_layer_absorption_table_indexed_by_the_bits_of_the_depth():
    """kLWlAbs (ab_gtables.hpp): WL_COARE's absorbed fraction of the solar flux (mod_skin_coare.f90:167-168, 205-207) against its closed
    form in 40-digit arithmetic, with the index and the local variable formed exactly as wl_absorb() forms them from the bits of H:
    every depth the scheme can produce (H clamped to [0.1, 20]) lands inside the table, on the right interval, within 1e-15."""
    mp.mp.dps = 40
    tab = _bits_table("kLWlAbs", 72)

    def closed(H):
        H = mp.mpf(H)
        c = [mp.mpf(0.28 * 0.014), mp.mpf(0.27 * 0.357), mp.mpf(0.45 * 12.82)]
        a = [mp.mpf(0.014), mp.mpf(0.357), mp.mpf(12.82)]
        return 1 - sum(ci * (-mp.expm1(-H / ai)) for ci, ai in zip(c, a)) / H

    rng = np.random.default_rng(5)
    hs = np.concatenate([[0.1, np.nextafter(0.1, 1), 20.0, np.nextafter(20.0, 0), 0.125, 1.0, 16.0, np.nextafter(16.0, 0)],
                         np.exp(rng.uniform(np.log(0.1), np.log(20.0), 400))])
    worst = 0.0
    for H in hs:
        ref = closed(H)
        worst = max(worst, float(abs(mp.mpf(_bits_eval(tab, H, -4)) - ref) / abs(ref)))
    assert worst < 1e-15, worst


def test_coare_psi_tables_indexed_by_the_bits_of_their_argument():
    """kLPsiCoareM / H (ab_gtables.hpp): COARE's blended unstable psi_m / psi_h as functions of y = |1 - 15 zeta|, the interval taken from
    the exponent and the top three mantissa bits of y exactly as psi_coare<kPsiBits / kPsiBitsLds> takes it: against the reference's formulas
    (mod_common_coare.f90:235-252, :326-342, tools/gen_gtab.py) in 40-digit arithmetic, every zeta the iteration can produce."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_gtab
    mp.mp.dps = 40
    rng = np.random.default_rng(11)
    zetas = -np.concatenate([[0.0, 1e-12, 1e-6, 50.0, 49.999, 1.0 / 15.0, 2.0 / 15.0], 10.0 ** rng.uniform(-8, np.log10(50.0), 300)])
    for which, name in (("m", "kLPsiCoareM"), ("h", "kLPsiCoareH")):
        tab = _bits_table(name, 80)
        f = gen_gtab.coare(which)
        worst = 0.0
        for z in zetas:
            y = abs(1.0 - 15.0 * z)
            worst = max(worst, float(abs(mp.mpf(_bits_eval(tab, y, 0)) - f(mp.log(mp.mpf(y))))))
        assert worst < 2e-15, (which, worst)      # values up to 6.6
