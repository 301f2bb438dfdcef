"""GPU: accuracy of the engine's fp64 device math (ab_fastmath.hpp: v_rcp/v_rsq/v_log_f32 seeds + Newton +
near-minimax polynomials) measured on the MI355X in ulp against 80-bit long-double references."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OPS = {"div": 0, "rcp": 1, "sqrt": 2, "log": 3, "log10": 4, "exp": 5, "exp10": 6, "atan": 7, "cbrt": 8, "rcbrt": 9, "e_sat": 10, "pow": 11, "rqrt": 12, "e_sat_tab": 13}


def run(op, x, y=None):
    from aerobulk_amd import _lib
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    yp = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
    rc = lib.ab_test_math(OPS[op], x.ctypes.data_as(_lib.dp), yp.ctypes.data_as(_lib.dp) if yp is not None else C.cast(None, _lib.dp),
                          out.ctypes.data_as(_lib.dp), x.size)
    assert rc == 0, lib.ab_last_error()
    return out


def ulps(got, want):
    want = np.asarray(want, dtype=np.longdouble)
    _, e = np.frexp(want.astype(np.float64))
    ulp = np.ldexp(np.longdouble(1.0), e - 53)
    return float(np.max(np.abs(got.astype(np.longdouble) - want) / ulp))


RNG = np.random.default_rng(20261001)
N = 2_000_000
CASES = [  # (op, sampler, reference(longdouble), max ulp)
    ("log", lambda: np.exp(RNG.uniform(-60, 60, N)), np.log, 2.5),
    ("log", lambda: RNG.uniform(0.5, 2.0, N), np.log, 2.5),
    ("log", lambda: 1.0 + RNG.uniform(-8e-3, 8e-3, N), np.log, 2.5),        # the table bin around 1 (c = 1, log c = 0 exactly)
    ("log", lambda: 1.0 + RNG.uniform(-1e-9, 1e-9, N), np.log, 2.5),        # results of 1e-9 and below: relative accuracy kept
    ("log10", lambda: np.exp(RNG.uniform(-30, 15, N)), np.log10, 4.0),
    ("exp", lambda: RNG.uniform(-700, 700, N), np.exp, 1.5),
    ("exp", lambda: RNG.uniform(-3, 3, N), np.exp, 1.5),
    ("exp10", lambda: RNG.uniform(-300, 300, N), lambda v: np.power(np.longdouble(10), v), 2.0),
    ("exp10", lambda: RNG.uniform(-4, 4, N), lambda v: np.power(np.longdouble(10), v), 2.0),
    ("atan", lambda: RNG.uniform(-50, 50, N), np.arctan, 2.5),
    ("atan", lambda: np.exp(RNG.uniform(-18, 18, N)), np.arctan, 2.5),
    ("sqrt", lambda: np.exp(RNG.uniform(-400, 400, N)), np.sqrt, 0.51),
    ("rcp", lambda: np.exp(RNG.uniform(-400, 400, N)), lambda v: 1 / v, 0.51),
    ("cbrt", lambda: np.exp(RNG.uniform(-55, 55, N)), np.cbrt, 8.0),
    ("rcbrt", lambda: np.exp(RNG.uniform(-55, 55, N)), lambda v: 1 / np.cbrt(v), 4.0),
    ("rqrt", lambda: np.exp(RNG.uniform(-69, 69, N)), lambda v: 1 / np.sqrt(np.sqrt(v)), 4.0),   # 1e-30 .. 1e30: seed error grows with |log2 x|
    ("rqrt", lambda: np.exp(RNG.uniform(-35, 12, N)), lambda v: 1 / np.sqrt(np.sqrt(v)), 1.5),   # where 1 + x^0.75 differs from 1 at all
]


@pytest.mark.parametrize("op,sampler,ref,limit", CASES, ids=[f"{c[0]}_{i}" for i, c in enumerate(CASES)])
def test_unary_function_accuracy(op, sampler, ref, limit):
    x = sampler()
    got = run(op, x)
    assert np.all(np.isfinite(got))
    err = ulps(got, ref(x.astype(np.longdouble)))
    print(op, "max ulp error", err)
    assert err <= limit


def test_division_accuracy():
    a = np.exp(RNG.uniform(-150, 150, N)) * RNG.choice([-1.0, 1.0], N)
    b = np.exp(RNG.uniform(-150, 150, N)) * RNG.choice([-1.0, 1.0], N)
    got = run("div", a, b)
    err = ulps(got, a.astype(np.longdouble) / b.astype(np.longdouble))
    print("div max ulp error", err)
    assert err <= 0.51


def test_edge_values():
    assert run("exp", np.array([-1e4, 1e4, 0.0])).tolist() == [0.0, np.inf, 1.0]
    assert run("sqrt", np.array([0.0, 4.0])).tolist() == [0.0, 2.0]
    assert run("cbrt", np.array([0.0, 1e-40, 27.0]))[:2].tolist() == [0.0, 0.0]
    assert run("log", np.array([1.0]))[0] == 0.0
    assert run("atan", np.array([0.0, -0.0]))[0] == 0.0


def test_e_sat_matches_oracle(oracle):
    """Goff (1957) saturation vapour pressure, the most-called transcendental block (mod_phymbl.f90:777-800)."""
    t = RNG.uniform(170.0, 330.0, 200000)
    got = run("e_sat", t)
    want = np.array([oracle.lib().abo_e_sat(v) for v in t])
    rel = np.abs(got - want) / want
    print("e_sat max rel err vs oracle", rel.max(), "at T =", t[np.argmax(rel)])
    # both sides carry ~1e-15 of rounding in the exponent (A ~ 0.3..2, times ln 10)
    assert rel.max() < 6e-15
    # the same through the piecewise LDS table of the tiled flux kernels (265 K <= T < 312 K; the formula path elsewhere): the table
    # itself is good to 1e-16, what is left is the oracle's own rounding
    tt = np.concatenate([t, [265.0, np.nextafter(312.0, 0.0), 312.0, np.nextafter(265.0, 0.0)], RNG.uniform(265.0, 312.0, 200000)])
    got = run("e_sat_tab", tt)
    want = np.array([oracle.lib().abo_e_sat(v) for v in tt])
    rel = np.abs(got - want) / want
    print("e_sat (table) max rel err vs oracle", rel.max(), "at T =", tt[np.argmax(rel)])
    assert rel.max() < 6e-15
