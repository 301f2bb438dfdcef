"""Adversarial fields (GPU): EVERY cell tuned so that the air-sea difference the bulk formula multiplies — q_zu - q_s or theta_zu - T_s
at the end of the iteration — lands on a chosen tiny value (1e-13 ... 1e-7 in q, 1e-10 ... 1e-4 K, both signs, on either side of the
floors of the TURB_* routines).  A random field meets such a cell once in 1e7 — it took 100 fuzz seeds to find the four that exposed
q_s being rounded in one place and not in another (profiles/r2_fuzz_wide.txt, entry 6); here all cells are of that kind, C_e reaches
1e6, a third of the fluxes are beyond any forward bar — and every value must still be what the reference computes for inputs within
8 ulp (the backward clause of oracle/parity.py, no budget on how many values need it).  The tuning uses the oracle only."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_fuzz import _fields

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT = (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s"))
TARGETS = np.array([1e-7, 1e-8, 3e-9, 1e-9, 3e-10, 1e-10, 1e-11, 1e-12, 3e-13, 1e-13])


def adversarial_fields(po, algo, skin, zt, zu, niter, n, seed):
    """(fields, which): which[i] = 0 humidity difference tuned, 1 temperature difference tuned, 2 both"""
    L = po.lib()
    qs_fn = L.abo_q_sat
    qs_fn.restype, qs_fn.argtypes = C.c_double, [C.c_double, C.c_double]
    f = _fields(seed, n)
    keep = np.hypot(f["u_zu"], f["v_zu"]) < 25.0
    f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
    n = f["sst"].size
    r = np.random.default_rng(seed + 1)
    which = r.integers(0, 3, n)
    dq_t = r.choice(TARGETS, n) * r.choice([-1.0, 1.0], n)
    dt_t = r.choice(TARGETS * 1e3, n) * r.choice([-1.0, 1.0], n)           # kelvin: 1e-10 ... 1e-4 (floors 1e-9, 1e-6)
    f0 = {k: f[k].copy() for k in ("hum_zt", "t_zt")}
    prev = None
    for it in range(10):                                                   # secant iteration per cell on hum_zt and t_zt (oracle only)
        s = po.OracleSession(algo, n, 1, skin)
        o = s.compute(1, zt, zu, niter, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None, diag=True)
        ts = o["t_s"] if skin else f["sst"]
        q_s = 0.98 * np.array([qs_fn(float(a), float(b)) for a, b in zip(ts, f["slp"])])
        rq, rt = (o["q_zu"] - q_s) - dq_t, (o["t_zu"] - ts) - dt_t         # residuals
        cur = (f["hum_zt"].copy(), f["t_zt"].copy(), rq, rt)
        sq, st = np.ones(n), np.ones(n)
        if prev is not None:
            with np.errstate(all="ignore"):
                a = (rq - prev[2]) / (cur[0] - prev[0])
                b = (rt - prev[3]) / (cur[1] - prev[1])
            sq = np.where(np.isfinite(a) & (np.abs(a) > 0.05) & (np.abs(a) < 20.0), a, 1.0)
            st = np.where(np.isfinite(b) & (np.abs(b) > 0.05) & (np.abs(b) < 20.0), b, 1.0)
        prev = cur
        mq, mt = which != 1, which != 0
        f["hum_zt"] = np.where(mq, np.maximum(f["hum_zt"] - rq / sq, 1e-5), f["hum_zt"])
        f["t_zt"] = np.where(mt, f["t_zt"] - rt / st, f["t_zt"])
    # a cell whose secant iteration ran away (no root within reach: the clamps of a very unstable calm cell) keeps its original inputs
    lost = ~np.isfinite(f["t_zt"]) | ~np.isfinite(f["hum_zt"]) | (np.abs(f["t_zt"] - f["sst"]) > 14.0) | (f["hum_zt"] <= 1e-5) | (f["hum_zt"] > 0.04)
    for k in f0:
        f[k] = np.where(lost, f0[k], f[k])
    return f, which



@pytest.mark.gpu
@pytest.mark.parametrize("algo,skin,zt,zu,niter", [("coare3p6", True, 18.0, 25.0, 5), ("coare3p0", True, 3.5, 17.0, 4), ("coare3p6", False, 2.0, 10.0, 8),
                                                    ("andreas", False, 8.0, 12.0, 7), ("ecmwf", True, 2.0, 10.0, 6), ("ncar", False, 2.0, 10.0, 5)])
def test_vanishing_air_sea_differences_everywhere(oracle, algo, skin, zt, zu, niter):
    import aerobulk_amd as ab
    from oracle import parity
    f, which = adversarial_fields(oracle, algo, skin, zt, zu, niter, 5000, 900)
    m = f["sst"].size
    nt = 2 if skin else 1
    rad = dict(rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
    ins = [f[k] for k in IN8[:6]]
    osess = oracle.OracleSession(algo, m, nt, skin)
    sens = parity.OracleSensitivity(oracle, algo, skin, zt, zu, niter, {k: f[k] for k in (IN8 if skin else IN8[:6])}, nt=nt)
    keys = OUT if skin else OUT[:5]
    flagged = 0
    with ab.Session(algo, m, 1, nt, skin) as s:
        for jt in range(1, nt + 1):
            ref = osess.compute(jt, zt, zu, niter, *ins, **rad)
            got = s.compute(jt, zt, zu, *ins, Niter=niter, **rad)
            rep = parity.parity_report({kr: got[k] for k, kr in keys}, ref, [kr for _, kr in keys], sens=sens, jt=jt)
            for k, r in rep.items():
                assert r["n_nonfinite"] == 0 and r["n_unexplained"] == 0, (algo, jt, k, r)
                assert r.get("backward_ratio_max", 0.0) <= 1.0, (algo, jt, k, r)
                flagged += r["n_gt_tol"]
    assert flagged > 200, flagged          # the field is what it is meant to be: hundreds of values no forward bar can hold


def test_the_tuner_reaches_its_targets(oracle):
    """CPU: the fields are what the GPU test says they are (oracle only)."""
    algo, skin, zt, zu, niter = "coare3p6", True, 18.0, 25.0, 5
    f, which = adversarial_fields(oracle, algo, skin, zt, zu, niter, 1500, 900)
    n = f["sst"].size
    o = oracle.OracleSession(algo, n, 1, skin).compute(1, zt, zu, niter, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], diag=True)
    L = oracle.lib()
    L.abo_q_sat.restype, L.abo_q_sat.argtypes = C.c_double, [C.c_double, C.c_double]
    q_s = 0.98 * np.array([L.abo_q_sat(float(a), float(b)) for a, b in zip(o["t_s"], f["slp"])])
    dq, dt = np.abs(o["q_zu"] - q_s), np.abs(o["t_zu"] - o["t_s"])
    assert np.median(dq[which != 1]) < 2e-9 and np.median(dt[which != 0]) < 2e-6, (np.median(dq[which != 1]), np.median(dt[which != 0]))
    assert np.all(np.isfinite(o["ql"])) and np.all(np.isfinite(o["qh"]))
    assert o["Ce"].max() > 10.0              # the transfer coefficient of a vanishing difference: q* of the pass before over 1e-9 ... 1e-13
