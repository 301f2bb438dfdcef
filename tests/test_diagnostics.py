"""TURB_* diagnostics: transfer coefficients, adjusted theta/q, bulk wind, neutral coefficients, z0, u*, L,
UN10, skin increments.  Golden data: the reference's TURB_* routines called directly with all OPTIONAL outputs
(oracle/ref_turb_driver.f90 -> tests/golden/diag_*.npz)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity

IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
MAN = json.load(open(os.path.join(GOLDEN, "diag_manifest.json")))
NAMES = ("Cd", "Ch", "Ce", "t_zu", "q_zu", "Ubzu", "CdN", "ChN", "CeN", "z0", "u_star", "L", "UN10", "dT_cs", "dT_wl", "Hz_wl")


def _inputs(n):
    f = dict(np.load(os.path.join(GOLDEN, "sweep_inputs.npz")))
    return {k: np.ascontiguousarray(v[:n]) for k, v in f.items()}


def _keys(case):
    return NAMES if case["skin"] else NAMES[:13]


def _finite_ref(ref, got, keys):
    """L = 1/(1/L) is +-inf / huge on neutral cells: compare 1/L there instead of L."""
    r, g = dict(ref), dict(got)
    r["L"], g["L"] = 1.0 / ref["L"], 1.0 / got["L"]
    return r, g


@pytest.mark.parametrize("case", MAN, ids=lambda c: c["name"])
def test_oracle_diagnostics_match_reference(oracle, case):
    f = _inputs(case["n"])
    ref = dict(np.load(os.path.join(GOLDEN, case["name"] + ".npz")))
    o = oracle.OracleSession(case["algo"], case["n"], 1, case["skin"]).compute(
        1, case["zt"], case["zu"], case["niter"], *[f[k] for k in IN6], rad_sw=f["rad_sw"] if case["skin"] else None,
        rad_lw=f["rad_lw"] if case["skin"] else None, diag=True)
    r, g = _finite_ref(ref, o, _keys(case))
    assert_parity(g, r, _keys(case), tol=1e-12, abs_frac=1e-13, label=case["name"])


def _well_conditioned(oracle, case, f, keys):
    """Ch = (u*/U) t*/dt and Ce = (u*/U) q*/dq divide by air-sea differences that the skin scheme has just updated: on a
    cell where q_s ~ q_zu (or T_s ~ theta_zu) to 6+ digits the coefficient answers a ONE-ulp change of the inputs with a
    >1e-11 relative change, so no 1e-10 comparison between two different fp64 evaluation orders is meaningful there.
    Such cells are found with the oracle itself (one-ulp moves of each input in turn) and left out; at most 1 %."""
    def run(ff):
        return oracle.OracleSession(case["algo"], case["n"], 1, case["skin"]).compute(
            1, case["zt"], case["zu"], case["niter"], *[ff[k] for k in IN6], rad_sw=ff["rad_sw"] if case["skin"] else None,
            rad_lw=ff["rad_lw"] if case["skin"] else None, diag=True)
    a = run(f)
    ok = np.ones(case["n"], bool)
    for name in IN6:                # every input in turn, both directions
        for sgn in (np.inf, -np.inf):
            b = run(dict(f, **{name: np.nextafter(f[name], sgn)}))
            for k in ("Ch", "Ce", "L"):     # L = 1/(1/L): near neutrality the Obukhov length answers a one-ulp move the same way
                ok &= np.abs(a[k] - b[k]) <= 1e-11 * np.abs(a[k])
    assert (~ok).sum() <= 10, (~ok).sum()
    return ok


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN, ids=lambda c: c["name"])
def test_hip_diagnostics_match_reference(oracle, case):
    import aerobulk_amd as ab
    f = _inputs(case["n"])
    ref = dict(np.load(os.path.join(GOLDEN, case["name"] + ".npz")))
    with ab.Session(case["algo"], case["n"], 1, 1, case["skin"]) as s:
        d = s.set_diagnostics(_keys(case))
        plain = None
        out = s.compute(1, case["zt"], case["zu"], *[f[k] for k in IN6], Niter=case["niter"],
                        rad_sw=f["rad_sw"] if case["skin"] else None, rad_lw=f["rad_lw"] if case["skin"] else None)
        got = {k: v.copy() for k, v in d.items()}
        s.set_diagnostics(None)      # back to the lean kernel: fluxes must not depend on the instantiation
        plain = s.compute(1, case["zt"], case["zu"], *[f[k] for k in IN6], Niter=case["niter"],
                          rad_sw=f["rad_sw"] if case["skin"] else None, rad_lw=f["rad_lw"] if case["skin"] else None)
    for k in plain:
        np.testing.assert_array_equal(out[k], plain[k], err_msg=k)
    r, g = _finite_ref(ref, got, _keys(case))
    ok = _well_conditioned(oracle, case, f, _keys(case))
    assert_parity({k: v[ok] for k, v in g.items()}, {k: v[ok] for k, v in r.items()}, _keys(case), abs_frac=1e-11, label=case["name"])
    assert_parity(g, r, _keys(case), tol=1e-8, abs_frac=1e-9, label=case["name"] + " (all cells)")


@pytest.mark.gpu
def test_hip_diagnostics_device_arrays(oracle):
    import torch
    import aerobulk_amd as ab
    ni, nj = 300, 40
    f = oracle.synth_fields(ni, nj)
    names = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
    dev = [torch.from_numpy(f[k]).cuda() for k in names]
    o = oracle.OracleSession("coare3p6", ni * nj, 1, True).compute(1, 2.0, 10.0, 5, *[f[k] for k in names], rad_sw=f["rad_sw"],
                                                                    rad_lw=f["rad_lw"], diag=True)
    with ab.Session("coare3p6", ni, nj, 1, True) as s:
        d = s.set_diagnostics(("Cd", "u_star", "dT_cs", "Hz_wl"), device="cuda")
        s.compute(1, 2.0, 10.0, *dev, Niter=5, rad_sw=torch.from_numpy(f["rad_sw"]).cuda(), rad_lw=torch.from_numpy(f["rad_lw"]).cuda())
        got = {k: v.cpu().numpy() for k, v in d.items()}
    assert_parity(got, o, ("Cd", "u_star", "dT_cs", "Hz_wl"), label="device diagnostics")


def test_oracle_diagnostics_close_to_readme_toy_table(oracle):
    """README.md:188-204 toy table (older revision, 4-5 digits): sanity on C_D, C_E, C_H, z0, u*, L, UN10, C_D_N."""
    rows = {"coare3p0": (1.1954, 1.3345, 1.3345, 4.40936E-05, 0.17578, -20.383, 5.4192, 1.0521),
            "coare3p6": (1.0775, 1.3729, 1.3729, 2.19285E-05, 0.16672, -16.919, 5.4311, 0.94234),
            "ncar": (1.2038, 1.3618, 1.2776, 4.49880E-05, 0.17348, -20.494, 5.3396, 1.0555),
            "ecmwf": (1.2862, 1.3143, 1.2635, 6.98835E-05, 0.18192, -24.029, 5.3992, 1.1353),
            "andreas": (1.0167, 1.1565, 1.1103, 1.56119E-05, 0.1594, -18.558, 5.3289, 0.8950)}
    a = lambda v: np.array([v], dtype=np.float64)
    for algo, (cd, ce, ch, z0, us, L, un10, cdn) in rows.items():
        o = oracle.OracleSession(algo, 1).compute(1, 2.0, 10.0, 20, a(295.15), a(293.15), a(0.012), a(5.0), a(0.0), a(101000.0), diag=True)
        got = (o["Cd"][0] * 1e3, o["Ce"][0] * 1e3, o["Ch"][0] * 1e3, o["z0"][0], o["u_star"][0], o["L"][0], o["UN10"][0], o["CdN"][0] * 1e3)
        np.testing.assert_allclose(got, (cd, ce, ch, z0, us, L, un10, cdn), rtol=6e-3, err_msg=algo)
