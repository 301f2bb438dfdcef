"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): fluxes within 1e-10 relative of the Fortran reference in fp64, evaluated with the metric of
oracle/parity.py (1e-6 floor of SURVEY §8d; cells beyond it must be within 8 ulp of backward error on every input, and few).
The oracle is itself pinned to the compiled reference (tests/test_oracle_vs_ref.py, tests/golden).
"""
import numpy as np
import pytest

from conftest import assert_hot_parity, sensitivity

pytestmark = pytest.mark.gpu

TOL = 1e-10
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
OUT = ("ql", "qh", "tau_x", "tau_y", "evap")
CAP = {"ql": "QL", "qh": "QH", "tau_x": "Tau_x", "tau_y": "Tau_y", "evap": "Evap", "t_s": "T_s"}


def _run_both(oracle, algo, skin, niter, zt, ni=360, nj=180, nt=1, hum="sh", fields=None):
    import aerobulk_amd as ab
    f = fields or oracle.synth_fields(ni, nj)
    n = f["sst"].size
    osess = oracle.OracleSession(algo, n, nt, skin, hum)
    results = []
    _run_both.last_fields = f
    _run_both.sens = sensitivity(oracle, algo, skin, zt, 10., niter, f, nt=nt, hum_type=hum)
    with ab.Session(algo, n, 1, nt, skin) as s:
        s.set_humidity(hum)
        for jt in range(1, nt + 1):
            rs, rl = (f["rad_sw"], f["rad_lw"]) if skin else (None, None)
            ref = osess.compute(jt, zt, 10., niter, *[f[k] for k in IN6], rad_sw=rs, rad_lw=rl)
            got = s.compute(jt, zt, 10., *[f[k] for k in IN6], Niter=niter, rad_sw=rs, rad_lw=rl)
            got = {k: got[CAP[k]] for k in OUT + (("t_s",) if skin else ())}
            results.append((got, ref))
    return results


def _dump_outliers(got, ref, keys, nmax=3):
    from oracle.parity import rel_err
    f = _run_both.last_fields
    for k in keys:
        e = rel_err(got[k], ref[k])
        for c in np.argsort(e)[::-1][:nmax]:
            if e[c] > TOL:
                print(f"OUTLIER {k} cell {c} rel {e[c]:.3e}: " + " ".join(f"{n}={f[n][c]!r}" for n in f)
                      + " | " + " ".join(f"{q}: got {got[q][c]!r} ref {ref[q][c]!r}" for q in keys))


CASES = [(a, sk) for a in ("coare3p0", "coare3p6", "ncar", "ecmwf", "andreas")
         for sk in ((False, True) if a in ("coare3p0", "coare3p6", "ecmwf") else (False,))]


@pytest.mark.parametrize("algo,skin", CASES)
@pytest.mark.parametrize("niter,zt", [(5, 2.), (8, 10.)])
def test_parity_single_record(oracle, algo, skin, niter, zt):
    (got, ref), = _run_both(oracle, algo, skin, niter, zt)
    keys = OUT + (("t_s",) if skin else ())
    _dump_outliers(got, ref, keys)
    assert_hot_parity(got, ref, keys, sens=_run_both.sens, label=f"{algo} skin={skin} n={niter} zt={zt}")


@pytest.mark.parametrize("algo", ["coare3p0", "coare3p6", "ecmwf"])
def test_parity_warm_layer_carry_over(oracle, algo):
    """nt=3 identical records: warm-layer state must persist between jt calls (SURVEY §8c pin)."""
    res = _run_both(oracle, algo, True, 8, 2., ni=128, nj=96, nt=3)
    for jt, (got, ref) in enumerate(res, 1):
        assert_hot_parity(got, ref, OUT + ("t_s",), sens=_run_both.sens, jt=jt, label=f"{algo} carry-over jt={jt}")
    # outputs must actually drift between records (state is being carried)
    assert np.max(np.abs(res[0][0]["t_s"] - res[2][0]["t_s"])) > 1e-6


@pytest.mark.parametrize("niter", [6, 64, 65])
def test_parity_warm_layer_live_iterations(oracle, niter):
    """WL_COARE commits its state in the iterations jit with MOD(nb_iter, jit) == 0 (mod_blk_coare3p6.f90:370, mod_skin_coare.f90:239-248).
    The kernel takes them from a 64-bit mask made on the host (FluxArgs::wl_live) and divides only beyond 64 iterations: 6 has four
    divisors (a mask with several bits), 64 is the mask's last bit, 65 the fallback.  Two records, so that the committed state is read back."""
    res = _run_both(oracle, "coare3p6", True, niter, 2., ni=96, nj=64, nt=2)
    for jt, (got, ref) in enumerate(res, 1):
        assert_hot_parity(got, ref, OUT + ("t_s",), sens=_run_both.sens, jt=jt, label=f"coare3p6 nb_iter={niter} jt={jt}")


@pytest.mark.parametrize("hum", ["rh", "dp"])
def test_parity_humidity_types(oracle, hum):
    f = oracle.synth_fields(200, 100)
    if hum == "rh":
        qs = np.array([oracle.lib().abo_q_sat(t, p) for t, p in zip(f["t_zt"], f["slp"])])
        f["hum_zt"] = 100. * np.clip(f["hum_zt"] / qs, 0., 1.)
    else:
        f["hum_zt"] = f["t_zt"] - 3.
    (got, ref), = _run_both(oracle, "coare3p6", False, 5, 2., hum=hum, fields=f)
    assert_hot_parity(got, ref, OUT, sens=_run_both.sens, label=f"humidity {hum}")
