"""The grid of the reference's own sweep driver src/tests/test_cx_vs_wind.f90 — 1201 winds x 75 air-sea virtual-temperature differences
x 7 relative humidities, nb_iter = 20 (:19-26,77,96-120) — through TURB_* of the engine: its densest exercise of the Charnock ramps, the
LKB bins and NCAR's 33 m/s threshold.  Golden: the UNMODIFIED reference's TURB_* on the same 630 525 cells, reduced the way the driver
reduces them (mean over the humidities in its order) for all winds and 19 of the 75 differences (tools/gen_cx_vs_wind_golden.py ->
tests/golden/cx_vs_wind.npz).
  CPU: the C oracle reproduces the stored means (the oracle is what the other parity tests lean on).
  GPU: ab_session_turb on the whole grid in one call per algorithm; and the reference's UNCHANGED driver, compiled against this
       repository's modules (oracle/_ref/dropin/test_cx_vs_wind.x: one-cell TURB_* / Ri_bulk / visc_air calls through the Fortran
       host), writes the same numbers into its .dat files (soak tier: two million tiny launches)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_parity

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_cx_vs_wind_golden as gen  # noqa: E402  (the grid's definition lives with the generator; nothing here needs the reference)

FIELDS = ("Cd", "Ch", "Ce", "z0", "us")


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "cx_vs_wind.npz"))
    return {k: z[k] for k in z.files}


def test_grid_is_the_drivers(gold):
    w = gold["winds"]
    assert w.size == 1201 and w[0] == 0. and abs(w[-1] - 40.) < 1e-9          # (the quarter steps below 5 m/s keep the table short of wind_max = 50)
    d = np.diff(w)
    assert np.allclose(d[w[:-1] < 5. - 1e-9], 50. / 1200 / 4) and np.allclose(d[w[:-1] >= 30.], 50. / 600)
    assert gold["t_dvt"].size == 75 and gold["t_dvt"][38] == 0. and 38 in gold["keep"] and gold["vrh"].tolist() == [70., 75., 80., 85., 90., 95., 100.]
    assert gold["coare3p6_Cd"].shape == (19, 1201)
    # the couples solve the driver's fixed point: T_v(theta, q(RH, theta)) = T_v(SST, q_sat(SST)) + dT
    assert np.array_equal(gen.winds(), w)


def _cells(gold, rows):
    """The driver's (dT, wind, humidity) cells of the temperature differences `rows`, humidity fastest."""
    cpl = gold["couples"][rows]
    nd, nw, nh = len(rows), gold["winds"].size, gold["vrh"].size
    n = nd * nw * nh
    th = np.ascontiguousarray(np.broadcast_to(cpl[:, 0][:, None, :], (nd, nw, nh)).reshape(n))
    q = np.ascontiguousarray(np.broadcast_to(cpl[:, 1][:, None, :], (nd, nw, nh)).reshape(n))
    ww = np.ascontiguousarray(np.broadcast_to(gold["winds"][None, :, None], (nd, nw, nh)).reshape(n))
    return th, q, ww, (nd, nw, nh)


def _rh_mean(x, shape):
    x = x.reshape(shape)
    m = np.zeros(shape[:2])
    for jh in range(shape[2]):
        m = m + x[:, :, jh] / shape[2]
    return m


def test_oracle_reproduces_the_sweep(oracle, gold):
    rows = list(gold["keep"][[0, 9, 18]])            # dT = -12, 0, +12 of the stored ones
    th, q, ww, shape = _cells(gold, rows)
    n = th.size
    for algo in gen.ALGOS:
        one = np.ones(n)
        rec = np.stack([float(gold["sstk"]) * one, th, float(gold["qsat_sst"]) * one, q, ww, 0. * one, 0. * one, 101000. * one])
        o = oracle.oracle_turb_series(algo, 0, 0, 20, 2.0, 10.0, np.zeros(n), np.array([43200.]), rec[None])[0]
        sel = [list(gold["keep"]).index(r) for r in rows]
        got = {k: _rh_mean(o[gen.FIELDS[k]], shape) for k in FIELDS}
        ref = {k: gold[f"{algo}_{k}"][sel] for k in FIELDS}
        assert_parity(got, ref, FIELDS, tol=1e-12, abs_frac=1e-13, label=f"oracle cx_vs_wind {algo}")


@pytest.mark.gpu
@pytest.mark.parametrize("algo", gen.ALGOS)
def test_engine_reproduces_the_sweep(gold, algo):
    import aerobulk_amd as ab
    rows = list(range(75))                              # the whole grid, as the driver runs it: 630 525 cells in one launch
    th, q, ww, shape = _cells(gold, rows)
    n = th.size
    with ab.Session(algo, n, 1, 1, False) as s:
        d = s.set_diagnostics(["z0", "u_star"])
        T_s, q_s = np.full(n, float(gold["sstk"])), np.full(n, float(gold["qsat_sst"]))
        o = s.turb(1, 2.0, 10.0, T_s, th, q_s, q, ww, nb_iter=20)
        cell = {"Cd": np.array(o["Cd"]), "Ch": np.array(o["Ch"]), "Ce": np.array(o["Ce"]), "z0": np.array(d["z0"]), "us": np.array(d["u_star"])}
    keep = gold["keep"]
    got = {k: _rh_mean(v, shape)[keep] for k, v in cell.items()}
    ref = {k: gold[f"{algo}_{k}"] for k in FIELDS}
    assert_parity(got, ref, FIELDS, label=f"cx_vs_wind {algo}")


@pytest.mark.gpu
def test_the_references_unchanged_sweep_driver_runs_on_the_engine(gold, tmp_path):
    """src/tests/test_cx_vs_wind.f90, not a character changed, against this repository's Fortran modules (oracle/_ref/dropin, built by
    aerobulk_amd/build.py).  NB the driver branches on a CHARACTER variable it never sets (`stab`, :48,:136): built with amdflang — against
    the reference's own library just the same — it takes the `GOTO 201` and skips its sweep, so what can be compared is what it prints
    before: the virtual SST, which comes from mod_phymbl's q_sat (:131-134, here a one-cell launch), and its closing line.  Should a
    compiler make it run the sweep, every TURB_* / Ri_bulk / visc_air call of the triple loop is a one-cell launch and its cd / ch / ce /
    us files (f16.8) are held against the golden means."""
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin", "test_cx_vs_wind.x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin not built (needs the reference tree and amdflang at build time)")
    os.makedirs(tmp_path / "dat")
    r = subprocess.run([exe, "ncar", "22"], cwd=tmp_path, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Number of itterations used:" in r.stdout and "20" in r.stdout.split("Number of itterations used:")[1]
    # Virtual Sea Surface temperature = 24.891393 with the reference's own modules (REAL(.,4) of sst (1 + rctv0 0.98 q_sat(sst, Patm)) - rt0)
    v = float(r.stdout.split("Virtual Sea Surface temperature =")[1].split()[0])
    sst_v = float(gold["sstk"]) * (1. + (461.495 / 287.05 - 1.) * float(gold["qsat_sst"])) - 273.15
    assert abs(v - sst_v) < 2e-6 and abs(v - 24.891393) < 2e-6, (v, sst_v)
    files = [f for f in os.listdir(tmp_path / "dat") if f.endswith("_sst_22_ncar.dat")]
    for ik, jdt in enumerate(gold["keep"]):
        want = int(100 * float(gold["t_dvt"][jdt]))
        for kind, key, scale in (("cd", "Cd", 1000.), ("ch", "Ch", 1000.), ("ce", "Ce", 1000.), ("us", "us", 1.)):
            match = [f for f in files if f.startswith(f"{kind}_dtv_") and int(f.split("_")[2]) == want]
            if not match:
                continue
            tab = np.loadtxt(tmp_path / "dat" / match[0])
            assert tab.shape == (1201, 2)
            np.testing.assert_allclose(tab[:, 0], gold["winds"], atol=6e-9)
            np.testing.assert_allclose(tab[:, 1], scale * gold[f"ncar_{key}"][ik], atol=6e-9, rtol=0)
