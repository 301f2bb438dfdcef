"""GPU: the fp32-array sessions against the fp64 ORACLE (BASELINE.json config 5, "fp32 path ... tolerance re-stated").

No fp32 reference exists (wp = dp, mod_const.f90:10-12), and the reference warns about single precision in exactly this scheme
(mod_blk_ecmwf.f90:556-561): the differences theta_zu - T_s and q_zu - q_s.  The tolerance is restated against the fp64 oracle fed
with the fp32-rounded inputs (the numbers a session receives), as SURVEY §8d proposes:

    |x - ref| <= 1e-4 max(|ref|, floor)     floor = 1 W/m2 (Q_L, Q_H), 1e-3 N/m2 (tau), 4e-7 kg/m2/s (E)
    i.e. 1e-4 relative where |flux| > 1 W/m2, 1e-4 absolute (in W/m2-equivalents) below; T_s: 1e-4 K absolute;
    outliers counted, at most 1e-5 of the cells of a field.

AB_F32_MIXED (what `bench.py --config 5` times): fp32 arrays, fp64 anchors (SST, theta, T_s, q, q_s, their differences, q_sat), fp32
hardware transcendentals elsewhere: meets the bar.  AB_F32_STORAGE (fp64 arithmetic on fp32 arrays): one rounding of the result.
AB_F32 (fp32 arithmetic throughout): does NOT meet it (p99 2e-4, p99.99 1e-3); its error is bounded by quantile for what it is.
"""
import numpy as np
import pytest

from conftest import _log, oracle_on_cells

pytestmark = pytest.mark.gpu
IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT = (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s"))
FLOOR = {"ql": 1.0, "qh": 1.0, "tau_x": 1e-3, "tau_y": 1e-3, "evap": 4e-7, "t_s": None}   # 1 W/m2 and its equivalents; T_s absolute [K]
TOL = 1e-4                  # SURVEY §8d's proposal for config 5
OUTLIER_SHARE = 1e-5        # ... "outlier count reported": asserted


def restated_error(got, ref, key):
    d = np.abs(np.asarray(got, dtype=np.float64) - ref)
    return d if FLOOR[key] is None else d / np.maximum(np.abs(ref), FLOOR[key])


def oracle_parallel(algo, skin, niter, f64, nt=1):
    """The oracle on the host cores, cells cut into chunks (pointwise path): list over records of dicts of arrays.  Fresh worker
    interpreters (conftest.oracle_pool: forkserver), never a fork of this process, which holds the HIP runtime."""
    return oracle_on_cells(algo, skin, niter, f64, nt=nt)


def check_restated(got, ref, label, tol=TOL, share=OUTLIER_SHARE, p99=None, tol_ts=None):
    """Asserts the restated tolerance with its outlier budget; returns the report.  tol_ts: bar of T_s [K] if not `tol` (T_s is an fp32
    number of about 300 K: its last bit is 3e-5 K)."""
    rep = {}
    for c, k in OUT:
        if c not in got:
            continue
        e = restated_error(got[c], ref[k], k)
        n_out = int((e > (tol_ts if (tol_ts is not None and FLOOR[k] is None) else tol)).sum())
        rep[k] = dict(p50=float(np.quantile(e, 0.5)), p99=float(np.quantile(e, 0.99)), p9999=float(np.quantile(e, 0.9999)), max=float(e.max()),
                      n=int(e.size), n_gt_tol=n_out)
    print(label, rep)
    for k, r in rep.items():
        assert np.isfinite(r["max"]), (label, k, r)
        assert r["n_gt_tol"] <= int(share * r["n"]), (label, k, r)
        if p99 is not None and FLOOR[k] is not None:      # (T_s is an fp32 number of about 300 K: half an ulp is 1.5e-5 K)
            assert r["p99"] <= p99, (label, k, r)
    return rep


@pytest.mark.parametrize("algo", ["ecmwf", "coare3p6", "coare3p0"])
def test_mixed_sessions_meet_the_restated_tolerance(oracle, algo):
    """360x180, the three algorithms with skin schemes, three records with the warm-layer state carried (fp32 planes): every value
    within 1e-4 (no outlier allowed at 64 800 cells: 1e-5 of them is less than one), p99 <= 3e-6 (measured 9e-7)."""
    import aerobulk_amd as ab
    ni, nj, nt = 360, 180, 3
    f = oracle.synth_fields(ni, nj)
    f32 = {k: f[k].astype(np.float32) for k in IN8}
    f64 = {k: f32[k].astype(np.float64) for k in IN8}
    ref = oracle_parallel(algo, True, 5, f64, nt=nt)
    with ab.Session(algo, ni, nj, nt, True, precision="f32_mixed") as s:
        for jt in range(1, nt + 1):
            g = s.compute(jt, 2.0, 10.0, *[f32[k] for k in IN8[:6]], Niter=5, rad_sw=f32["rad_sw"], rad_lw=f32["rad_lw"])
            assert all(v.dtype == np.float32 for v in g.values())
            check_restated(g, ref[jt - 1], f"mixed {algo} 360x180 jt={jt}", p99=3e-6)


@pytest.mark.parametrize("algo,skin,niter", [("ecmwf", False, 8), ("coare3p6", False, 5), ("coare3p0", False, 5), ("ncar", False, 5), ("andreas", False, 5),
                                             ("ecmwf", True, 8), ("coare3p6", True, 2)])
def test_mixed_mode_serves_every_flux_configuration(oracle, algo, skin, niter):
    """The mode is a property of the session, not of one kernel: all five algorithms, other pass counts."""
    import aerobulk_amd as ab
    ni, nj = 360, 180
    f = oracle.synth_fields(ni, nj)
    f32 = {k: f[k].astype(np.float32) for k in IN8}
    f64 = {k: f32[k].astype(np.float64) for k in IN8}
    ref = oracle_parallel(algo, skin, niter, f64)[0]
    with ab.Session(algo, ni, nj, 1, skin, precision="f32_mixed") as s:
        g = s.compute(1, 2.0, 10.0, *[f32[k] for k in IN8[:6]], Niter=niter, rad_sw=f32["rad_sw"] if skin else None, rad_lw=f32["rad_lw"] if skin else None)
    # ANDREAS reads its roughness lengths from a table with jumps (z0tq_LKB, mod_phymbl.f90:1658-1667): one cell in 64 800 may sit on one
    check_restated(g, ref, f"mixed {algo} skin={skin} n={niter}", share=2e-5 if algo == "andreas" else OUTLIER_SHARE, p99=3e-6)


def test_mixed_mode_with_relative_humidity_and_device_arrays(oracle):
    """Humidity given as relative humidity (q_air_rh in fp64, parked as a float) and the arrays resident on the device."""
    import torch
    import aerobulk_amd as ab
    ni, nj = 256, 96
    f = oracle.synth_fields(ni, nj)
    f32 = {k: f[k].astype(np.float32) for k in IN8}
    qsat = np.array([oracle.lib().abo_q_sat(float(t), float(p)) for t, p in zip(f32["t_zt"].astype(np.float64), f32["slp"].astype(np.float64))])
    f32["hum_zt"] = (100.0 * np.clip(f32["hum_zt"].astype(np.float64) / qsat, 0., 1.)).astype(np.float32)      # % relative humidity
    f64 = {k: f32[k].astype(np.float64) for k in IN8}
    o = oracle.OracleSession("ecmwf", ni * nj, 1, True, "rh").compute(1, 2.0, 10.0, 5, *[f64[k] for k in IN8[:6]], rad_sw=f64["rad_sw"], rad_lw=f64["rad_lw"])
    dev = {k: torch.from_numpy(v).cuda() for k, v in f32.items()}
    with ab.Session("ecmwf", ni, nj, 1, True, precision="f32_mixed") as s:
        s.set_humidity("rh")
        g = s.compute(1, 2.0, 10.0, *[dev[k] for k in IN8[:6]], Niter=5, rad_sw=dev["rad_sw"], rad_lw=dev["rad_lw"])
    check_restated({k: v.cpu().numpy() for k, v in g.items()}, o, "mixed ecmwf rh device", p99=3e-6)


def test_orca36_fp32_sessions_against_the_oracle_on_a_subsample():
    """BASELINE config 5 at its full size on ONE GPU (12960 x 10800 = 140 M cells, ECMWF + cool-skin/warm-layer, fp32 arrays).  The oracle
    cannot do 140 M cells in a test; it does every 31st (4.5 M cells; the stride is coprime to the row length, so all columns
    are visited) on the host cores, and each of the three fp32-array modes is checked against it:
      AB_F32_MIXED    the restated tolerance, outliers <= 1e-5 of the cells (the timed path of config 5);
      AB_F32_STORAGE  <= 1e-6 (one rounding of the fp64 result);
      AB_F32          p99.9 <= 2e-3 and <= 1e-4 of the cells beyond 1e-2 (what round 2 asserted against the fp64 HIP path, now against
                      the oracle): outside the restated tolerance, which is why it is not the headline.
    Then j-block invariance at that size (a block computed alone = the same rows of the full launch, bit for bit)."""
    import torch
    import aerobulk_amd as ab
    from oracle import pyoracle as po
    import time
    t0 = time.time()

    def stage(what):      # (this test took 5 s on most leases and 665 s on one: the run log says where)
        torch.cuda.synchronize()
        _log(f"[ab-test]   orca36 stage: {what}  t+{time.time() - t0:.1f}s")

    ni, nj, stride = 12960, 10800, 31
    names = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp", "rad_sw", "rad_lw")
    f = ab.synth_fields_device(ni, nj, precision="f32")
    stage("fields generated on the device")
    sub = {k.lower(): f[k][::stride].cpu().numpy().astype(np.float64) for k in names}
    assert sub["sst"].size >= 4_000_000
    stage("subsample on the host")
    ref = oracle_parallel("ecmwf", True, 5, sub)[0]
    stage("oracle on the subsample")
    del po
    full = {}
    for prec in ("f32_mixed", "f32_storage", "f32"):
        with ab.Session("ecmwf", ni, nj, 1, True, precision=prec) as s:
            got = s.compute(1, 2.0, 10.0, *[f[k] for k in names[:6]], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        stage(f"{prec}: computed")
        for k in got:
            assert bool(torch.isfinite(got[k]).all()), (prec, k)
        g = {k: v[::stride].cpu().numpy() for k, v in got.items()}
        stage(f"{prec}: finite, subsample on the host")
        if prec == "f32_mixed":
            check_restated(g, ref, "ORCA36 subsample f32_mixed", p99=3e-6)
            full = got
        elif prec == "f32_storage":
            check_restated(g, ref, "ORCA36 subsample f32_storage", tol=1e-6, share=0.0, tol_ts=2e-5)
        else:
            for c, k in OUT:
                e = restated_error(g[c], ref[k], k)
                frac = float((e > 1e-2).mean())
                print(f"ORCA36 subsample f32 {k}: p99.9 {np.quantile(e, 0.999):.2e} max {e.max():.2e} share beyond 1e-2 {frac:.2e}")
                assert np.quantile(e, 0.999) <= 2e-3 and frac <= 1e-4, (k, frac)
        if prec != "f32_mixed":
            del got
    # a j-block computed alone (what one of 8 ranks owns) is bit-identical to the same rows of the full launch
    stage("three modes checked")
    j0, njl = 4050, 1350
    fs = ab.synth_fields_device(ni, nj, j0, njl, precision="f32")
    with ab.Session("ecmwf", ni, njl, 1, True, precision="f32_mixed") as s:
        part = s.compute(1, 2.0, 10.0, *[fs[k] for k in names[:6]], Niter=5, rad_sw=fs["rad_sw"], rad_lw=fs["rad_lw"])
    for k in full:
        assert torch.equal(full[k][j0 * ni:(j0 + njl) * ni], part[k]), k
    stage("j-block invariance")
