"""Lane regrouping of the flux kernel (include/aerobulk_amd.h: ab_session_set_regroup): which lane computes which cell must
not change a single output bit — compared with the natural order, for every tile size, ragged sizes and WL state carry-over."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")


def _run(algo, skin, f, n, nt=1, on=True, precision="f64"):
    import aerobulk_amd as ab
    outs = []
    with ab.Session(algo, n, 1, nt, skin, precision=precision) as s:
        s.set_regroup(on)
        for jt in range(1, nt + 1):
            outs.append(s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"] if skin else None,
                                  rad_lw=f["rad_lw"] if skin else None))
    return outs


@pytest.mark.parametrize("algo,skin", [("coare3p6", True), ("coare3p6", False), ("coare3p0", True), ("ecmwf", True), ("ecmwf", False),
                                       ("ncar", False), ("andreas", False)])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_regrouped_bits_equal_natural_order(oracle, algo, skin, precision):
    ni, nj = 1013, 37                      # 37481 cells: ragged last tile for every tile size
    f = oracle.synth_fields(ni, nj)
    n = ni * nj
    ref = _run(algo, skin, f, n, on=False, precision=precision)[0]
    got = _run(algo, skin, f, n, precision=precision)[0]
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{algo} {k}")
        assert np.isfinite(got[k]).all()


def test_regrouped_multi_record_warm_layer_state(oracle):
    ni, nj = 700, 23
    f = oracle.synth_fields(ni, nj)
    f["u_zu"] = f["u_zu"] * 0.2            # light winds: the warm layer builds up over the records
    f["v_zu"] = f["v_zu"] * 0.2
    n = ni * nj
    ref = _run("coare3p6", True, f, n, nt=4, on=False)
    got = _run("coare3p6", True, f, n, nt=4)
    assert np.abs(ref[3]["T_s"] - f["sst"]).max() > 1.0
    for a, b in zip(got, ref):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_regrouped_ecmwf_series_day_then_night(oracle):
    """WL_ECMWF over a series: two sunny records build the layer, two dark ones erode it (the ten-pass branch of cells that lose heat
    WITH a layer, which the forecast bins separately from the second record on; cells without a layer take the exact shortcut).
    Regrouped == natural order bit for bit, and both follow the oracle."""
    import aerobulk_amd as ab
    ni, nj = 640, 29
    f = oracle.synth_fields(ni, nj)
    f["u_zu"] = f["u_zu"] * 0.15
    f["v_zu"] = f["v_zu"] * 0.15
    n = ni * nj
    sw = [f["rad_sw"] * 1.0 + 200.0, f["rad_sw"] * 1.0 + 200.0, f["rad_sw"] * 0.0, f["rad_sw"] * 0.0]

    def run(on):
        outs = []
        with ab.Session("ecmwf", n, 1, 4, True) as s:
            s.set_regroup(on)
            for jt in range(1, 5):
                outs.append(s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=sw[jt - 1], rad_lw=f["rad_lw"]))
        return outs

    ref, got = run(False), run(True)
    assert (ref[1]["T_s"] - f["sst"]).max() > 0.3                       # a layer was built ...
    assert ((ref[1]["T_s"] - ref[3]["T_s"]) > 0.05).mean() > 0.2       # ... and eroded at night
    for a, b in zip(got, ref):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    o = oracle.OracleSession("ecmwf", n, 4, True)
    for jt in range(1, 5):
        want = o.compute(jt, 2.0, 10.0, 5, *[f[k] for k in IN6], rad_sw=sw[jt - 1], rad_lw=f["rad_lw"])
        for k, ko in (("QL", "ql"), ("QH", "qh"), ("T_s", "t_s")):
            np.testing.assert_allclose(got[jt - 1][k], want[ko], rtol=1e-9, atol=1e-9, err_msg=f"jt={jt} {k}")


def test_regrouped_tiny_and_large_grids(oracle):
    for ni, nj in ((7, 1), (300, 1), (4320, 400)):      # fewer cells than one wave / one tile / 1.7 M cells
        f = oracle.synth_fields(ni, nj)
        n = ni * nj
        ref = _run("ecmwf", True, f, n, on=False)[0]
        got = _run("ecmwf", True, f, n)[0]
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{ni}x{nj} {k}")


@pytest.mark.parametrize("algo,cs,wl", [("coare3p6", True, True), ("coare3p6", True, False), ("coare3p0", False, True),
                                        ("ecmwf", True, True), ("coare3p6", False, False), ("andreas", False, False)])
def test_regrouped_turb_bits_equal_natural_order(oracle, algo, cs, wl):
    """The TURB_* entry (ab_session_turb) runs on the same tile machinery."""
    import aerobulk_amd as ab
    ni, nj = 911, 29
    f = oracle.synth_fields(ni, nj)
    n = ni * nj
    L = oracle.lib()
    wnd = np.sqrt(f["u_zu"] ** 2 + f["v_zu"] ** 2)
    theta = f["t_zt"] + 0.0196
    ssq = 0.98 * 3.8e-3 * np.exp(0.0687 * (f["sst"] - 273.15))
    skin = cs or wl
    res = []
    for on in (False, True):
        with ab.Session(algo, n, 1, 3, False) as s:
            s.set_regroup(on)
            d = s.set_diagnostics(("CdN", "u_star", "L", "dT_cs", "dT_wl"))
            rec = []
            for kt in (1, 2, 3):
                T_s, q_s = f["sst"].copy(), ssq.copy()
                o = s.turb(kt, 2.0, 10.0, T_s, theta, q_s, f["hum_zt"], wnd, cs, wl, Qsw=0.934 * f["rad_sw"] if skin else None,
                           rad_lw=f["rad_lw"] if skin else None, slp=f["slp"] if skin else None, nb_iter=5)
                rec.append({**{k: v.copy() for k, v in o.items()}, **{k: v.copy() for k, v in d.items()}, "T_s": T_s, "q_s": q_s})
            res.append(rec)
    for a, b in zip(*res):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{algo} cs={cs} wl={wl} {k}")
