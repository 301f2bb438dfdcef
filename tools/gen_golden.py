#!/usr/bin/env python
"""tools/gen_golden.py — generate the golden input/output vectors under tests/golden/.

Runs in the BUILD container only: it drives the UNMODIFIED reference (oracle/_ref/libaerobulk_ref.so,
compiled from /root/reference by oracle/Makefile) through its own C entry points and stores inputs and
reference outputs as .npz.  The fixtures are data (float64 arrays), never source.

    python tools/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")


def halton(n, base):
    r = np.zeros(n)
    for i in range(n):
        f, x, k = 1.0, 0.0, i + 1
        while k > 0:
            f /= base
            x += f * (k % base)
            k //= base
        r[i] = x
    return r


def sweep_inputs(n=2048):
    """Stratified sweep: stable/unstable, calm .. hurricane winds, Charnock / LKB / Ri thresholds, rad on/off."""
    h = [halton(n, b) for b in (2, 3, 5, 7, 11, 13, 17)]
    sst = 271.5 + 33.0 * h[0]
    dT = -8.0 + 14.0 * h[1]                       # t_zt - sst in [-8, +6]
    t_zt = sst + dT
    slp = 95000.0 + 9000.0 * h[2]
    rh = 0.40 + 0.60 * h[3]
    qsat = np.array([po.lib().abo_q_sat(t, p) for t, p in zip(t_zt, slp)])
    q = rh * qsat
    # wind speed ladder: thresholds of the algorithms (0.2/0.25/0.5 floors, 10/18 Charnock, 33 NCAR) + continuous part
    ladder = np.array([0.0, 1e-4, 5e-4, 0.05, 0.2, 0.25, 0.5, 1.0, 2.0, 3.5, 5.0, 7.5, 10.0, 10.0001, 14.0, 18.0,
                       18.0001, 22.0, 27.0, 33.0, 33.0001, 38.0])
    wind = np.where(np.arange(n) % 3 == 0, ladder[(np.arange(n) // 3) % ladder.size], 30.0 * h[4] ** 1.5)
    ang = 2.0 * np.pi * h[5]
    u, v = wind * np.cos(ang), wind * np.sin(ang)
    rad_sw = np.where(np.arange(n) % 4 == 0, 0.0, 1000.0 * h[6])
    rad_lw = 250.0 + 200.0 * h[0]
    # a few exact edge cells
    t_zt[:8] = sst[:8]                             # zero air-sea temperature difference
    q[8:16] = 0.98 * np.array([po.lib().abo_q_sat(t, p) for t, p in zip(sst[8:16], slp[8:16])])  # zero dq
    u[16:20], v[16:20] = 0.0, 0.0                  # exactly calm
    return dict(sst=sst, t_zt=t_zt, hum_zt=q, u_zu=u, v_zu=v, slp=slp, rad_sw=rad_sw, rad_lw=rad_lw)


def run(algo, f, zt, niter, skin, nt=1, hum=None):
    ff = dict(f)
    if hum is not None:
        ff["hum_zt"] = hum
    rec = {k: ff[k] for k in IN6}
    if skin:
        rec["rad_sw"], rec["rad_lw"] = ff["rad_sw"], ff["rad_lw"]
    res = po.run_reference(algo, [rec] * nt, zt, 10.0, niter, use_skin=skin)
    keys = ("ql", "qh", "tau_x", "tau_y", "evap") + (("t_s",) if skin else ())
    return [{k: r[k] for k in keys} for r in res]


def main():
    assert po.have_reference(), "build oracle/_ref first (make -C oracle ref)"
    os.makedirs(OUT, exist_ok=True)
    f = sweep_inputs()
    np.savez_compressed(os.path.join(OUT, "sweep_inputs.npz"), **f)
    manifest = []

    def add(name, algo, zt, niter, skin, nt=1, hum_type="sh", hum=None):
        res = run(algo, f, zt, niter, skin, nt, hum)
        arrays = {}
        for jt, r in enumerate(res, 1):
            for k, v in r.items():
                arrays[f"jt{jt}_{k}"] = v
        if hum is not None:
            arrays["hum_zt"] = hum
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        manifest.append(dict(name=name, algo=algo, zt=zt, zu=10.0, niter=niter, skin=skin, nt=nt, hum_type=hum_type))
        print("wrote", name)

    for algo in po.ALGOS:
        for skin in ((False, True) if algo in po.SKIN_ALGOS else (False,)):
            tag = f"{algo}_{'skin' if skin else 'noskin'}"
            add(f"{tag}_n5_zt2", algo, 2.0, 5, skin)
            add(f"{tag}_n8_zt10", algo, 10.0, 8, skin)
    add("coare3p6_noskin_n1_zt2", "coare3p6", 2.0, 1, False)
    for algo in po.SKIN_ALGOS:          # warm-layer carry-over across 3 records
        add(f"{algo}_skin_n8_zt2_nt3", algo, 2.0, 8, True, nt=3)
    qsat = np.array([po.lib().abo_q_sat(t, p) for t, p in zip(f["t_zt"], f["slp"])])
    rh = 100.0 * np.clip(f["hum_zt"] / qsat, 0.0, 1.0)
    add("coare3p6_noskin_n5_zt2_rh", "coare3p6", 2.0, 5, False, hum_type="rh", hum=rh)
    add("ncar_noskin_n5_zt2_dp", "ncar", 2.0, 5, False, hum_type="dp", hum=f["t_zt"] - 2.5)
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)

    # TURB_* diagnostics (mandatory + OPTIONAL outputs) from the reference's routines called directly
    # (oracle/ref_turb_driver.f90), first 1024 cells of the sweep
    nd = 1024
    fd = {k: np.ascontiguousarray(v[:nd]) for k, v in f.items()}
    diag_manifest = []
    for algo in po.ALGOS:
        for skin in ((False, True) if algo in po.SKIN_ALGOS else (False,)):
            for zt, niter in ((2.0, 5), (10.0, 8)):
                r = po.run_reference_turb(algo, fd, zt, 10.0, niter, skin)
                name = f"diag_{algo}_{'skin' if skin else 'noskin'}_n{niter}_zt{int(zt)}"
                np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
                diag_manifest.append(dict(name=name, algo=algo, zt=zt, zu=10.0, niter=niter, skin=skin, n=nd))
                print("wrote", name)
    with open(os.path.join(OUT, "diag_manifest.json"), "w") as fh:
        json.dump(diag_manifest, fh, indent=1)

    # the reference's own captured example output doc/ex_ab.dat (nb_iter=50, 7 significant digits):
    # inputs from src/tests/example_call_aerobulk.f90:29-44, printed values from doc/ex_ab.dat:28-33,62-67,96-101,130-134,163-167
    ex = dict(
        inputs=dict(sst=[295.15, 295.15], t_zt=[293.15, 298.15], hum_zt=[0.012, 0.012], u_zu=[5.0, 5.0], v_zu=[0.0, 0.0],
                    slp=[101000.0, 101000.0], rad_sw=[0.0, 0.0], rad_lw=[350.0, 350.0], zt=2.0, zu=10.0, niter=50),
        cases={
            "coare3p0": dict(skin=True, qh=[-15.15451, 17.84016], ql=[-81.38467, -50.83000], evap_mm_day=[-2.870619, -1.792949],
                             t_s_degC=[21.72197, 21.75757], tau_x=[3.5783499e-02, 1.7346080e-02], loose=True),
            "coare3p6": dict(skin=True, qh=[-15.38655, 17.08068], ql=[-83.07884, -48.43779], evap_mm_day=[-2.930330, -1.708553],
                             t_s_degC=[21.70580, 21.74849], tau_x=[3.2181785e-02, 1.5142991e-02]),
            "ecmwf": dict(skin=True, qh=[-14.38223, 17.65283], ql=[-80.29590, -52.46112], evap_mm_day=[-2.832244, -1.850494],
                          t_s_degC=[21.73254, 21.76303], tau_x=[3.8438912e-02, 1.9324517e-02]),
            "ncar": dict(skin=False, qh=[-16.69695, 10.72617], ql=[-88.47819, -71.90122], evap_mm_day=[-3.121663, -2.536799],
                         tau_x=[3.5851959e-02, 2.7732996e-02]),
            "andreas": dict(skin=False, qh=[-14.41300, 15.19631], ql=[-74.46378, -51.69934], evap_mm_day=[-2.627210, -1.824042],
                            tau_x=[3.0277077e-02, 1.7942309e-02]),
        })
    with open(os.path.join(OUT, "ex_ab.json"), "w") as fh:
        json.dump(ex, fh, indent=1)

    # SURVEY §8c pins measured on the compiled reference (17 digits), re-generated here
    n = 2
    pin_f = dict(sst=np.full(n, 295.15), t_zt=np.array([293.15, 298.15]), hum_zt=np.full(n, 0.012), u_zu=np.full(n, 4.0),
                 v_zu=np.full(n, 9.0), slp=np.full(n, 101000.0), rad_sw=np.full(n, 600.0), rad_lw=np.full(n, 350.0))
    pins = {}
    for algo in po.ALGOS:
        for skin in ((False, True) if algo in po.SKIN_ALGOS else (False,)):
            rec = {k: pin_f[k] for k in IN6}
            if skin:
                rec["rad_sw"], rec["rad_lw"] = pin_f["rad_sw"], pin_f["rad_lw"]
            r = po.run_reference(algo, [rec], 2.0, 10.0, 8, use_skin=skin)[0]
            pins[f"{algo}_{'skin' if skin else 'noskin'}"] = {k: [float.hex(float(x)) for x in r[k]] for k in
                                                               ("ql", "qh", "tau_x", "tau_y", "evap") + (("t_s",) if skin else ())}
    with open(os.path.join(OUT, "pins_2cell.json"), "w") as fh:
        json.dump(dict(inputs={k: [float(x) for x in v] for k, v in pin_f.items()}, zt=2.0, zu=10.0, niter=8, outputs=pins), fh, indent=1)
    print("done:", len(manifest), "sweep cases")


if __name__ == "__main__":
    main()
