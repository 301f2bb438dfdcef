"""CPU: the PRODUCT's per-cell physics header (aerobulk_amd/csrc/ab_physics.hpp + ab_fastmath.hpp) compiled for the host
(tests/physics_host.cpp, hardware seeds emulated at their accuracy) against the golden vectors of the unmodified reference and
the 17-digit pins: every algorithm, skin on/off, nb_iter 1/5/8, zt = zu and zt != zu, three humidity types, warm-layer carry-over.
A regression net for the physics that needs no GPU (the GPU parity tests remain the gate: the seeds differ in their last bits).
Test infrastructure only: nothing in the library runs on the host."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_hot_parity, load_golden_case, load_manifest, sensitivity

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
ALGOS = {"coare3p0": 1, "coare3p6": 2, "ncar": 3, "ecmwf": 4, "andreas": 5}
HUM = {"sh": 0, "dp": 1, "rh": 2}


# two builds: the polynomial psi functions (turb / ice kernels, NCAR) and the piecewise LDS tables of the tiled flux kernels
@pytest.fixture(scope="module", params=["psi polynomials", "psi tables"])
def exe(request, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("physics_host") / "physics_host")
    flags = ["-DAB_PSI_LDS_TABLES=1"] if request.param == "psi tables" else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=fast", "-march=x86-64-v3", *flags, "-o", out,
                           os.path.join(ROOT, "tests", "physics_host.cpp")])
    return out


def run_host(exe, tmp_path, algo, skin, niter, nt, hum_type, zt, zu, f):
    n = f["sst"].size
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("<5iq2d", ALGOS[algo], int(skin), niter, nt, HUM[hum_type], n, zt, zu))
        for k in IN8:
            np.ascontiguousarray(f.get(k, np.zeros(n)) if f.get(k) is not None else np.zeros(n), dtype=np.float64).tofile(fh)
    subprocess.check_call([exe, fin, fout])
    return np.fromfile(fout).reshape(nt, 6, n)


@pytest.mark.parametrize("case", load_manifest(), ids=lambda c: c["name"])
def test_product_physics_on_host_matches_reference_golden(oracle, exe, tmp_path, case):
    inp, recs, keys = load_golden_case(case)
    got = run_host(exe, tmp_path, case["algo"], case["skin"], case["niter"], case["nt"], case["hum_type"], case["zt"], case["zu"], inp)
    sens = sensitivity(oracle, case["algo"], case["skin"], case["zt"], case["zu"], case["niter"], inp, nt=case["nt"], hum_type=case["hum_type"])
    order = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
    for jt, ref in enumerate(recs, 1):
        g = {k: got[jt - 1, order.index(k)] for k in keys}
        assert_hot_parity(g, ref, keys, sens=sens, jt=jt, label=f"host physics {case['name']} jt={jt}", quiet=True)


def test_product_physics_on_host_reproduces_pins(exe, tmp_path):
    p = json.load(open(os.path.join(GOLDEN, "pins_2cell.json")))
    f = {k: np.array(v, dtype=np.float64) for k, v in p["inputs"].items()}
    order = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
    for name, outs in p["outputs"].items():
        algo, sk = name.rsplit("_", 1)
        got = run_host(exe, tmp_path, algo, sk == "skin", p["niter"], 1, "sh", p["zt"], p["zu"], f)
        for k, hexes in outs.items():
            ref = np.array([float.fromhex(h) for h in hexes])
            np.testing.assert_allclose(got[0, order.index(k)], ref, rtol=1e-11, atol=0, err_msg=f"{name} {k}")


def test_product_physics_on_host_within_the_references_spread_on_illconditioned_cells(exe, tmp_path):
    """The fixture of tests/test_illcond_cells.py (cells where a forward 1e-10 is undefined): the host build of the product's physics
    also stays within max(1e-10 bar, reference spread under <= 4 ulp input moves and its own build flags) of the reference."""
    d = np.load(os.path.join(GOLDEN, "illcond_cells.npz"))
    meta = json.loads(str(d["meta"]))
    worst = 0.0
    for tag, m in meta.items():
        f = {k: np.ascontiguousarray(d[tag + "_inputs"][i]) for i, k in enumerate(IN8)}
        nf = 6 if m["skin"] else 5
        got = run_host(exe, tmp_path, m["algo"], m["skin"], m["niter"], m["nt"], "sh", m["zt"], m["zu"], f)
        ref, scale = d[tag + "_ref"], d[tag + "_scale"]
        err = np.abs(got - ref)[:, :nf]
        bar = 1e-10 * np.maximum(np.abs(ref[:, :nf]), 1e-6 * scale[None, :nf, None])
        allowed = np.maximum(bar, d[tag + "_spread_ref4"][:, :nf])
        worst = max(worst, float((err / allowed).max()))
        assert np.all(err <= allowed), (tag, float((err / allowed).max()))
    print("largest |host physics - reference| / max(bar, S_ref4):", worst)
