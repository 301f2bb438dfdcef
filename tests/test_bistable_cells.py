"""Cells whose flux the REFERENCE ITSELF computes in two ways, depending on the flag set of its own arch/ files (tests/golden/
bistable_cells.npz, tools/gen_bistable.py).  Found by the round-3 soak (seed 5119, profiles/r3_fuzz.txt item 10): an ECMWF + skin,
zt = zu = 10, nb_iter = 10 dead-calm night cell whose Q_L of record 3 the HIP kernel gives 2.56e-10 off the reference's default
build — the frozen metric (oracle/parity.py, untouched) rejects it at 1.255 of ONE_INPUT_CEILING = 1.25 — and equal to every digit to
the reference built with FMA contraction ("-xHOST -O3": arch/make.macro_OCCIGEN:17 and the ifort / ifx macros; the default build is
arch/make.macro_GnuLinux:17's plain -O3).

Rounds 3-5 kept these cells as fixtures under the rule "a value the frozen metric would reject must be one of the reference's own
builds' values to 1e-12".  ROUND 6 FOUND THE EXPRESSION and the rule is now the plain one: on every cell of the four fixtures the
kernel gives the reference's DEFAULT build (forward clause of the frozen metric, 1e-10), and on the three cells where the reference
has exactly two answers it gives the default one to 1e-12.

The expression: Ri_bulk's difference of two virtual temperatures, t (1 + c q) - t' (1 + c q') (mod_phymbl.f90:712-747), which ECMWF
re-evaluates in every iteration (mod_blk_ecmwf.f90:261).  Contracted into an FMA one product is rounded and the other is not; in
stable near-calm air the two agree to 1e-5 and ten passes of a non-contracting iteration amplify that asymmetric 1e-16 ~1e7 times.
Found by compiling the PRODUCT's physics header for the host with clang (which, with -ffp-contract=fast, gives the HIP kernel's
values on all four cells to four digits of the deviation) and switching contraction off per function: Ri_bulk alone moves all four
cells onto the default build (4e-15, 2e-15, 8e-16 relative; seed 11029's continuum cell to 4e-11).  The kernel now computes
ri_bulk / virt_temp under `#pragma clang fp contract(off)` (aerobulk_amd/csrc/ab_physics.hpp; the library is built with
-ffp-contract=fast-honor-pragmas) at three more instructions per call.

The fixtures hold the 96 cells around each cell: inputs and the unmodified reference's outputs under -O2, -O0, -O3, -O3 + FMA and
-O3 + fast-math, three records with the warm-layer state carried."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT6 = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
CAP = {"ql": "QL", "qh": "QH", "tau_x": "Tau_x", "tau_y": "Tau_y", "evap": "Evap", "t_s": "T_s"}
VARIANTS = ("O2", "O0", "O3", "O3fma", "fast")
ALGO, ZT, ZU, NITER, NT = "ecmwf", 10.0, 10.0, 10, 3
MATCH = 1e-12                 # "equals the default build of the reference": relative, on every flux of the soak's cell
FWD = 1e-10                   # forward clause of the frozen metric (oracle/parity.py), floor 1e-6 of the field's largest magnitude


# fixture file, record of the rejected value, the reference's own two answers in Q_L (relative gap between its default and its FMA build)
# Second fixture: round-4 soak, seed 9443 (profiles/r4_fuzz.txt item 10) — a near-calm stable night cell of the same configuration, record 2,
# rejected at 1.318 of the one-input ceiling; the kernels from before the round's last change give the same numbers.
# Third fixture of this kind: round 5, seed 11252 (profiles/r5_fuzz.txt), wind 0.33 m/s, stable, by day, record 3: rejected at 1.383 of the ceiling, the reference's FMA build to 1e-15.
FIXTURES = (("bistable_cells.npz", 3, (2e-10, 3e-10)), ("bistable_cells_9443.npz", 2, (6e-10, 7.5e-10)), ("bistable_cells_11252.npz", 3, (1.5e-10, 2.5e-10)))


def _load(name="bistable_cells.npz"):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d["in_" + k] for k in IN8}, {v: d["ref_" + v] for v in VARIANTS}, int(d["cell"])


@pytest.mark.parametrize("name,record,gap", FIXTURES)
def test_the_oracle_is_the_default_build_and_the_cell_has_two_reference_answers(oracle, name, record, gap):
    f, ref, c = _load(name)
    s = oracle.OracleSession(ALGO, f["sst"].size, NT, True)
    for jt in range(1, NT + 1):
        o = s.compute(jt, ZT, ZU, NITER, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        for i, k in enumerate(OUT6):
            np.testing.assert_array_equal(o[k], ref["O2"][jt - 1, i], err_msg=f"jt={jt} {k}")     # restatement == reference, bit for bit
    np.testing.assert_array_equal(ref["O0"], ref["O2"])
    np.testing.assert_array_equal(ref["O3"], ref["O2"])
    ql2, qlf = ref["O2"][record - 1, 0, c], ref["O3fma"][record - 1, 0, c]
    assert gap[0] < abs(qlf - ql2) / abs(ql2) < gap[1]               # the reference's own two answers: 2.56e-10 (6.8e-10) apart in Q_L
    others = np.delete(np.abs(ref["O3fma"] - ref["O2"]) / np.maximum(np.abs(ref["O2"]), 1e-30), c, axis=2)
    assert others.max() < 1e-12                                      # ... on this cell only: its neighbours agree to the last digits


# fourth cell, of another kind: round-5 closing soak, seed 11029 (profiles/r5_fuzz.txt) — a near-calm (0.23 m/s) stable cell by day, record 3, on which the
# reference's own builds form a CONTINUUM: Q_L = default - 4.8e-10 (fast-math), default (-O0 / -O2 / -O3), default + 5.6e-10 (FMA); round 5's kernel gave
# + 9.6e-10 and a bespoke "twice the spread" bar (dropped: ADVICE round 5).  With Ri_bulk uncontracted the kernel is 4e-11 from the default build: forward clause.
SPREAD_FIXTURE = ("bistable_cells_11029.npz", 3)
ALL_FIXTURES = tuple((x[0], x[1], True) for x in FIXTURES) + ((SPREAD_FIXTURE[0], SPREAD_FIXTURE[1], False),)


def test_a_cell_on_which_no_two_builds_of_the_reference_agree(oracle):
    name, record = SPREAD_FIXTURE
    f, ref, c = _load(name)
    s = oracle.OracleSession(ALGO, f["sst"].size, NT, True)
    for jt in range(1, NT + 1):
        o = s.compute(jt, ZT, ZU, NITER, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        for i, k in enumerate(OUT6):
            np.testing.assert_array_equal(o[k], ref["O2"][jt - 1, i], err_msg=f"jt={jt} {k}")     # restatement == default build, bit for bit
    np.testing.assert_array_equal(ref["O0"], ref["O2"])
    np.testing.assert_array_equal(ref["O3"], ref["O2"])
    ql = {v: ref[v][record - 1, 0, c] for v in VARIANTS}
    d_fma, d_fast = (ql["O3fma"] - ql["O2"]) / abs(ql["O2"]), (ql["fast"] - ql["O2"]) / abs(ql["O2"])
    assert 4e-10 < d_fma < 7e-10 and -6e-10 < d_fast < -3e-10          # one build above the default, one below: a continuum, not two states
    for v in ("O3fma", "fast"):
        others = np.delete(np.abs(ref[v] - ref["O2"]) / np.maximum(np.abs(ref["O2"]), 1e-30), c, axis=2)
        assert others.max() < 2e-13                                    # ... on this cell only


def assert_default_build(got, ref, c, exact, label):
    """got, ref: (NT, 6, n).  Every value inside the forward clause against the reference's DEFAULT build; the soak's cell, where the
    reference has two discrete answers (exact), equal to the default one to 1e-12 on every flux."""
    for jt in range(NT):
        for i, k in enumerate(OUT6):
            g, r = got[jt, i], ref[jt, i]
            fwd = np.abs(g - r) <= FWD * np.maximum(np.abs(r), 1e-6 * np.abs(r).max())
            assert fwd.all(), (label, jt + 1, k, np.nonzero(~fwd)[0], g[~fwd], r[~fwd])
            if exact:
                assert abs(g[c] - r[c]) <= MATCH * abs(r[c]), (label, jt + 1, k, float(g[c]), float(r[c]))


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU: the product's physics header on the host (tests/physics_host.cpp, TEST INFRASTRUCTURE) compiled by clang, the compiler that honours
# the header's `#pragma clang fp contract(off)` as hipcc does for the device.
CLANG = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"


def _host_exe(tmp, contract):
    exe = os.path.join(tmp, "physics_host_" + contract.replace("-", "_"))
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-ffp-contract=" + contract, "-march=x86-64-v3", "-DAB_PSI_LDS_TABLES=1", "-o", exe,
                           os.path.join(ROOT, "tests", "physics_host.cpp")])
    return exe


def _run_host(exe, tmp, f):
    n = f["sst"].size
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("<5iq2d", 4, 1, NITER, NT, 0, n, ZT, ZU))
        for k in IN8:
            np.ascontiguousarray(f[k], dtype=np.float64).tofile(fh)
    subprocess.check_call([exe, fin, fout])
    return np.fromfile(fout).reshape(NT, 6, n)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ (the pragma is clang's)")
def test_product_physics_on_host_gives_the_default_build_and_contraction_of_ri_bulk_was_the_cause(tmp_path):
    tmp = str(tmp_path)
    honoured = _host_exe(tmp, "fast-honor-pragmas")      # what the library is built with
    ignored = _host_exe(tmp, "fast")                     # clang's "fast" contracts in the back end whatever the pragma says: round 5's arithmetic
    for name, record, exact in ALL_FIXTURES:
        f, ref, c = _load(name)
        assert_default_build(_run_host(honoured, tmp, f), ref["O2"], c, exact, name)
        old = _run_host(ignored, tmp, f)[record - 1, 0, c]
        r2, rf = ref["O2"][record - 1, 0, c], ref["O3fma"][record - 1, 0, c]
        assert abs(old - r2) > 1.5e-10 * abs(r2)                          # the defect of rounds 3-5, reproduced on the host ...
        if exact:
            assert abs(old - rf) <= 1e-13 * abs(rf)                       # ... as the reference's own FMA answer, where it has two


@pytest.mark.gpu
@pytest.mark.parametrize("name,record,exact", ALL_FIXTURES)
def test_hip_gives_the_references_default_build(name, record, exact):
    import aerobulk_amd as ab
    f, ref, c = _load(name)
    n = f["sst"].size
    got = np.empty((NT, 6, n))
    with ab.Session(ALGO, n, 1, NT, True) as s:
        for jt in range(1, NT + 1):
            o = s.compute(jt, ZT, ZU, *[f[k] for k in IN8[:6]], Niter=NITER, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
            for i, k in enumerate(OUT6):
                got[jt - 1, i] = o[CAP[k]]
    assert_default_build(got, ref["O2"], c, exact, name)
