"""The reference's own drivers — the interactive toy src/tests/aerobulk_toy.F90, src/tests/test_phymbl.f90 and src/ice/test_ice.f90, not a character changed — compiled against this repository's
Fortran modules (mod_const, mod_phymbl, mod_blk_coare3p0 / coare3p6 / ncar / ecmwf / andreas -> libaerobulk_amd.so -> HIP kernels;
oracle/_ref/dropin/aerobulk_toy.x, aerobulk_amd/build.py) and fed the inputs of the reference's test_algos.sh: every number it prints
(TURB_* with all OPTIONAL outputs, BULK_FORMULA, Ri_bulk, Theta_from_z_P0_T_q, q_sat, rho_air ... the five algorithms side by side, the
table of README.md:188-211) against what the same source prints when linked with the reference's own library
(tests/golden/ref_toy_outputs.json, tools/gen_ref_driver_golden.py)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")
CASES = json.load(open(os.path.join(GOLDEN, "ref_toy_outputs.json")))


def numbers(text):
    """every number the driver prints, in order (escape sequences and the exponents of the units taken out first; list-directed output
    wraps its lines at 80 columns wherever the digits happen to end, so the text is compared as one sequence)"""
    t = re.sub(r"\x1b\[[0-9;]*m", " ", text)
    t = re.sub(r"10\^-?\d+|m\^2|\^2|\*\*2|/m2|10m|2m|\bz0\b|N10", " ", t)
    return [float(x) for x in NUM.findall(t)]


def labelled(text, label):
    """the values behind `label =` (up to the next letter)"""
    m = re.search(re.escape(label) + r"\s*=\s*((?:[-+.0-9eE]+\s+)+)", text)
    return [float(x) for x in m.group(1).split()]


def test_golden_holds_the_readme_table():
    """The captured run is the README's toy table (README.md:188-211; the README is from an older revision: ~3 digits)."""
    toy = next(c for c in CASES if c["name"] == "test_algos.sh")
    cd = labelled(toy["stdout"], "C_D    ")
    assert len(cd) == 5
    np.testing.assert_allclose(cd, [1.1954, 1.0775, 1.2038, 1.2862, 1.0167], rtol=3e-3)     # coare3p0 coare3p6 ncar ecmwf andreas
    assert all(len(numbers(c["stdout"])) > (10 if c.get("exe") == "test_phymbl" else 150) for c in CASES)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_unchanged_driver_prints_what_it_prints_with_the_reference(case, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin", case.get("exe", "aerobulk_toy") + ".x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin not built (needs the reference tree and amdflang at build time)")
    # (cwd: test_ice.f90 writes z0_z0t_z0q__ustar_test.dat beside itself — not into the repository)
    pr = subprocess.run([exe, *case["args"]], input=case["stdin"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    got, ref = numbers(pr.stdout), numbers(case["stdout"])
    assert len(got) == len(ref) and len(ref) > (10 if case.get("exe") == "test_phymbl" else 150), (len(got), len(ref))
    if case.get("exe") == "test_phymbl":
        # src/tests/test_phymbl.f90, its potential-temperature / pressure branch: q_sat, Theta_from_z_P0_T_q, Pz_from_P0_tz_qz, pot_temp with the
        # OPTIONAL pPref, gamma_moist — scalar specifics (one-cell kernels); f7.3 / REAL(.,4) fields and two numbers at full precision
        np.testing.assert_allclose(got, ref, rtol=3e-7, atol=1e-9)
        return
    if case.get("exe") == "test_ice":
        # rough_leng_m / rough_leng_tq of mod_blk_ice_an05 (here: ab_phymbl functions 40 / 41) on 101 friction velocities, printed at 17 digits
        np.testing.assert_allclose(got, ref, rtol=1e-12)
        return
    # REAL(.,4) list-directed output: 7-8 digits; L near neutrality, gust and the RMS of five numbers are differences: absolute floor
    np.testing.assert_allclose(got, ref, rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(labelled(pr.stdout, "QL       "), labelled(case["stdout"], "QL       "), rtol=1e-6)
