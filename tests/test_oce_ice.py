"""A cell that is part open water (leads), part sea ice: the composition of the reference's src/ice/test_aerobulk_oce+ice.f90 — saturation
humidities over water and ice, TURB_ECMWF over the leads, TURB_ICE_NEMO / AN05 / LG15_IO over the ice, Ri_bulk, the moist lapse rate, the air
density at zu, BULK_FORMULA for both surfaces (l_ice for the ice), fluxes weighted by the ice fraction — on 256 cells.  The reference has no
entry of its own for such cells (that program composes them by hand and cannot run under amdflang: it re-opens unit 6 with RECL=); ONE driver
source (aerobulk_amd/fortran/oce_ice_driver.f90) is built against the unmodified reference (golden: tests/golden/oce_ice.npz,
tools/gen_oce_ice_golden.py) and against this repository's modules, where every call lands on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import phymbl_cases as pc
from conftest import GOLDEN, ROOT

DRV = os.path.join(ROOT, "aerobulk_amd", "fortran", "oce_ice_driver.x")


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "oce_ice.npz"))
    return z["inputs"], {k[2:]: z[k] for k in z.files if k.startswith("r_")}


def test_golden_is_a_mixed_cell_population(gold):
    x, rec = gold
    assert x.shape == (7, 256) and len(rec) == 72
    frci = x[5]
    assert frci.min() > 0.05 and frci.max() < 1.0
    for a in ("nemo", "an05", "lg15_io"):
        # the cell's flux is the weighted sum of its two surfaces (formed by the driver: a check of the record layout, not of physics)
        np.testing.assert_allclose(rec[f"{a}_qh_cell"], frci * rec[f"{a}_qh"] + (1. - frci) * rec["w_qh"], rtol=1e-14)
        assert np.all(rec[f"{a}_cd"] > 0) and np.all(rec[f"{a}_tau"] < 10.)
    # the ice part differs between the algorithms, the water part is one
    assert np.abs(rec["lg15_io_cd"] / rec["nemo_cd"] - 1.).max() > 0.05


def test_driver_is_built():
    assert os.path.exists(DRV), "python -m aerobulk_amd.build"


@pytest.mark.gpu
def test_engine_reproduces_the_references_mixed_cells(gold, tmp_path):
    x, rec = gold
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.ascontiguousarray(x, dtype=np.float64).tofile(fin)
    pr = subprocess.run([DRV, str(x.shape[1]), fin, fout], capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    got = pc.read_records(fout)
    assert set(got) == set(rec)
    worst = {}
    for k, r in rec.items():
        g = got[k]
        assert g.shape == r.shape and np.all(np.isfinite(g)), k
        # 1e-10 relative with the hot path's floor (1e-6 of the record's largest value: fluxes and L cross zero); the 20 iterations of the
        # sea-ice algorithms and of TURB_ECMWF are the engine's kernels, everything else its helper kernels
        scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max())
        worst[k] = float((np.abs(g - r) / scale).max())
        assert worst[k] <= 1e-10, (k, worst[k], int(np.argmax(np.abs(g - r) / scale)))
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:6])
