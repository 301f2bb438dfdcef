"""mod_skin_coare / mod_skin_ecmwf at the Fortran boundary: CS_COARE, CS_ECMWF, WL_COARE, WL_ECMWF standing alone and the
warm layer's state as the modules' PUBLIC arrays (/root/reference/src/mod_skin_coare.f90:28-38,48-93,97-250, mod_skin_ecmwf.f90:49-57,68-110,113-230).

Golden data: aerobulk_amd/fortran/skin_driver.f90 — ONE source — linked against the UNMODIFIED reference modules (oracle/_ref/ref_skin_driver.x,
tools/gen_skin_golden.py -> tests/golden/skin_modules.npz).  Checked three ways:
  * CPU: the product's header on the host (tests/phymbl_host.cpp, test infrastructure) runs the driver's sequences through ph_cell<57..60>;
  * GPU: the same sequences through the C ABI (`ab_phymbl`, elementwise HIP kernels: one launch per step over all cells);
  * GPU: skin_driver.x built against THIS repository's modules (Fortran host -> C ABI -> HIP; one-cell launches, as an unchanged caller would issue
    them), every record the reference-linked build wrote, incl. Qnt_ac / Tau_ac / dT_wl / Hz_wl read from the modules after TURB_COARE3P6 / 3P0 / ECMWF
    (what src/tests/test_aerobulk_buoy_series_oce.f90:16,463-464 does).
Tolerance: 1e-12 relative (floor 1e-12 of the record's largest magnitude) for the schemes; the TURB_* records inherit the flux path's 1e-10."""
import os
import struct
import subprocess

import numpy as np
import pytest

import skin_cases as sc
from conftest import GOLDEN, ROOT

TOL = 1e-12


@pytest.fixture(scope="module")
def gold():
    d = np.load(os.path.join(GOLDEN, "skin_modules.npz"))
    return d["columns"], {k[2:]: d[k] for k in d.files if k.startswith("r_")}


def close(got, ref, label, tol=TOL):
    scale = max(np.abs(ref).max(), 1e-300)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale)
    bad = np.nonzero(~(err <= tol))[0]
    assert bad.size == 0, (label, bad[:5], got[bad[:5]], ref[bad[:5]], float(np.nanmax(err)))
    return float(err.max()) if err.size else 0.0


def test_golden_covers_the_branches(gold):
    cols, rec = gold
    assert len(rec) == 78 and cols.shape == (13, 256)
    assert list(rec["parameters"]) == [sc.HWL_MAX, sc.RD0]
    np.testing.assert_array_equal(rec["wlc_wait_dT"], rec["wlc_dT_01"])          # iwait /= 0: WL_COARE leaves the state alone
    np.testing.assert_array_equal(rec["wlc_wait_Qac"], rec["wlc_Qac_01"])
    grown = rec["wlc_dT_04"] > 0.01
    assert grown.sum() > 40 and rec["wlc_Hz_04"][grown].min() < 19.0             # a warm layer was built, shallower than Hwl_max
    dawn = (rec["wlc_dT_06"] > 0) & (rec["wlc_dT_07"] == 0) & (rec["wlc_Hz_07"] == sc.HWL_MAX)
    assert dawn.sum() > 5                                                        # ... and destroyed by the dawn reset where 4h30 UTC is ]4h, 6h30] solar
    assert ((rec["wlc_dT_06"] > 0) & (rec["wlc_dT_07"] > 0)).sum() > 5           # ... but not elsewhere
    assert (rec["cs_ecmwf_heating"] > 0).any() and (rec["cs_coare"] < 0).any()
    assert (rec["wle_dT_04"] > 0.05).sum() > 20 and (rec["t36_Qac_03"] > 0).sum() > 50 and (rec["tec_dT_03"] > 0).any()


def run_sequences(cols, call):
    """skin_driver.f90's scheme sequences on whole columns.  call(fn, inputs, par0, flag, n_out) -> list of output arrays."""
    c = lambda k: sc.col(cols, k)
    n = cols.shape[1]
    out = {}
    out["cs_coare"] = call(sc.CS_COARE, [c("Qsw"), c("Qns"), c("us"), c("SST"), c("Qlat")], 0.0, 0, 1)[0]
    out["cs_coare_night"] = call(sc.CS_COARE, [np.zeros(n), c("Qns"), c("us"), c("SST"), np.zeros(n)], 0.0, 0, 1)[0]
    out["cs_ecmwf"] = call(sc.CS_ECMWF, [c("Qsw"), c("Qns"), c("us"), c("SST")], 0.0, 0, 1)[0]
    out["cs_ecmwf_heating"] = call(sc.CS_ECMWF, [c("Qsw"), -0.2 * c("Qns"), c("us"), c("SST")], 0.0, 0, 1)[0]
    st = [np.zeros(n), np.full(n, sc.HWL_MAX), np.zeros(n), np.zeros(n)]
    for jh, (h, rs) in enumerate(zip(sc.HOURS, sc.RSUN), 1):
        isd = (h * 3600) % 86400
        args = [rs * c("Qsw"), c("Qns"), c("Tau"), c("SST"), c("lon")]
        wait = call(sc.WL_COARE, args + st, float(isd), 1, 4)
        for a, b in zip(wait, st):
            np.testing.assert_array_equal(a, b)                                  # iwait = 1 hands the state back untouched
        st = [np.array(x) for x in call(sc.WL_COARE, args + st, float(isd), 0, 4)]
        for nm, v in zip(("dT", "Hz", "Qac", "Tac"), st):
            out[f"wlc_{nm}_{jh:02d}"] = v
    dT, Hz = np.zeros(n), np.full(n, sc.RD0)
    for jh, rs in enumerate(sc.RSUN, 1):
        ustk = c("ustk") if jh % 2 == 0 else None
        dT = np.array(call(sc.WL_ECMWF, [rs * c("Qsw"), c("Qns"), c("us"), c("SST"), dT, Hz, ustk], 0.0, 0, 1)[0])
        out[f"wle_dT_{jh:02d}"] = dT
    return out


# ---------------------------------------------------------------- CPU: the product header on the host
@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("skin_host") / "phymbl_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=fast", "-march=x86-64-v3", "-o", out, os.path.join(ROOT, "tests", "phymbl_host.cpp")])
    return out


def test_product_header_on_host_matches_the_reference_modules(gold, host_exe, tmp_path):
    cols, rec = gold
    fin, fout = str(tmp_path / "calls.bin"), str(tmp_path / "out.bin")

    def call(fn, inputs, par0, flag, n_out):
        present = [x for x in inputs if x is not None]
        n = present[0].size
        with open(fin, "wb") as fh:
            fh.write(struct.pack("<2i", n, len(present)))
            np.ascontiguousarray(np.stack(present)).tofile(fh)
            idx, k = [], 0
            for x in inputs:
                idx.append(-1 if x is None else k)
                k += x is not None
            fh.write(struct.pack("<i", 1))
            fh.write(struct.pack("<4id11i", fn, flag, len(inputs), n_out, par0, *(idx + [-1] * (11 - len(idx)))))
        subprocess.check_call([host_exe, fin, fout])
        return list(np.fromfile(fout).reshape(n_out, n))

    got = run_sequences(cols, call)
    worst = {k: close(v, rec[k], "host " + k) for k, v in got.items()}
    print({k: float(f"{v:.1e}") for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})


def test_reference_caller_of_the_module_arrays_fails_on_netcdf_only():
    """src/tests/test_aerobulk_buoy_series_oce.f90 (USE mod_skin_coare, ONLY: Qnt_ac, Tau_ac) against this repository's modules: every
    message of the compiler traces back to io_ezcdf — the NetCDF layer, out of scope — and none to a module of the flux path."""
    src = "/root/reference/src/tests/test_aerobulk_buoy_series_oce.f90"
    fc = "/opt/rocm/bin/amdflang"
    if not (os.path.exists(src) and os.path.exists(fc)):
        pytest.skip("reference tree / amdflang absent")
    from aerobulk_amd import build
    build.build_fortran_host()
    pr = subprocess.run([fc, "-fsyntax-only", "-fdefault-real-8", "-I", os.path.join(ROOT, "aerobulk_amd", "fortran"), src], capture_output=True, text=True)
    errs = [ln for ln in pr.stderr.splitlines() if "error:" in ln]
    assert any("io_ezcdf" in ln for ln in errs)
    ours = ("mod_const", "mod_phymbl", "mod_skin_coare", "mod_skin_ecmwf", "mod_blk_", "mod_common_coare", "Qnt_ac", "Tau_ac", "TURB_", "turb_")
    assert not [ln for ln in errs if any(w in ln for w in ours)], errs[:20]


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_c_abi_matches_the_reference_modules(gold):
    import aerobulk_amd as ab
    cols, rec = gold

    def call(fn, inputs, par0, flag, n_out):
        outs, _ = ab.phymbl(fn, [None if x is None else np.ascontiguousarray(x) for x in inputs], par0, flag, n_out)
        return outs

    got = run_sequences(cols, call)
    worst = {k: close(v, rec[k], "ab_phymbl " + k) for k, v in got.items()}
    print({k: float(f"{v:.1e}") for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})


@pytest.mark.gpu
def test_fortran_modules_on_the_gpu_write_what_the_reference_modules_write(gold, tmp_path):
    cols, rec = gold
    exe = os.path.join(ROOT, "aerobulk_amd", "fortran", "skin_driver.x")
    if not os.path.exists(exe):
        pytest.skip("Fortran host not built (amdflang absent at build time)")
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    sc.write_input(fin, cols)
    pr = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    got = sc.read_records(fout)
    assert set(got) == set(rec)
    worst = {}
    for k, ref in rec.items():
        turb = k.startswith(("t36_", "t30_", "tec_"))          # through TURB_*: the flux path's tolerance; the state integrals carry it
        worst[k] = close(got[k], ref, "skin_driver " + k, 1e-10 if turb else TOL)
    print({k: float(f"{v:.1e}") for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]})
