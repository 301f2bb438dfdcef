#!/usr/bin/env python3
"""Golden data of tests/test_oce_ice.py: cells that are part leads, part sea ice — the composition of the reference's
src/ice/test_aerobulk_oce+ice.f90 on arrays (aerobulk_amd/fortran/oce_ice_driver.f90) linked against the UNMODIFIED reference modules
(oracle/_ref/ref_oce_ice_driver.x, oracle/Makefile).  Build container only.  Data: inputs and every record the driver writes."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phymbl_cases as pc  # noqa: E402   (read_records: the drivers share the record format)

FIELDS = ("sst", "sit", "t_zt", "q_zt", "W10", "frci", "SLP")


def make_inputs(n=256, seed=20251004):
    g = np.random.default_rng(seed)
    u = lambda lo, hi: g.uniform(lo, hi, n)
    sst = u(271.35, 275.0)                       # the water of the leads: near freezing
    sit = u(240.0, 271.0)                        # ice surface
    t_zt = sit + u(-5.0, 8.0)
    slp = u(98000.0, 103000.0)
    es = 611.2 * np.exp(17.67 * (t_zt - 273.15) / (t_zt - 29.65))
    q_zt = u(0.4, 0.95) * 0.622 * es / (slp - 0.378 * es)
    w10 = u(1.0, 18.0)
    frci = u(0.1, 0.99)
    return np.stack([sst, sit, t_zt, q_zt, w10, frci, slp])


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_oce_ice_driver.x")
    if not os.path.exists(exe):
        sys.exit("make -C oracle all first (needs /root/reference)")
    x = make_inputs()
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        np.ascontiguousarray(x, dtype=np.float64).tofile(fin)
        subprocess.check_call([exe, str(x.shape[1]), fin, fout])
        rec = pc.read_records(fout)
    out = os.path.join(ROOT, "tests", "golden", "oce_ice.npz")
    np.savez_compressed(out, inputs=x, **{"r_" + k: v for k, v in rec.items()})
    print(f"{out}: {len(rec)} records, n = {x.shape[1]}")
    for k in ("w_qh", "nemo_qh", "an05_qh", "lg15_io_qh", "lg15_io_qh_cell", "lg15_io_tau"):
        print(k, float(np.min(rec[k])), float(np.max(rec[k])))


if __name__ == "__main__":
    main()
