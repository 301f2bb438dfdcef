#!/usr/bin/env python3
"""Golden data of tests/test_skin_modules.py: the reference's own mod_skin_coare / mod_skin_ecmwf (and TURB_COARE3P6 / 3P0 / ECMWF leaving their
state in those modules' PUBLIC arrays), driven by aerobulk_amd/fortran/skin_driver.f90.

Build container only (needs oracle/_ref/ref_skin_driver.x = that driver linked against the UNMODIFIED reference modules, oracle/Makefile).
Stores inputs and every record the driver writes in tests/golden/skin_modules.npz: data, no reference source."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skin_cases as sc  # noqa: E402


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_skin_driver.x")
    if not os.path.exists(exe):
        sys.exit("build oracle/_ref first: make -C oracle all")
    cols = sc.make_columns()
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        sc.write_input(fin, cols)
        subprocess.check_call([exe, fin, fout], cwd=d)
        rec = sc.read_records(fout)
    out = os.path.join(ROOT, "tests", "golden", "skin_modules.npz")
    np.savez_compressed(out, columns=cols, **{"r_" + k: v for k, v in rec.items()})
    print(f"{out}: {len(rec)} records, n = {cols.shape[1]}")
    for k in ("parameters", "cs_coare", "wlc_dT_04", "wlc_dT_07", "wle_dT_07", "t36_Qac_03", "tec_dT_03"):
        v = rec[k]
        print(f"{k:12s} min {v.min():.6g} max {v.max():.6g} nonzero {np.count_nonzero(v)}")


if __name__ == "__main__":
    main()
