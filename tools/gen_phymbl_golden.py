#!/usr/bin/env python3
"""Golden data of tests/test_phymbl.py: the reference's own mod_phymbl, driven by aerobulk_amd/fortran/phymbl_driver.f90.

Build container only (needs oracle/_ref/ref_phymbl_driver.x = that driver linked against the UNMODIFIED reference modules,
oracle/Makefile).  Stores inputs and every record the driver writes in tests/golden/phymbl.npz: data, no reference source."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phymbl_cases as pc  # noqa: E402


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_phymbl_driver.x")
    if not os.path.exists(exe):
        sys.exit("build oracle/_ref first: make -C oracle all")
    cols = pc.make_columns()
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        pc.write_input(fin, cols)
        subprocess.check_call([exe, fin, fout])
        rec = pc.read_records(fout)
    out = os.path.join(ROOT, "tests", "golden", "phymbl.npz")
    np.savez_compressed(out, columns=cols, **{"r_" + k: v for k, v in rec.items()})
    print(f"{out}: {len(rec)} records, n = {cols.shape[1]}")
    for k in ("mod_const", "pref_sticky_s", "variance_vmean", "type_of_humidity"):
        print(k, rec[k])


if __name__ == "__main__":
    main()
