#!/usr/bin/env python
"""One cell of a fuzz case (tests/test_gpu_fuzz.py) under the microscope: fluxes and every TURB_* diagnostic of the HIP kernel next
to the oracle's, record by record.  Run on the GPU box.

    python tools/cell_probe.py seed algo skin zt zu niter cell [seed algo skin zt zu niter cell ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aerobulk_amd as ab  # noqa: E402
from aerobulk_amd import _lib  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_fuzz import _fields  # noqa: E402

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT = (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s"))


def main():
    a = sys.argv[1:]
    for c in range(0, len(a), 7):
        seed, algo, skin, zt, zu, niter, cell = int(a[c]), a[c + 1], a[c + 2] == "1", float(a[c + 3]), float(a[c + 4]), int(a[c + 5]), int(a[c + 6])
        f = _fields(seed, 60000 + 13 * seed)
        if seed % 2:
            keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
            f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
        nt = 3 if skin else 1
        names = [k for k in _lib.Diag.NAMES if skin or k not in ("dT_cs", "dT_wl", "Hz_wl")]
        sub = {k: np.ascontiguousarray(v[cell:cell + 1]) for k, v in f.items()}
        so = po.OracleSession(algo, 1, nt, skin)
        print(f"seed {seed} {algo} skin={int(skin)} zt={zt} zu={zu} nb_iter={niter} cell {cell}: " + " ".join(f"{k}={sub[k][0]!r}" for k in IN8))
        with ab.Session(algo, f["sst"].size, 1, nt, skin) as s:
            d = s.set_diagnostics(names)
            for jt in range(1, nt + 1):
                got = s.compute(jt, zt, zu, *[f[k] for k in IN8[:6]], Niter=niter, rad_sw=f["rad_sw"] if skin else None,
                                rad_lw=f["rad_lw"] if skin else None)
                ref = so.compute(jt, zt, zu, niter, *[sub[k] for k in IN8[:6]], rad_sw=sub["rad_sw"] if skin else None,
                                 rad_lw=sub["rad_lw"] if skin else None, diag=True)
                for kg, kr in OUT if skin else OUT[:5]:
                    g, r = float(got[kg][cell]), float(ref[kr][0])
                    print(f"  jt={jt} {kg:7s} ref {r: .16e} hip {g: .16e} rel {abs(g - r) / max(abs(r), 1e-300):.2e}")
                for k in names:
                    g, r = float(d[k][cell]), float(ref[k][0])
                    print(f"  jt={jt} {k:7s} ref {r: .16e} hip {g: .16e} rel {abs(g - r) / max(abs(r), 1e-300):.2e}")


if __name__ == "__main__":
    main()
