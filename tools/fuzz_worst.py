#!/usr/bin/env python
"""Dump the cells of a tools/fuzz_many.py configuration that exceed the parity bar: inputs, oracle and HIP values.
    python tools/fuzz_worst.py <algo> <skin 0|1> <zt> <zu> <niter> [first_seed] [n_seeds] [cells]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aerobulk_amd as ab  # noqa: E402
import pyoracle as po  # noqa: E402
from conftest import rel_err  # noqa: E402
from test_gpu_fuzz import OUT, _fields  # noqa: E402


def main():
    algo, skin, zt, zu, niter = sys.argv[1], sys.argv[2] == "1", float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    s0 = int(sys.argv[6]) if len(sys.argv) > 6 else 100
    ns = int(sys.argv[7]) if len(sys.argv) > 7 else 12
    n = int(sys.argv[8]) if len(sys.argv) > 8 else 200_003
    nt = 3 if skin else 1
    shown = 0
    for seed in range(s0, s0 + ns):
        f = _fields(seed, n)
        keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
        f = {k: v[keep] for k, v in f.items()}
        m = int(keep.sum())
        ins = [f[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")]
        osess = po.OracleSession(algo, m, nt, skin)
        with ab.Session(algo, m, 1, nt, skin) as s:
            for jt in range(1, nt + 1):
                ref = osess.compute(jt, zt, zu, niter, *ins, rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
                got = s.compute(jt, zt, zu, *ins, Niter=niter, rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
                for k, kr in (OUT if skin else OUT[:5]):
                    e = rel_err(np.asarray(got[k]), ref[kr])
                    for i in np.nonzero(e > 1e-10)[0][:3]:
                        if shown < 12:
                            shown += 1
                            print(f"seed {seed} jt {jt} {kr} cell {i}: rel {e[i]:.2e} ref {ref[kr][i]:.17g} got {np.asarray(got[k])[i]:.17g} | "
                                  + " ".join(f"{kk}={f[kk][i]:.6g}" for kk in f) + f" | wind {np.hypot(f['u_zu'][i], f['v_zu'][i]):.4g} dT {f['t_zt'][i] - f['sst'][i]:.4g}")


if __name__ == "__main__":
    main()
