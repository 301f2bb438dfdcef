#!/usr/bin/env python
"""One-off: parity of the headline configuration on ALL 15 552 000 cells of the 4320x3600 grid, HIP against the C oracle
(run on the host cores in j-blocks).  Prints the parity report of tests/conftest.py per output field."""
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NI, NJ = 4320, 3600
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")


def oracle_block(args):
    j0, njl, algo, skin, niter = args
    from oracle import pyoracle as po
    f = po.synth_fields(NI, NJ, j0, njl)
    o = po.OracleSession(algo, NI * njl, 1, skin).compute(1, 2.0, 10.0, niter, *[f[k] for k in IN6], rad_sw=f["rad_sw"] if skin else None,
                                                          rad_lw=f["rad_lw"] if skin else None)
    o = {k: o[k] for k in ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")}
    o.update({"in_" + k: v for k, v in f.items()})      # the inputs travel back so that HIP gets the very same bits
    return j0, o


def main():
    algo = sys.argv[1] if len(sys.argv) > 1 else "coare3p6"
    skin = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
    niter = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    nproc = min(os.cpu_count() or 1, 48)
    per = -(-NJ // (nproc * 4))
    blocks = [(j0, min(per, NJ - j0), algo, skin, niter) for j0 in range(0, NJ, per)]
    t0 = time.time()
    ref = {k: np.empty(NI * NJ) for k in ("ql", "qh", "tau_x", "tau_y", "evap", "t_s") + tuple("in_" + k for k in IN6 + ("rad_sw", "rad_lw"))}
    with ProcessPoolExecutor(nproc) as ex:
        for j0, o in ex.map(oracle_block, blocks):
            for k, v in o.items():
                ref[k][j0 * NI:j0 * NI + v.size] = v
    t_or = time.time() - t0
    import aerobulk_amd as ab
    from conftest import parity_report
    import torch
    fd = ab.synth_fields_device(NI, NJ)
    names = dict(sst="sst", t_zt="t_zt", hum_zt="hum_zt", u_zu="U_zu", v_zu="V_zu", slp="slp", rad_sw="rad_sw", rad_lw="rad_lw")
    f = {}
    for k, kd in names.items():
        f[kd] = torch.from_numpy(ref.pop("in_" + k)).cuda()
        same = torch.equal(f[kd], fd[kd])
        print(f"device-generated {kd} == host-generated: {same}" + ("" if same else f" (max rel diff {float(((f[kd] - fd[kd]).abs() / f[kd].abs().clamp_min(1e-300)).max()):.1e})"))
    with ab.Session(algo, NI, NJ, 1, skin) as s:
        got = s.compute(1, 2.0, 10.0, *[f[k] for k in ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")], Niter=niter,
                        rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
    g = {kr: got[k].cpu().numpy() for k, kr in (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s")) if k in got}
    rep = parity_report(g, ref, list(g), 1e-10)
    print(f"oracle on {nproc} processes: {t_or:.1f} s;  sum QL = {ref['ql'].sum():.14e} (oracle) {g['ql'].sum():.14e} (HIP)")
    print(algo, skin, niter, {k: (f"{v['max_rel']:.1e}", v["n_bad"], f"{v['max_abs_over_scale']:.1e}") for k, v in rep.items()})
    for k in g:     # the cells beyond the bar, if any: how small is the flux, how large the absolute difference
        top = np.abs(ref[k]).max()
        err = np.abs(g[k] - ref[k])
        bad = err > 1e-10 * np.maximum(np.abs(ref[k]), 1e-4 * top)
        if bad.any():
            print(f"{k}: {int(bad.sum())} of {bad.size} cells beyond 1e-10 max(|ref|, 1e-4 max|ref|); field max {top:.4g}; among them max |ref| "
                  f"{np.abs(ref[k][bad]).max():.3e}, max |got-ref| {err[bad].max():.3e}; whole field max |got-ref| {err.max():.3e}")


if __name__ == "__main__":
    main()
