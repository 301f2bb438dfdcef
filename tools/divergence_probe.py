#!/usr/bin/env python
"""How much of the flux kernel's time is lane divergence on the quasi-random synthetic fields?  Re-run the same cells
globally sorted by (stability sign, warm-layer active): every wave becomes uniform.  Upper bound for in-block regrouping."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def timed(s, f, skin, it=5, reps=5):
    best = 1e9
    for _ in range(reps):
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=it, rad_sw=f["rad_sw"] if skin else None,
                  rad_lw=f["rad_lw"] if skin else None, check=False)
        best = min(best, s.last_kernel_ms())
    return best


def main():
    ni, nj = 4320, 3600
    f = ab.synth_fields_device(ni, nj)
    with ab.Session("coare3p6", ni, nj, 1, False) as s:      # clocks ramp up during the first ~100 ms of work
        for _ in range(60):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
        s.last_kernel_ms()
    for algo, skin in (("coare3p6", True), ("coare3p6", False), ("ecmwf", True), ("ncar", False)):
        with ab.Session(algo, ni, nj, 1, skin) as s:
            t_on = timed(s, f, skin)
            s.set_regroup(False)
            t0 = timed(s, f, skin)
            d = s.set_diagnostics(("L", "dT_wl", "dT_cs"), device="cuda")
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"] if skin else None,
                      rad_lw=f["rad_lw"] if skin else None)
            stable = (d["L"] > 0).to(torch.int64)
            wl = (d["dT_wl"] != 0).to(torch.int64)
            s.set_diagnostics(None)
            res = [f"{algo} skin={int(skin)}: regrouped {t_on:.3f} ms; natural order {t0:.3f} ms; stable frac {stable.double().mean():.3f}, wl-active frac {wl.double().mean():.3f}"]
            for name, key in (("by stability", stable), ("by wl", wl), ("by both", stable * 2 + wl)):
                idx = torch.argsort(key, stable=True)
                g = {k: v[idx].contiguous() for k, v in f.items()}
                res.append(f"{name} {timed(s, g, skin):.3f} ms")
            print("; ".join(res), flush=True)


if __name__ == "__main__":
    main()
