#!/usr/bin/env python
"""Kernel timing of the TURB_* entry (ab_session_turb) on device-resident fields, next to the fused flux kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def main():
    ni, nj = 4320, 3600
    f = ab.synth_fields_device(ni, nj)
    wnd = torch.sqrt(f["U_zu"] ** 2 + f["V_zu"] ** 2)
    theta = f["t_zt"] + 0.0196
    qs = 0.98 * 3.8e-3 * torch.exp(0.0687 * (f["sst"] - 273.15))
    qsw = 0.934 * f["rad_sw"]
    for algo, skin in (("coare3p6", True), ("coare3p6", False), ("ecmwf", True), ("ncar", False)):
        with ab.Session(algo, ni, nj, 2, False) as s:
            for _ in range(20):
                s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
            best = 1e9
            for _ in range(5):
                T_s, q_s = f["sst"].clone(), qs.clone()
                s.turb(1, 2.0, 10.0, T_s, theta, q_s, f["hum_zt"], wnd, skin, skin, Qsw=qsw if skin else None,
                       rad_lw=f["rad_lw"] if skin else None, slp=f["slp"] if skin else None, nb_iter=5)
                best = min(best, s.last_kernel_ms())
            print(f"TURB_{algo} skin={int(skin)}: {best:.3f} ms -> {ni * nj / best / 1e3:.0f} Mcell/s", flush=True)


if __name__ == "__main__":
    main()
