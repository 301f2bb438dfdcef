#!/usr/bin/env python
"""End-to-end timing of the HOST calling convention (the reference's: caller arrays in host memory):
H2D staging + kernel + D2H through ab_session_compute(AB_MEM_HOST), vs the device-resident kernel time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab
from oracle import pyoracle as po

ni, nj = 4320, 3600
f = po.synth_fields(ni, nj)
ins = [f[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")]
with ab.Session("coare3p6", ni, nj, 1, True) as s:
    t0 = time.perf_counter(); s.init(*ins, rad_sw=f["rad_lw"], rad_lw=f["rad_lw"]); t_init = time.perf_counter() - t0
    out = {k: np.empty(ni * nj) for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")}
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        s.compute(1, 2.0, 10.0, *ins, Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], out=out)
        best = min(best, time.perf_counter() - t0)
    k = s.last_kernel_ms()
    import torch
    dev = {kk: torch.from_numpy(v).cuda() for kk, v in f.items()}
    d = s.compute(1, 2.0, 10.0, *[dev[kk] for kk in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")], Niter=5,
                  rad_sw=dev["rad_sw"], rad_lw=dev["rad_lw"])
    same = all(np.array_equal(out[kk], d[kk].cpu().numpy()) for kk in out)
    print("host (pipelined) path == device path:", same)
cells = ni * nj
gb = (8 + 6) * 8 * cells / 1e9
print(f"host path: {best*1e3:.1f} ms per record ({cells/best/1e6:.1f} Mcell/s, {gb/best:.1f} GB/s over PCIe incl. kernel {k:.2f} ms); AEROBULK_INIT pass {t_init*1e3:.1f} ms")
