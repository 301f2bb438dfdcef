#!/usr/bin/env python
"""End-to-end timing of the HOST calling convention (the reference's: caller arrays in pageable host memory) on the 4320x3600
grid, COARE3p6 + skin:
  * AEROBULK_MODEL first record (jt = 1: AEROBULK_INIT checks + aerobulk_compute), with the statistics riding on the pipelined
    pass (default) and as a pass of their own followed by a compute that reuses the staged fields (AEROBULK_AMD_NO_FUSED_INIT=1);
  * steady state (jt > 1): H2D | kernel | D2H chunk pipeline through ab_session_compute(AB_MEM_HOST);
  * the same record over k shards on this GPU (the row-block sharding of the library; k devices drive k PCIe links);
against the device-resident kernel time."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
from aerobulk_amd import _lib
from oracle import pyoracle as po
lib = _lib.load()
ni, nj = 4320, 3600
n = ni * nj
f = po.synth_fields(ni, nj)
out = [np.zeros(n) for _ in range(6)]           # caller-owned, already touched (a model reuses its arrays)
dp = lambda a: a.ctypes.data_as(_lib.dp)
lib.ab_model.restype = C.c_int
with ab.Session("ncar", 1024, 1) as s:          # context + clocks
    pass
ts = []
for rep in range(3):
    for jt in (1, 2, 3):
        t0 = time.perf_counter()
        rc = lib.ab_model(jt, 3, b"coare3p6", 8, C.c_double(2.0), C.c_double(10.0), dp(f["sst"]), dp(f["t_zt"]), dp(f["hum_zt"]), dp(f["u_zu"]),
                          dp(f["v_zu"]), dp(f["slp"]), dp(out[0]), dp(out[1]), dp(out[2]), dp(out[3]), dp(out[4]), 5, 1, dp(f["rad_sw"]),
                          dp(f["rad_lw"]), dp(out[5]), C.c_long(ni), C.c_long(nj), None)
        assert rc == 0, lib.ab_last_error()
        ts.append((jt, time.perf_counter() - t0))
first_ever = ts[0][1]
first = min(t for jt, t in ts[3:] if jt == 1)
steady = min(t for jt, t in ts[3:] if jt > 1)
print("RESULT", first, steady, float(out[0].sum()), first_ever)
"""


def model_times(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    o = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        return None
    _, a, b, c, d = line[0].split()
    return float(a), float(b), float(c), float(d)


def main():
    import aerobulk_amd as ab
    from oracle import pyoracle as po
    ni, nj = 4320, 3600
    cells = ni * nj
    gb = (8 + 6) * 8 * cells / 1e9
    for label, env in (("AEROBULK_MODEL, statistics fused into the pipelined pass (default)", {}),
                       ("AEROBULK_MODEL, statistics as a pass of their own + staged fields reused", {"AEROBULK_AMD_NO_FUSED_INIT": "1"}),
                       ("AEROBULK_MODEL over 2 shards on this GPU (AEROBULK_AMD_DEVICES=0,0), statistics fused into every shard's pass", {"AEROBULK_AMD_DEVICES": "0,0"}),
                       ("AEROBULK_MODEL over 2 shards, statistics as a pass of their own", {"AEROBULK_AMD_DEVICES": "0,0", "AEROBULK_AMD_NO_FUSED_INIT": "1"})):
        r = model_times(env)
        if r is None:
            print(label, "FAILED")
            continue
        print(f"{label}: first record (jt=1, incl. AEROBULK_INIT) {r[0] * 1e3:.1f} ms; steady state {r[1] * 1e3:.1f} ms per record "
              f"({cells / r[1] / 1e6:.0f} Mcell/s, {gb / r[1]:.0f} GB/s over PCIe, both directions); very first record of the process "
              f"{r[3] * 1e3:.1f} ms; sum QL {r[2]:.10e}")
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
        kd = s.last_kernel_ms()
    print(f"explicit session: ab_session_init {t_init * 1e3:.1f} ms; ab_session_compute(AB_MEM_HOST) {best * 1e3:.1f} ms per record "
          f"(kernels of the chunks {k:.2f} ms); device-resident kernel {kd:.2f} ms; host path == device path: {same}")


if __name__ == "__main__":
    main()
