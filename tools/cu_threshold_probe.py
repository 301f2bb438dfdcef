#!/usr/bin/env python
"""flux_kernel_cu against the 256-thread flux_kernel on slabs of the benchmark grid (what one rank of 8 / 4 / 2 owns): kernel time per
cell with AEROBULK_AMD_CU_KERNEL=1 (forced) and =0 (never), same box, interleaved.  Picks the size from which the CU kernel pays.
    python tools/cu_threshold_probe.py            (GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
out = {}
with ab.Session("coare3p6", 4320, 3600, 1, False) as s:      # clock ramp
    f = ab.synth_fields_device(4320, 3600)
    for _ in range(60):
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
    s.last_kernel_ms()
for nj in (28, 56, 113, 225, 450, 900, 1800, 3600):
    f = ab.synth_fields_device(4320, 3600, 0, nj)
    with ab.Session("coare3p6", 4320, nj, 1, True) as s:
        ms = []
        for _ in range(14):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], check=False)
            ms.append(s.last_kernel_ms())
        ms.sort()
        out[nj] = ms[len(ms) // 2]
print("RESULT " + json.dumps(out))
"""
res = {"0": [], "1": []}
for p in range(3):
    for mode in (("0", "1") if p % 2 == 0 else ("1", "0")):
        e = dict(os.environ, AEROBULK_AMD_CU_KERNEL=mode)
        pr = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=e, capture_output=True, text=True)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            raise SystemExit(pr.stdout[-1000:] + pr.stderr[-3000:])
        res[mode].append(json.loads(line[-1][7:]))
print(f"{'rows of 4320':>13s} {'cells':>10s} {'tiles':>7s}   block kernel [ms]   CU kernel [ms]    CU / block    Mcell/s (CU)")
for nj in res["0"][0]:
    a = sorted(r[nj] for r in res["0"])[1]
    b = sorted(r[nj] for r in res["1"])[1]
    n = 4320 * int(nj)
    print(f"{nj:>13s} {n:10d} {n // 512:7d}   {a:12.4f}        {b:12.4f}      {b / a:8.3f}      {n / b / 1e3:9.0f}")
