#!/usr/bin/env python
"""End of a short launch of flux_kernel_cu, team by team (trace variant -DAB_CU_TRACE -DAB_CU_TRACE_END): when each of the 1 024 teams
entered, started its last tile and left, on the chip-wide 100 MHz clock.   python tools/cu_trace_end.py [rows]      (GPU box)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import torch
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
nj = int(sys.argv[2])
f = ab.synth_fields_device(4320, 3600, 0, nj)
with ab.Session("coare3p6", 4320, nj, 1, True) as s:
    for i in range(3):
        print(f"LAUNCH {i}", flush=True)
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], check=False)
        torch.cuda.synchronize()
"""
nj = sys.argv[1] if len(sys.argv) > 1 else "450"
e = dict(os.environ, AEROBULK_AMD_CU_KERNEL="1", AEROBULK_AMD_LIB=os.path.join(ROOT, "build", "var", os.environ.get("CU_TRACE_LIB", "libab_traceend.so")))
pr = subprocess.run([sys.executable, "-c", CHILD, ROOT, nj], env=e, capture_output=True, text=True)
import re
allrows = [[int(x) for x in m] for m in re.findall(r"CUEND (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)(?!\d)", pr.stdout)]
launches = [allrows[i:i + 1024] for i in range(0, len(allrows), 1024)]      # (device lines and host lines interleave: cut by count)
for i, rows in enumerate(launches):
    a = np.array(rows, dtype=np.int64)
    if a.size == 0:
        continue
    t0 = a[:, 2].min()
    ent, last, ext, nt = (a[:, 2] - t0) / 100., (a[:, 3] - t0) / 100., (a[:, 4] - t0) / 100., a[:, 5]
    print(f"launch {i}: {len(a)} teams, rows {nj}; times in us since the first team's entry")
    print(f"  entry      : max {ent.max():7.1f}")
    print(f"  own tiles end: p10 {np.percentile(last, 10):7.1f}  median {np.median(last):7.1f}  p90 {np.percentile(last, 90):7.1f}  max {last.max():7.1f}")
    print(f"  exit       :            p10 {np.percentile(ext, 10):7.1f}  median {np.median(ext):7.1f}  p90 {np.percentile(ext, 90):7.1f}  max {ext.max():7.1f}")
    for t in range(4):
        m = a[:, 1] == t
        print(f"  team {t}: tiles mean {nt[m].mean():5.2f} (min {nt[m].min()}, max {nt[m].max()}); exit median {np.median(ext[m]):7.1f}, p90 {np.percentile(ext[m], 90):7.1f}, max {ext[m].max():7.1f}; "
              f"own work ended median {np.median(last[m]):7.1f}, p90 {np.percentile(last[m], 90):7.1f}")
    busy = np.array([(last > x).sum() for x in np.arange(0., min(last.max(), 600.), 10.)])
    print("  teams still on tiles of their own at t = 0, 10, 20 ... us:", " ".join(str(b) for b in busy))
print(pr.stderr[-800:])
