#!/usr/bin/env python
"""Where a team of flux_kernel_cu spends a short launch: runs the trace variant of the library (tools/build_variant.sh trace -DAB_CU_TRACE:
teams 0 and 1 of workgroup 0 print their time stamps, ns since kernel entry: tables ready | per tile: phase 1 done, phase 2 done, phase 3
done (this wave), phase 3 done (team), phase 4 done | exit).   python tools/cu_trace.py [rows ...]      (GPU box)"""
import os
import subprocess
import sys

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
        print(f"--- launch {i} rows {nj}", flush=True)
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], check=False)
        torch.cuda.synchronize()
        print(f"    host: {s.last_kernel_ms() * 1e3:.1f} us", flush=True)
"""
for nj in (sys.argv[1:] or ["1", "28", "450"]):
    e = dict(os.environ, AEROBULK_AMD_CU_KERNEL="1", AEROBULK_AMD_LIB=os.path.join(ROOT, "build", "var", os.environ.get("CU_TRACE_LIB", "libab_trace.so")))
    pr = subprocess.run([sys.executable, "-c", CHILD, ROOT, nj], env=e, capture_output=True, text=True)
    print(pr.stdout[-6000:], pr.stderr[-1500:])
