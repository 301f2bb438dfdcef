#!/usr/bin/env python
"""Per-call cost of ab_session_compute(AB_MEM_DEVICE) on a small grid (360x180 NCAR: BASELINE config 1, a 12 us kernel): the C ABI driven
through ctypes with prebuilt arguments, N calls back to back on one stream, one synchronisation at the end.  What is left after the kernel
is the library's own work per record: argument checks, hipSetDevice, event records, the launch.

    python tools/call_overhead.py [lib.so ...]        (default: the in-tree library)
"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, sys, time
sys.path.insert(0, sys.argv[1])
import torch
import aerobulk_amd as ab
from aerobulk_amd import _lib
ni, nj = 360, 180
f = ab.synth_fields_device(ni, nj)
lib = _lib.load()
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
with ab.Session("ncar", ni, nj, 1, False) as s:
    out = s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    args = [s._h, 1, C.c_double(2.0), C.c_double(10.0), 5] + [C.c_void_p(f[k].data_ptr()) for k in IN6] + [None, None] \
        + [C.c_void_p(out[k].data_ptr()) for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap")] + [None, 1, C.c_void_p(st)]
    fn = lib.ab_session_compute
    res = []
    for rep in range(5):
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n):
            fn(*args)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    # the Python wrapper on top
    t0 = time.perf_counter()
    for _ in range(3000):
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, out=out, check=False)
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / 3000 * 1e6
    k = s.last_kernel_ms() * 1e3
res.sort()
print("RESULT host time per call %.2f us, per record incl. drain %.2f us (median of 5 x 3000); through Session.compute %.2f us; kernel %.1f us"
      % (res[2][0], res[2][1], tp, k))
"""


def main():
    libs = sys.argv[1:] or [None]
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["AEROBULK_AMD_LIB"] = os.path.abspath(lib)
        o = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
        print(lib or "in-tree", ":", line[0][7:] if line else o.stderr[-1500:])


if __name__ == "__main__":
    main()
